"""Soak: N training steps of the bench workload on ONE fixed batch (it must overfit: loss falls monotonically-ish, stays finite) in each
16-bit precision, and the f32 master weights of two identical runs must agree bit for bit except through f32-atomic ordering (reported).
`--deterministic`: SegTrainer(deterministic=True) -- the two runs must then be BIT-IDENTICAL (asserted)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pistoseg_amd.seg_model import ResNet38dSeg
from pistoseg_amd.trainer import SegTrainer, init_weights_he

D = torch.device("cuda:0")
DET = "--deterministic" in sys.argv
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
steps = int(argv[0]) if argv else 60
for prec in ("bf16", "fp16"):
    finals = []
    for run in range(2):
        torch.manual_seed(0)
        model = ResNet38dSeg(3, prec); init_weights_he(model, 42); model = model.to(D)
        tr = SegTrainer(model, lr=2e-4, weight_decay=0.01, ignore_index=3, deterministic=DET)
        g = torch.Generator().manual_seed(5)
        x = torch.randn(32, 3, 224, 224, generator=g).to(D); y = torch.randint(0, 4, (32, 224, 224), generator=g).to(D)
        losses = [float(tr.train_step(x, y)) for _ in range(steps)]
        assert all(l == l and l < 1e4 for l in losses), losses
        finals.append(tr.p_flat.clone())
        if run == 0:
            if hasattr(tr, "settle"): tr.settle()
            print(f"[{prec}] loss: " + " ".join(f"{l:.3f}" for l in losses[::max(1, steps // 12)]) + f" -> {losses[-1]:.3f}; skipped steps {tr.skipped_steps}, loss scale {tr.loss_scale}", flush=True)
            assert losses[-1] < 0.8 * losses[0], "the fixed batch is not being fitted"
    d = (finals[0] - finals[1]).abs()
    print(f"[{prec}] two identical runs{' (deterministic mode)' if DET else ''}: max |dw| {float(d.max()):.3e}, mean {float(d.mean()):.3e}"
          + ("" if DET else " (f32 atomic ordering in wgrad)"), flush=True)
    assert not DET or torch.equal(finals[0], finals[1]), "deterministic mode: two identical runs differ"
print("soak ok")
