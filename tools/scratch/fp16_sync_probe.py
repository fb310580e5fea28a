"""What the fp16 trainers' per-step overflow check (a 4-byte D2H + host wait) costs: dynamic loss scale vs a fixed one (no check), same box."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
from pistoseg_amd.seg_model import ResNet38dSeg
from pistoseg_amd.trainer import SegTrainer, init_weights_he
D = torch.device('cuda:0')
for prec, B, c in (("fp16", 128, 4), ("fp16x3", 64, 3)):
    res = {}
    for label, ls in (("dynamic (check every step)", None), ("fixed 65536 (no check)", 65536.0)):
        model = ResNet38dSeg(c, prec); init_weights_he(model, seed=42); model = model.to(D)
        tr = SegTrainer(model, lr=2e-4, loss_scale=ls, ignore_index=None if c == 4 else 3)
        x = torch.randn(B, 3, 224, 224, device=D); y = torch.randint(0, c, (B, 224, 224), device=D)
        for _ in range(3): tr.train_step(x, y)
        best = 1e9
        for r in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(8): tr.train_step(x, y)
            torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 8)
        res[label] = best
        print(f"{prec} bs={B} {label:28s}: {best*1e3:7.2f} ms/step {B/best:7.1f} tiles/s", flush=True)
        del tr, model; torch.cuda.empty_cache()

from pistoseg_amd import ops
g = torch.randn(105_000_000, device=D)
out = torch.zeros(1, device=D, dtype=torch.int32)
ops.nonfinite_count(g, out=out); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): ops.nonfinite_count(g, out=out)
e1.record(); torch.cuda.synchronize()
print(f"nonfinite_count over 105M floats: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us  ({0.42 / (e0.elapsed_time(e1) / 10 * 1e-3):.0f} GB/s)")
