import sys, time, torch
sys.path.insert(0, '/root/repo')
from pistoseg_amd import ops
from pistoseg_amd.seg_model import ResNet38dSeg
from pistoseg_amd.trainer import SegTrainer, init_weights_he
D = torch.device('cuda:0')
prec, B, c = "fp16", 128, 4
for label, ls in (("dynamic", None), ("fixed 65536", 65536.0), ("fixed 1024", 1024.0)):
    model = ResNet38dSeg(c, prec); init_weights_he(model, seed=42); model = model.to(D)
    tr = SegTrainer(model, ignore_index=None, loss_scale=ls)
    x = torch.randn(B, 3, 224, 224, device=D); y = torch.randint(0, c, (B, 224, 224), device=D)
    for _ in range(30): tr.train_step(x, y)
    if ls is None: tr.settle()
    best = 1e9
    for r in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(8): loss = tr.train_step(x, y)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / 8)
    if ls is None: tr.settle()
    print(f"{label:14s}: {best*1e3:7.2f} ms/step; loss {float(loss):.4f}; skipped {tr.skipped_steps}, scale {tr.loss_scale}, steps {tr.step_count}, "
          f"nonfinite in g {int(ops.nonfinite_count(tr.g_flat))}, in p {int(ops.nonfinite_count(tr.p_flat))}", flush=True)
    del tr, model; torch.cuda.empty_cache()
