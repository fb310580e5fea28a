import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pistoseg_amd import _lib
from pistoseg_amd.seg_model import ResNet38dSeg
from pistoseg_amd.trainer import SegTrainer, init_weights_he
_lib.load()
D = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "fp16x3"
model = ResNet38dSeg(3, prec); init_weights_he(model, seed=42); model = model.to(D)
tr = SegTrainer(model, lr=2e-4)
x = torch.randn(64, 3, 224, 224, device=D); y = torch.randint(0, 4, (64, 224, 224), device=D)
def one(): tr.train_step(x, y)
for _ in range(4): one()
best = {0: 1e9, 1: 1e9}
for r in range(3):
    for v in (0, 1):
        model.hi_copies = bool(v)
        one(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8): one()
        torch.cuda.synchronize()
        best[v] = min(best[v], (time.perf_counter() - t0) / 8)
for v in (0, 1): print(f"{prec} hi_copies={v}: {1e3 * best[v]:7.3f} ms/step  {64 / best[v]:7.1f} tiles/s")
