"""Whole-step A/B: tiles_per_block for the DATA-GRADIENT launches of the two-stream backward only (the weight gradients on the side stream and the
forward keep the static persistent schedule).  python tools/backward_tpb_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pistoseg_amd import ops
from pistoseg_amd.seg_model import ResNet38dSeg
from pistoseg_amd.trainer import SegTrainer, init_weights_he

D = torch.device("cuda:0")
model = ResNet38dSeg(3, "bf16"); init_weights_he(model, seed=42); model = model.to(D)
tr = SegTrainer(model)
x = torch.randn(64, 3, 224, 224, device=D); y = torch.randint(0, 4, (64, 224, 224), device=D)
_dgrad, _wgrad = ops.conv2d_dgrad, ops.conv2d_wgrad
MODE = {"dgrad": 0, "wgrad": 0}
def dgrad(*a, **k):
    prev, ops.TILES_PER_BLOCK = ops.TILES_PER_BLOCK, MODE["dgrad"]
    try: return _dgrad(*a, **k)
    finally: ops.TILES_PER_BLOCK = prev
def wgrad(*a, **k):
    prev, ops.TILES_PER_BLOCK = ops.TILES_PER_BLOCK, MODE["wgrad"]
    try: return _wgrad(*a, **k)
    finally: ops.TILES_PER_BLOCK = prev
ops.conv2d_dgrad, ops.conv2d_wgrad = dgrad, wgrad
variants = [(0, 0), (1, 0), (2, 0), (4, 0), (0, 1), (0, 2), (2, 2)]
for _ in range(5): tr.train_step(x, y)
best = {v: 1e9 for v in variants}
for r in range(3):
    for v in variants:
        MODE["dgrad"], MODE["wgrad"] = v
        tr.train_step(x, y); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): tr.train_step(x, y)
        torch.cuda.synchronize()
        best[v] = min(best[v], (time.perf_counter() - t0) / 10)
for v in variants: print(f"tiles_per_block dgrad={v[0]} wgrad={v[1]}: {1e3 * best[v]:7.3f} ms/step  {64 / best[v]:7.1f} tiles/s")
