"""Host-side cost of one training step (time to ENQUEUE it) against its GPU time: the margin that keeps the step GPU-bound when
several ranks share the host."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pistoseg_amd.seg_model import ResNet38dSeg
from pistoseg_amd.trainer import SegTrainer, init_weights_he
D = torch.device("cuda:0")
model = ResNet38dSeg(3, "bf16"); init_weights_he(model, seed=1); model = model.to(D)
tr = SegTrainer(model)
x = torch.randn(64, 3, 224, 224, device=D); y = torch.randint(0, 4, (64, 224, 224), device=D)
for _ in range(3): tr.train_step(x, y)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): tr.train_step(x, y)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/10:.2f} ms/step (host), complete {1e3*(t2-t0)/10:.2f} ms/step (GPU-bound if larger)")
ts = []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train_step(x, y)
    ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
print("single step enqueue on an empty queue (ms):", [round(1e3 * t, 2) for t in ts])
