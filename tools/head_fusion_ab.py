"""Same-box A/B of the inference head fusion (relu(bn7) + fc8 folded into b7's last conv launch, ps_conv1x1_head_fwd): inference batches of the bench
shape with `model.fuse_head` on and off, interleaved rounds, best of each:  python tools/head_fusion_ab.py [batch] [tile]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pistoseg_amd.seg_model import ResNet38dSeg
from pistoseg_amd.trainer import init_weights_he

D = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
s = int(sys.argv[2]) if len(sys.argv) > 2 else 224
model = ResNet38dSeg(3, "bf16"); init_weights_he(model, 42); model = model.to(D); model.eval()
x = torch.randn(n, 3, s, s, device=D)
best = {True: 1e9, False: 1e9}
with torch.no_grad():
    for _ in range(5): model(x)
    for r in range(4):
        for fuse in (True, False):
            model.fuse_head = fuse
            model(x); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): model(x)
            e1.record(); torch.cuda.synchronize()
            best[fuse] = min(best[fuse], e0.elapsed_time(e1) / 10)
for fuse in (False, True):
    print(f"fuse_head={fuse}: {best[fuse]:.3f} ms per {n}-tile batch = {n / best[fuse] * 1e3:.0f} tiles/s")
