"""Work items per persistent block (ps_conv_geom.tiles_per_block) vs time, no contention: 0 = one block per CU for the whole launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pistoseg_amd import _lib, ops
D = torch.device("cuda:0"); dt = torch.bfloat16
LAYERS = [("256->256 3x3 @56", 256, 256, 3, 1, 56), ("512->512 3x3 @28", 512, 512, 3, 1, 28), ("512->1024 3x3 d2", 512, 1024, 3, 2, 28),
          ("1024->2048 3x3 d4", 1024, 2048, 3, 4, 28), ("2048->4096 1x1", 2048, 4096, 1, 1, 28), ("4096->4096 1x1", 4096, 4096, 1, 1, 28), ("1024->2048 1x1", 1024, 2048, 1, 1, 28)]
n = 64
for name, cin, cout, k, d, H in LAYERS:
    spec = ops.ConvSpec(cin, cout, k, 1, d)
    x = torch.randn(n, H, H, cin, device=D).to(dt); gy = torch.randn(n, H, H, cout, device=D).to(dt)
    wf = (torch.randn(cout, k, k, cin, device=D) * 0.02).to(dt); y = torch.empty(n, H, H, cout, device=D, dtype=dt)
    dw = torch.zeros(cout, k, k, cin, device=D)
    fns = {"fwd": lambda: ops.conv2d_fwd(spec, x, wf, out_raw=y), "wgrad": lambda: ops.conv2d_wgrad(spec, x, gy, dw)}
    for what, fn in fns.items():
        line = f"{name:20s} {what:5s}"
        res = {}
        for rnd in range(3):
            for tpb in (0, 1, 2, 3, 4):
                ops.TILES_PER_BLOCK = tpb
                fn(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): fn()
                e1.record(); torch.cuda.synchronize()
                res.setdefault(tpb, []).append(e0.elapsed_time(e1) / 10)
        ops.TILES_PER_BLOCK = 0
        print(line + " | " + "  ".join(f"tpb={t}: {min(v)*1e3:7.1f}us" for t, v in res.items()), flush=True)
