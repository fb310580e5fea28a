import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pistoseg_amd import _lib, ops
D = torch.device("cuda:0"); dt = torch.bfloat16
cin, cout, k, s, d, H = 512, 512, 3, 1, 1, 28
spec = ops.ConvSpec(cin, cout, k, s, d)
wf = (torch.randn(cout, k, k, cin, device=D) * 0.02).to(dt)
for n in list(range(48, 100, 4)) + [61, 62, 63, 65, 66, 67]:
    x = torch.randn(n, H, H, cin, device=D).to(dt); y = torch.empty(n, H, H, cout, device=D, dtype=dt)
    for _ in range(3): ops.conv2d_fwd(spec, x, wf, out_raw=y)
    torch.cuda.synchronize()
    best = 1e9
    for r in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): ops.conv2d_fwd(spec, x, wf, out_raw=y)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    M = n * H * H; blocks = ((M + 127) // 128) * (cout // 128)
    print(f"n={n:3d} M={M:6d} blocks={blocks:5d} rounds={blocks/512:5.2f} t={best*1e3:7.1f}us  {2.0*M*cout*cin*9/best/1e9:6.0f} TF  us/round={best*1e3/(blocks/512):6.1f}")
