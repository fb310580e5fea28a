import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pistoseg_amd import ops, _lib
_lib.use_debug_library()
D = torch.device("cuda:0")
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for dtype in (torch.bfloat16,):
    for n, s in ((64, 224), (32, 256)):
        x = torch.randn(n, 3, s, s, device=D)
        w1a = torch.randn(64, 3, 3, 3, device=D) * 0.3
        sc0, sh0 = torch.rand(64, device=D) + 0.5, torch.randn(64, device=D) * 0.2
        wb1 = (torch.randn(128, 1, 1, 64, device=D) * 0.1).to(dtype); w2a = (torch.randn(128, 3, 3, 64, device=D) * 0.05).to(dtype)
        sc1, sh1 = torch.rand(128, device=D) + 0.5, torch.randn(128, device=D) * 0.2
        ob, oa = torch.empty((n, s // 2, s // 2, 128), device=D, dtype=dtype), torch.empty((n, s // 2, s // 2, 128), device=D, dtype=dtype)
        a = torch.empty((n, s, s, 64), device=D, dtype=dtype)
        fused = lambda: ops.conv_front_s2(x, w1a, sc0, sh0, wb1, w2a, ob, sc1, sh1, oa)
        c1 = lambda: ops.conv1a_fwd(x, w1a, sc0, sh0, a)
        k1 = lambda: ops.conv2d_fwd(ops.ConvSpec(64, 128, 1, 2, 1), a, wb1, out_raw=ob)
        k3 = lambda: ops.conv2d_fwd(ops.ConvSpec(64, 128, 3, 2, 1), a, w2a, bn_scale=sc1, bn_shift=sh1, out_act=oa)
        print(f"n={n} {s}x{s}: fused {t(fused):7.1f} us | conv1a {t(c1):6.1f} + 1x1 s2 {t(k1):6.1f} + 3x3 s2 {t(k3):6.1f} us", flush=True)
