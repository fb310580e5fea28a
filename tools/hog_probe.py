"""How do the conv kernels behave when another kernel holds some CUs (an RCCL all-reduce beside the backward)?  A side stream
runs `hog` workgroups that each pin a CU slot; the conv launch is timed on the main stream meanwhile (HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pistoseg_amd import _lib, ops

lib = _lib.use_debug_library()  # the ps_debug_* switches live in libpistoseg_hip_debug.so only
D = torch.device("cuda:0")
dt = torch.bfloat16
side = torch.cuda.Stream()
cases = [("b4 512->512 3x3 (halo)", 512, 512, 3, 1, 28), ("b7 2048->4096 1x1 (ws2)", 2048, 4096, 1, 1, 28), ("b2 128->128 3x3 @112 (halo)", 128, 128, 3, 1, 112)]
for name, cin, cout, k, d, H in cases:
    n = 64
    spec = ops.ConvSpec(cin, cout, k, 1, d)
    x = torch.randn(n, H, H, cin, device=D).to(dt)
    wf = (torch.randn(cout, k, k, cin, device=D) * 0.02).to(dt)
    y = torch.empty(n, H, H, cout, device=D, dtype=dt)
    gy = torch.randn(n, H, H, cout, device=D).to(dt)
    dw = torch.zeros(cout, k, k, cin, device=D)
    fns = {"fwd": lambda: ops.conv2d_fwd(spec, x, wf, out_raw=y), "wgrad": lambda: ops.conv2d_wgrad(spec, x, gy, dw)}
    for what, fn in fns.items():
        line = f"{name:30s} {what:5s}"
        for hog, tpb in ((0, 0), (0, 2), (16, 0), (16, 2), (16, 1), (48, 0), (48, 2)):
            ops.TILES_PER_BLOCK = tpb
            ts = []
            for _ in range(3):
                fn(); torch.cuda.synchronize()
                if hog:
                    with torch.cuda.stream(side):
                        _lib.check(lib.ps_debug_hog(hog, 20000, 96 * 1024, side.cuda_stream), "hog")  # 96 KiB of LDS: the CU is lost to the conv blocks
                    # let the hog get resident first
                    torch.cuda._sleep(200000)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    fn()
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1) / 5 * 1e3)
                torch.cuda.synchronize()
            line += f" | hog={hog:2d} tpb={tpb}: {min(ts):7.1f}us"
        print(line, flush=True)
ops.TILES_PER_BLOCK = 0
