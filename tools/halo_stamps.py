"""Diagnostic: per-segment cycle counts of the halo kernel's consumer K-step, and what a tile costs OUTSIDE its K-steps (the epilogue), for the
store-only and the ResBlock epilogue (needs the -DPS_HALO_STAMPS A/B build:
python tools/ab_build.py stamps PS_HALO_STAMPS=1; PISTOSEG_HIP_DEBUG_LIB=pistoseg_amd/libpistoseg_hip_debug_stamps.so python tools/halo_stamps.py)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pistoseg_amd import _lib, ops
lib = _lib.use_debug_library()
lib.ps_debug_set_halo_sk(0); lib.ps_debug_set_halo_tail(0)  # the static schedule in ONE launch: whole tiles only (the stamps are the last launch's)
D = torch.device("cuda:0"); dt = torch.bfloat16
n = int(os.environ.get("STAMPS_BATCH", "64"))
for name0, cin, cout, d, H in (("512->512 d1 @28", 512, 512, 1, 28), ("1024->512 d2 @28", 1024, 512, 2, 28), ("1024->2048 d4 @28", 1024, 2048, 4, 28), ("256->256 d1 @56", 256, 256, 1, 56)):
    spec = ops.ConvSpec(cin, cout, 3, 1, d)
    x = torch.randn(n, H, H, cin, device=D).to(dt); wf = (torch.randn(cout, 3, 3, cin, device=D) * 0.02).to(dt)
    y = torch.empty(n, H, H, cout, device=D, dtype=dt)
    res_in = torch.randn(n, H, H, cout, device=D).to(dt); y2 = torch.empty_like(y)
    bsc, bsh = torch.rand(cout, device=D) + 0.5, torch.randn(cout, device=D)
    runs = {"raw": lambda: ops.conv2d_fwd(spec, x, wf, out_raw=y),  # store only
            "full": lambda: ops.conv2d_fwd(spec, x, wf, add0=res_in, out_raw=y, bn_scale=bsc, bn_shift=bsh, out_act=y2)}  # + residual -> raw; BN + ReLU -> 2nd output
    for epi, run in runs.items():
        name = f"{name0} [{epi}]"
        for _ in range(200 if cin < 1024 else 40): run()  # sustained load before the measured launch
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        buf = (C.c_ulonglong * (256 * 4 * 8))()
        fn = lib.ps_debug_read_stamps; fn.restype = C.c_int; fn.argtypes = [C.c_void_p]
        assert fn(buf) == 0
        a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 4, 8).astype(np.float64)
        steps = a[..., 3]
        per = a[..., :3] / steps[..., None]
        us = e0.elapsed_time(e1) * 1e3
        tot = per.sum(-1)
        clk = np.median(a[..., 4] / a[..., 5]) * 100.0  # shader cycles per 100 MHz tick -> MHz
        tiles = steps / (27.0 * cin / 64 / 3)  # K-steps per tile = 9 taps x cin / 64 K-lines ... (3 K-steps per window, 3 windows per K-line)
        outside = (a[..., 4] - a[..., :3].sum(-1)) / tiles  # cycles per tile between its last K-step and the next tile's first (epilogue + cursor)
        print(f"{name}: in-kernel shader clock {clk:.0f} MHz (s_memtime / s_memrealtime over the consumer loop, median of 1024 waves)")
        print(f"{name}: launch {us:.1f} us; per K-step cycles (mean over 1024 consumer waves): half0 {per[...,0].mean():.0f}  half1 {per[...,1].mean():.0f}  barrier {per[...,2].mean():.0f}"
              f"  total {tot.mean():.0f} (min {tot.min():.0f} max {tot.max():.0f}); steps/wave {steps.mean():.0f}; MFMA-only would be {28*16} + {28*16}")
        print(f"{name}: per tile: tail MFMAs {(a[...,6]/tiles).mean():.0f} cycles, epilogue (issue of its loads / stores, to the last instruction) {(a[...,7]/tiles).mean():.0f} cycles "
              f"= {(a[...,7]/tiles).mean()/clk:.2f} us (min {(a[...,7]/tiles).min():.0f} max {(a[...,7]/tiles).max():.0f})")
        print(f"{name}: consumer loop {a[...,4].mean():.0f} cycles = {a[...,4].mean()/clk:.1f} us; tiles/wave {tiles.mean():.2f}; OUTSIDE the K-steps per tile: "
              f"{outside.mean():.0f} cycles = {outside.mean()/clk:.2f} us (min {outside.min():.0f} max {outside.max():.0f}) = {100*outside.mean()*tiles.mean()/a[...,4].mean():.1f} % of the loop")
