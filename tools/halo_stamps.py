"""Diagnostic: per-segment cycle counts of the halo kernel's consumer K-step (needs the -DPS_HALO_STAMPS A/B build:
python tools/ab_build.py stamps PS_HALO_STAMPS=1; PISTOSEG_HIP_DEBUG_LIB=pistoseg_amd/libpistoseg_hip_debug_stamps.so python tools/halo_stamps.py)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pistoseg_amd import _lib, ops
lib = _lib.use_debug_library()
D = torch.device("cuda:0"); dt = torch.bfloat16
for name, cin, cout, d, H in (("512->512 d1 @28", 512, 512, 1, 28), ("1024->2048 d4 @28", 1024, 2048, 4, 28), ("256->256 d1 @56", 256, 256, 1, 56)):
    n = 64
    spec = ops.ConvSpec(cin, cout, 3, 1, d)
    x = torch.randn(n, H, H, cin, device=D).to(dt); wf = (torch.randn(cout, 3, 3, cin, device=D) * 0.02).to(dt)
    y = torch.empty(n, H, H, cout, device=D, dtype=dt)
    for _ in range(200 if cin < 1024 else 40): ops.conv2d_fwd(spec, x, wf, out_raw=y)  # sustained load before the measured launch
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.conv2d_fwd(spec, x, wf, out_raw=y); e1.record(); torch.cuda.synchronize()
    buf = (C.c_ulonglong * (256 * 4 * 6))()
    fn = lib.ps_debug_read_stamps; fn.restype = C.c_int; fn.argtypes = [C.c_void_p]
    assert fn(buf) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 4, 6).astype(np.float64)
    steps = a[..., 3]
    per = a[..., :3] / steps[..., None]
    us = e0.elapsed_time(e1) * 1e3
    tot = per.sum(-1)
    clk = np.median(a[..., 4] / a[..., 5]) * 100.0  # shader cycles per 100 MHz tick -> MHz
    print(f"{name}: in-kernel shader clock {clk:.0f} MHz (s_memtime / s_memrealtime over the consumer loop, median of 1024 waves)")
    print(f"{name}: launch {us:.1f} us; per K-step cycles (mean over 1024 consumer waves): half0 {per[...,0].mean():.0f}  half1 {per[...,1].mean():.0f}  barrier {per[...,2].mean():.0f}"
          f"  total {tot.mean():.0f} (min {tot.min():.0f} max {tot.max():.0f}); steps/wave {steps.mean():.0f}; MFMA-only would be {28*16} + {28*16}")
