import sys, torch
sys.path.insert(0, '.')
import os
from pistoseg_amd import ops, _lib
if os.environ.get("PISTOSEG_HIP_DEBUG_LIB"): _lib.use_debug_library(True)
D = torch.device('cuda:0')
for (c, hi, ho) in ((3, 28, 224), (4, 32, 256), (5, 28, 224)):
    gy = torch.randn(64, c, ho, ho, device=D); ds = torch.empty(64, hi, hi, c, device=D)
    for _ in range(3): ops.bilinear_bwd(gy, "nchw", ds, "nhwc", True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.bilinear_bwd(gy, "nchw", ds, "nhwc", True)
    e1.record(); torch.cuda.synchronize()
    print(f"bilinear_bwd C={c} {ho}->{hi}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
