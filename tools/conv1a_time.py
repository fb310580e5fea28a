import sys, torch
sys.path.insert(0, '.')
import os
from pistoseg_amd import ops, _lib
if os.environ.get('PISTOSEG_HIP_DEBUG_LIB'): _lib.use_debug_library(True)
D = torch.device('cuda:0')
x = torch.randn(64, 3, 224, 224, device=D); w = torch.randn(64, 3, 3, 3, device=D) * 0.2
sc = torch.rand(64, device=D) + 0.5; sh = torch.randn(64, device=D) * 0.1
for dt in (torch.bfloat16, torch.float16):
    act = torch.empty(64, 224, 224, 64, device=D, dtype=dt); raw = torch.empty_like(act)
    for outs in ((act, None), (act, raw)):
        for _ in range(3): ops.conv1a_fwd(x, w, sc, sh, *outs)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.conv1a_fwd(x, w, sc, sh, *outs)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        mb = (38.5 + 411 * (2 if outs[1] is not None else 1))
        print(f"conv1a {dt} outs={1 + (outs[1] is not None)}: {us:.1f} us, {mb / us:.2f} TB/s")
