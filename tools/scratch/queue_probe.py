"""tile_queue launch option: bit-identity with the static schedule, counter self-reset over many launches, and time alone (same box A/B).
   python tools/scratch/queue_probe.py [halo|wgrad|all]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pistoseg_amd import ops, _lib
_lib.load()
D = torch.device("cuda:0")
what = sys.argv[1] if len(sys.argv) > 1 else "all"
g = torch.Generator().manual_seed(5)
def w_fwd_layout(wt): return wt.permute(0, 2, 3, 1).contiguous()
def w_dgrad_layout(wt): return wt.flip(2, 3).permute(1, 2, 3, 0).contiguous()
def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
cases = [(40, 28, 28, 256, 256, 3, 2), (64, 28, 28, 512, 512, 3, 1), (64, 28, 28, 512, 1024, 3, 2), (64, 56, 56, 256, 256, 3, 1), (24, 28, 28, 1024, 2048, 3, 4), (9, 28, 28, 512, 512, 3, 1)]
for dtype in (torch.bfloat16, torch.float16):
    for n, h, w, cin, cout, k, d in cases:
        x = torch.randn(n, h, w, cin, generator=g).to(D, dtype)
        wt = torch.randn(cout, cin, k, k, generator=g) * 0.03
        wf, wd = w_fwd_layout(wt).to(D, dtype), w_dgrad_layout(wt).to(D, dtype)
        gy = torch.randn(n, h, w, cout, generator=g).to(D, dtype)
        spec = ops.ConvSpec(cin, cout, k, 1, d)
        y = torch.empty((n, h, w, cout), device=D, dtype=dtype); gx = torch.empty((n, h, w, cin), device=D, dtype=dtype)
        dw = torch.zeros((cout, k, k, cin), device=D, dtype=torch.float32)
        def fwd(): ops.conv2d_fwd(spec, x, wf, out_raw=y)
        def dgr(): ops.conv2d_dgrad(spec, gy, wd, (h, w), out_raw=gx)
        def wgr(): dw.zero_(); ops.conv2d_wgrad(spec, x, gy, dw)
        res = {}
        for q in (0, 1):
            ops.TILE_QUEUE = q
            out = []
            if what in ("halo", "all"):
                fwd(); dgr(); out += [y.clone(), gx.clone()]
                t_f, t_d = timeit(fwd), timeit(dgr)
            else:
                t_f = t_d = 0.0
            if what in ("wgrad", "all"):
                wgr(); out += [dw.clone()]
                t_w = timeit(wgr)
            else:
                t_w = 0.0
            res[q] = (out, t_f, t_d, t_w)
        ops.TILE_QUEUE = 0
        same = [bool(torch.equal(a, b)) if a.dtype != torch.float32 else float((a - b).abs().max() / b.abs().max()) for a, b in zip(res[1][0], res[0][0])]
        print(f"{str(dtype)[6:]:9s} n={n:3d} {h}x{w} {cin:4d}->{cout:4d} d{d}: identical {same}  fwd {res[0][1]:7.1f} -> {res[1][1]:7.1f} us  dgrad {res[0][2]:7.1f} -> {res[1][2]:7.1f}  wgrad {res[0][3]:7.1f} -> {res[1][3]:7.1f}", flush=True)
# counter blocks are reused ring-wise: many launches in a row, checked against the first
ops.TILE_QUEUE = 1
n, h, w, cin, cout, k, d = cases[0]
x = torch.randn(n, h, w, cin, generator=g).to(D, torch.bfloat16)
wf = w_fwd_layout(torch.randn(cout, cin, k, k, generator=g) * 0.03).to(D, torch.bfloat16)
spec = ops.ConvSpec(cin, cout, k, 1, d)
y0 = torch.empty((n, h, w, cout), device=D, dtype=torch.bfloat16); ops.conv2d_fwd(spec, x, wf, out_raw=y0)
bad = 0
for i in range(700):
    y = torch.empty_like(y0); ops.conv2d_fwd(spec, x, wf, out_raw=y)
    if i % 50 == 0 or i > 690: bad += int(not torch.equal(y, y0))
print("700 launches on one stream (ring of 256 counter blocks): mismatches", bad)
ops.TILE_QUEUE = 0
