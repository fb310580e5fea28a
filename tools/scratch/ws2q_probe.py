import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pistoseg_amd import ops, _lib
_lib.load()
D = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
def w_fwd_layout(wt): return wt.permute(0, 2, 3, 1).contiguous()
def w_dgrad_layout(wt): return wt.flip(2, 3).permute(1, 2, 3, 0).contiguous()
cases = [(24, 56, 56, 256, 512, 3, 2, 1), (24, 112, 112, 128, 256, 3, 2, 1), (24, 28, 28, 512, 1024, 1, 1, 1), (24, 28, 28, 1024, 512, 1, 1, 1), (24, 28, 28, 256, 512, 1, 1, 1), (24, 28, 28, 2048, 1024, 1, 1, 1), (24, 28, 28, 1536, 1024, 1, 1, 1)]
for n, h, w, cin, cout, k, s, d in cases:
    dtype = torch.bfloat16
    x = torch.randn(n, h, w, cin, generator=g).to(D, dtype)
    wt = torch.randn(cout, cin, k, k, generator=g) * 0.03
    wf, wd = w_fwd_layout(wt).to(D, dtype), w_dgrad_layout(wt).to(D, dtype)
    spec = ops.ConvSpec(cin, cout, k, s, d)
    ho, wo = spec.out_hw(h, w)
    gy = torch.randn(n, ho, wo, cout, generator=g).to(D, dtype)
    res = {}
    for q in (0, 1, 1, 1):
        ops.TILE_QUEUE = q
        y = torch.empty((n, ho, wo, cout), device=D, dtype=dtype); ops.conv2d_fwd(spec, x, wf, out_raw=y)
        gx = torch.empty((n, h, w, cin), device=D, dtype=dtype); ops.conv2d_dgrad(spec, gy, wd, (h, w), out_raw=gx)
        torch.cuda.synchronize()
        if q == 0: ref = (y, gx)
        else: res.setdefault("eq", []).append((bool(torch.equal(y, ref[0])), bool(torch.equal(gx, ref[1])), int((y != ref[0]).sum()), int((gx != ref[1]).sum())))
    ops.TILE_QUEUE = 0
    import ctypes as C
    gm = ops._geom(spec, 1, n, h, w, cin, cout)
    print(n, h, cin, cout, k, s, "variant fwd/dgrad", _lib.load().ps_conv_variant(C.byref(gm), 0), _lib.load().ps_conv_variant(C.byref(gm), 1), res["eq"], flush=True)
