import sys, torch
sys.path.insert(0, '/root/repo')
from pistoseg_amd import ops
D = torch.device('cuda:0')
n, H, cin, cout, k, d = 64, 28, 512, 512, 3, 1
spec = ops.ConvSpec(cin, cout, k, 1, d)
flops = 2.0 * n * H * H * cin * cout * 9
def t(fn, it=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
for dt in (torch.bfloat16, torch.float16):
    for xs, ys, name in ((1.0, 1.0, "x~1, dy~1"), (1.0, 1e-6, "dy subnormal-range (1e-6)"), (1e-6, 1.0, "x subnormal-range"), (1.0, 0.0, "dy = 0"), (1.0, 3e-4, "dy~3e-4 (normal, small)")):
        x = (torch.randn(n, H, H, cin, device=D) * xs).to(dt)
        gy = (torch.randn(n, H, H, cout, device=D) * ys).to(dt)
        dw = torch.zeros(cout, k, k, cin, device=D)
        ms = t(lambda: ops.conv2d_wgrad(spec, x, gy, dw))
        print(f"{str(dt)[6:]:9s} {name:28s}: {ms*1e3:7.1f} us  {flops/ms/1e9:6.0f} TF", flush=True)
# forward for comparison
for dt in (torch.bfloat16, torch.float16):
    for xs, ws, name in ((1.0, 0.02, "x~1, w~0.02"), (1.0, 1e-6, "w subnormal-range")):
        x = (torch.randn(n, H, H, cin, device=D) * xs).to(dt)
        wf = (torch.randn(cout, k, k, cin, device=D) * ws).to(dt)
        y = torch.empty(n, H, H, cout, device=D, dtype=dt)
        ms = t(lambda: ops.conv2d_fwd(spec, x, wf, out_raw=y))
        print(f"{str(dt)[6:]:9s} fwd {name:24s}: {ms*1e3:7.1f} us  {flops/ms/1e9:6.0f} TF", flush=True)
