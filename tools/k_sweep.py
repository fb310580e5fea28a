"""1x1 conv throughput vs K (cin) at fixed M x Cd: separates per-tile overhead from the steady-state K loop."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pistoseg_amd import _lib, ops
lib = _lib.use_debug_library(); D = torch.device("cuda:0"); dt = torch.bfloat16  # the ps_debug_* switches live in the debug library only
n, H = 64, 28
import argparse
ap = argparse.ArgumentParser(); ap.add_argument("--ablate", type=int, default=0); args = ap.parse_args()
lib.ps_debug_set_ablate(args.ablate)
for cout in (1024, 4096):
    for cin in (256, 512, 1024, 2048, 4096, 8192):
        spec = ops.ConvSpec(cin, cout, 1, 1, 1)
        x = torch.randn(n, H, H, cin, device=D).to(dt); wf = (torch.randn(cout, 1, 1, cin, device=D) * 0.02).to(dt)
        y = torch.empty(n, H, H, cout, device=D, dtype=dt)
        fn = lambda: ops.conv2d_fwd(spec, x, wf, out_raw=y)
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 10)
        t = min(ts); fl = 2.0 * n * H * H * cin * cout
        tiles = (n * H * H // 224) * (cout // 128); rounds = -(-tiles // 256)
        print(f"1x1 {cin:5d}->{cout:5d}: {t*1e3:8.1f} us {fl/t/1e9:7.0f} TF   K-steps/tile {cin//64:4d}  tiles {tiles} rounds {rounds}  us/tile-round {t*1e3/rounds:6.1f}  us/K-step {t*1e3/rounds/(cin//64):.3f}", flush=True)
lib.ps_debug_set_ablate(0)
