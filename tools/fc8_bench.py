"""Times the fc8 head kernels (forward / backward) at the training shape (optimisation harness)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pistoseg_amd import ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--classes", type=int, default=3)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    D = torch.device("cuda:0")
    n, g, k, c = args.batch, 28, 4096, args.classes
    x = torch.randn(n, g, g, k, device=D).to(torch.bfloat16)
    w = torch.randn(c, k, device=D) * 0.02
    drop = (torch.rand(n, k, device=D) >= 0.5).float() * 2
    s7 = torch.rand(k, device=D) + 0.5
    cam = torch.empty(n, g, g, c, device=D)
    dcam = torch.randn(n, g, g, c, device=D)
    dx = torch.empty_like(x)
    dw = torch.zeros(c, k, device=D)
    fns = {"fc8_fwd": lambda: ops.fc8_fwd(x, w, drop, cam), "fc8_bwd": lambda: ops.fc8_bwd(x, w, drop, s7, dcam, dx, dw),
           "fc8_bwd(no drop)": lambda: ops.fc8_bwd(x, w, None, s7, dcam, dx, dw)}
    gb = x.numel() * 2 / 1e9
    for name, fn in fns.items():
        fn()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / args.iters)
        traffic = gb * (2 if "bwd" in name else 1)
        print(f"{name:18s} {best*1e3:8.1f} us  {traffic/best:6.2f} TB/s (activation traffic only)", flush=True)


if __name__ == "__main__":
    main()
