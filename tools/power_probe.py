"""Is the conv stack power-limited?  Keeps ONE kernel family busy for a few seconds (launches queued asynchronously) and samples the
board's power and shader clock with `rocm-smi` (a child process; it never touches this process's HIP context) while the queue drains.
Prints, per kernel: TFLOP/s over the busy window, mean / max socket power, mean sclk.  An idle line first.
  python tools/power_probe.py [--seconds 3] [--data randn|zeros]
(zeros: the same instruction stream on all-zero operands -- what the clock does when nothing toggles; relu: activations max(randn, 0).)
  python tools/power_probe.py --step      the whole bf16 training step at bs = 64 (the bench's workload) instead of single kernels"""
import argparse, json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pistoseg_amd import ops

CASES = [  # name, cin, cout, k, dilation, what
    ("halo  512->512 3x3 fwd", 512, 512, 3, 1, "fwd"),
    ("halo 1024->2048 3x3 d4 fwd", 1024, 2048, 3, 4, "fwd"),
    ("gemm256 4096->4096 1x1 fwd", 4096, 4096, 1, 1, "fwd"),
    ("ws2   512->1024 1x1 fwd", 512, 1024, 1, 1, "fwd"),
    ("wgrad 512->512 3x3", 512, 512, 3, 1, "wgrad"),
    ("wgrad 2048->4096 1x1", 2048, 4096, 1, 1, "wgrad"),
]


def sample():
    """(power W, sclk MHz) from rocm-smi's JSON; None where the field is missing."""
    r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True)
    try:
        card = next(iter(json.loads([ln for ln in r.stdout.splitlines() if ln.lstrip().startswith("{")][-1]).values()))
    except Exception:
        return None, None
    p = c = None
    for k, v in card.items():
        kl = k.lower()
        if "power" in kl and p is None:
            try: p = float(str(v).split()[0])
            except ValueError: pass
        if "sclk" in kl and c is None:
            digits = "".join(ch for ch in str(v).split("(")[-1] if ch.isdigit())
            if digits: c = float(digits)
    return p, c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--data", default="randn")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--step", action="store_true")
    args = ap.parse_args()
    D = torch.device("cuda:0")
    dt = torch.bfloat16
    mk = {"randn": lambda *s: torch.randn(*s, device=D).to(dt), "zeros": lambda *s: torch.zeros(*s, device=D, dtype=dt),
          "relu": lambda *s: torch.randn(*s, device=D).relu().to(dt)}[args.data]  # relu: what a BN + ReLU output looks like (half zeros, no sign toggling)
    torch.zeros(1, device=D)
    idle = [sample() for _ in range(5)]
    print(f"idle: power {[p for p, _ in idle]} W, sclk {[c for _, c in idle]} MHz", flush=True)
    if args.step:
        from pistoseg_amd.seg_model import ResNet38dSeg
        from pistoseg_amd.trainer import SegTrainer, init_weights_he
        model = ResNet38dSeg(3, "bf16"); init_weights_he(model, seed=1); model = model.to(D)
        tr = SegTrainer(model)
        x = torch.randn(args.batch, 3, 224, 224, device=D); y = torch.randint(0, 4, (args.batch, 224, 224), device=D)
        for _ in range(3): tr.train_step(x, y)
        torch.cuda.synchronize()
        got, stop = [], threading.Event()
        def sampler():
            time.sleep(args.seconds * 0.2)
            while not stop.is_set():
                got.append(sample())
        th = threading.Thread(target=sampler); th.start()
        iters = int(args.seconds / 0.0275)
        t0 = time.perf_counter()
        for _ in range(iters): tr.train_step(x, y)
        torch.cuda.synchronize()
        secs = time.perf_counter() - t0
        stop.set(); th.join()
        ps = [p for p, _ in got if p is not None]; cs = [c for _, c in got if c is not None]
        print(f"training step bs={args.batch} bf16: {args.batch * iters / secs:7.0f} tiles/s ({secs / iters * 1e3:.2f} ms/step) | power mean {sum(ps) / max(len(ps), 1):6.0f} W max {max(ps) if ps else 0:6.0f} W"
              f" | sclk mean {sum(cs) / max(len(cs), 1):5.0f} MHz ({len(got)} samples)", flush=True)
        return
    for name, cin, cout, k, d, what in CASES:
        n, H = args.batch, 28
        spec = ops.ConvSpec(cin, cout, k, 1, d)
        x, gy = mk(n, H, H, cin), mk(n, H, H, cout)
        w = mk(cout, k, k, cin) if args.data == "zeros" else (torch.randn(cout, k, k, cin, device=D) * 0.02).to(dt)
        y = torch.empty(n, H, H, cout, device=D, dtype=dt)
        dw = torch.zeros(cout, k, k, cin, device=D)
        fn = (lambda: ops.conv2d_fwd(spec, x, w, out_raw=y)) if what == "fwd" else (lambda: ops.conv2d_wgrad(spec, x, gy, dw))
        flops = 2.0 * n * H * H * cout * cin * k * k
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): fn()
        e1.record(); torch.cuda.synchronize()
        per = e0.elapsed_time(e1) / 3 * 1e-3
        iters = max(4, int(args.seconds / per))
        got, stop = [], threading.Event()
        def sampler():  # (launching blocks once the HIP queue is full, so the samples are taken from a second thread)
            time.sleep(args.seconds * 0.2)  # let the governor settle
            while not stop.is_set():
                got.append(sample())
        th = threading.Thread(target=sampler); th.start()
        e0.record()
        for _ in range(iters): fn()
        e1.record()
        e1.synchronize()
        stop.set(); th.join()
        torch.cuda.synchronize()
        secs = e0.elapsed_time(e1) * 1e-3
        ps = [p for p, _ in got if p is not None]
        cs = [c for _, c in got if c is not None]
        mean = lambda v: sum(v) / len(v) if v else float("nan")
        print(f"{name:30s} [{args.data}] {flops * iters / secs / 1e12:7.0f} TFLOP/s over {secs:4.1f} s | power mean {mean(ps):6.0f} W max {max(ps) if ps else float('nan'):6.0f} W"
              f" | sclk mean {mean(cs):5.0f} MHz ({len(got)} samples)", flush=True)


if __name__ == "__main__":
    main()
