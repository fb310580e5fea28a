"""How long does the HOST take to enqueue one step, against how long the GPU takes to run it?

One process per GPU drives ~250 C-ABI launches per training step from Python.  At N = 8 eight such processes share one host, so the margin
between the enqueue time (CPU) and the step time (GPU) is what keeps the job GPU-bound.  Prints, for the training step and the no-grad
forward of bench.py's configs[1] workload: wall time of the enqueue alone (no synchronisation inside), GPU step time, and their ratio.
Optionally pins the process to `--cores` CPU cores first (what a rank gets on a busy host).

    python tools/host_time_probe.py [--batch 64] [--cores 2]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--cores", type=int, default=0, help="pin to this many cores (0 = leave the affinity alone)")
    args = ap.parse_args()
    if args.cores:
        os.sched_setaffinity(0, set(sorted(os.sched_getaffinity(0))[:args.cores]))
        torch.set_num_threads(args.cores)
    from pistoseg_amd.seg_model import ResNet38dSeg
    from pistoseg_amd.trainer import SegTrainer, init_weights_he

    dev = torch.device("cuda:0")
    model = ResNet38dSeg(classes=3, precision="bf16")
    init_weights_he(model, seed=42)
    model = model.to(dev)
    trainer = SegTrainer(model, lr=1e-3, weight_decay=0.05, ignore_index=3)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(args.batch, 3, 224, 224, generator=g).to(dev)
    y = torch.randint(0, 4, (args.batch, 224, 224), generator=g).to(dev)

    def infer():
        with torch.no_grad():
            model(x)

    for name, fn, prep in (("train step", lambda: trainer.train_step(x, y), model.train), ("no-grad forward", infer, model.eval)):
        prep()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        enq, tot = [], []
        for _ in range(args.steps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            enq.append(t1 - t0)
            tot.append(t2 - t0)
        enq.sort(), tot.sort()
        e, t = enq[len(enq) // 2], tot[len(tot) // 2]
        print(f"{name:16s} bs={args.batch} cores={args.cores or len(os.sched_getaffinity(0))}: enqueue {1e3 * e:6.2f} ms, enqueue+run {1e3 * t:6.2f} ms "
              f"-> host busy {100 * e / t:4.1f} % of a step")
    model.train()


if __name__ == "__main__":
    main()
