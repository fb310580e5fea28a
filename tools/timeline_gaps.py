"""GPU idle time inside a training step, from a rocprofv3 --kernel-trace CSV of bench.py (default two-stream run):
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-infer --no-power
  python tools/timeline_gaps.py gpurun_out/trace
A step = the dispatches between two consecutive optimizer kernels (adamw_kernel / sgd_kernel).  Per step: wall span, time with at least one kernel
running (union over both streams), time with two kernels running, idle time, and the largest idle gaps with the kernels on either side."""
import csv, glob, os, sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name[:70]


def main():
    root = sys.argv[1]
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
    rows.sort()
    opt = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2] or "sgd_kernel" in r[2]]
    if len(opt) < 3:
        print("fewer than three optimizer dispatches in the trace"); return
    steps = [(opt[i] + 1, opt[i + 1] + 1) for i in range(len(opt) - 1)]  # dispatches after optimizer i up to and including optimizer i + 1
    agg_gaps = defaultdict(lambda: [0, 0.0])
    print(f"{len(rows)} dispatches, {len(steps)} steps between optimizer kernels")
    for si, (a, b) in enumerate(steps):
        seg = rows[a:b]
        t0, t1 = seg[0][0], max(r[1] for r in seg)
        events = []
        for s, e, _, _ in seg:
            events.append((s, 1)); events.append((e, -1))
        events.sort()
        busy1 = busy2 = 0
        depth, last = 0, t0
        for t, d in events:
            if depth >= 1: busy1 += t - last
            if depth >= 2: busy2 += t - last
            depth += d; last = t
        # idle gaps: walk kernels in start order keeping the furthest end seen
        gaps, far, far_name = [], seg[0][1], seg[0][2]
        for s, e, n, _ in seg[1:]:
            if s > far:
                gaps.append((s - far, far_name, n))
            if e > far:
                far, far_name = e, n
        span = t1 - t0
        idle = span - busy1
        print(f"step {si}: span {span / 1e6:7.3f} ms | >= 1 kernel {busy1 / 1e6:7.3f} ms | >= 2 kernels {busy2 / 1e6:6.3f} ms | idle {idle / 1e6:6.3f} ms ({100.0 * idle / span:4.1f} %) in {len(gaps)} gaps, "
              f"{sum(1 for g in gaps if g[0] > 2000)} longer than 2 us")
        for g, before, after in gaps:
            k = (short(before), short(after))
            agg_gaps[k][0] += 1; agg_gaps[k][1] += g
    print("largest idle gaps by (kernel before -> kernel after), summed over the steps above:")
    for (before, after), (cnt, tot) in sorted(agg_gaps.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"  {tot / 1e3 / len(steps):8.1f} us/step in {cnt / len(steps):5.1f} gaps/step: {before}  ->  {after}")


if __name__ == "__main__":
    main()
