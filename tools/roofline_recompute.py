"""Recompute the dominant kernel's roofline fraction from ONE lease's artefacts, the way a reader without the box would:

    python tools/roofline_recompute.py <kernel_stats.csv of `rocprofv3 --kernel-trace --stats -- python bench.py --steps K --warmup W --no-overlap --no-power --no-cpu-baseline`>
                                       <that run's bench JSON line> <the un-profiled bench JSON line of the same lease> [git head]

Algorithmic work: the 3x3 stride-1 layers served by conv_igemm_halo_kernel are 18,703.5 GFLOP of a bs=64 training step (forward + data
gradient) and 9,943.5 GFLOP of a bs=64 inference pass -- also re-derived below from the bench line's own per-launch figures.  The profiled process ran
(W + K + 1) training steps (warm-up, timed, one instrumented) and (max(1, W) + K) inference passes; a halo LAUNCH is one main dispatch (<.., 4>) plus, where the
schedule re-issues a partial last round, one tail dispatch (<.., 2>): total time = both rows.  achieved = GFLOP / total ns; frac = achieved / 2500 TFLOP/s (dense bf16
MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md).  The line's own `roofline` (live HIP events of one instrumented step, same process as the headline) is printed
beside it with the lease's board power / shader clock."""
import csv
import json
import re
import sys

stats_csv, prof_json, live_json = sys.argv[1:4]
head = sys.argv[4] if len(sys.argv) > 4 else "unknown"
prof = json.loads([ln for ln in open(prof_json).read().splitlines() if ln.startswith("{")][-1])
live = json.loads([ln for ln in open(live_json).read().splitlines() if ln.startswith("{")][-1])
K, W = prof["steps"], prof["warmup"]
batch = prof["config"]["per_gpu_batch"]
n_train, n_infer = W + K + 1, (max(1, W) + K if "infer_value" in prof else 0)
dt = prof["dtype"]
peak = {"bf16": 2500.0, "fp16": 2500.0, "bf16x3": 2500.0 / 3, "fp16x3": 2500.0 / 3, "fp32": 157.3}[dt]
tag = {"bf16": "TraitsBF16,", "fp16": "TraitsF16,", "bf16x3": "TraitsBF16X3,", "fp16x3": "TraitsF16X3,", "fp32": "TraitsF32,"}[dt]
rows = [r for r in csv.DictReader(open(stats_csv)) if "conv_igemm_halo_kernel" in r["Name"] and tag in r["Name"]]
# fourth template argument <traits, tile width, ring depth, WI[, queue]>: cout fragments per wave (4 = main dispatch, 2 = tail half tiles)
wi = lambda r: re.search(r"conv_igemm_halo_kernel<[^,]+(?:::[^,]+)*, \d+, \d+, (\d)(?:, (?:true|false))*>\(", r["Name"]).group(1)
main = [r for r in rows if wi(r) == "4"]
tail = [r for r in rows if wi(r) == "2"]
ns = lambda rs: sum(float(r["TotalDurationNs"]) for r in rs)
calls = lambda rs: sum(int(r["Calls"]) for r in rs)
# GFLOP of the halo launches per step, from the line itself (flops_per_launch_avg x launches_per_step of the instrumented TRAINING step)
rl = prof["roofline"]
halo_train_gf_line = rl["flops_per_launch_avg"] * rl["launches_per_step"] / 1e9 if "halo" in rl["kernel"] else float("nan")
scale = batch / 64.0 * (prof["config"]["tile"] / 224.0) ** 2
halo_train_gf, halo_infer_gf = 18703.5 * scale, 9943.5 * scale
total_gf = n_train * halo_train_gf + n_infer * halo_infer_gf
total_ns = ns(main) + ns(tail)
ach = total_gf * 1e9 / (total_ns * 1e-9) / 1e12
print(f"# roofline recomputation from one lease (build {head}; dtype {dt}, batch {batch}, tile {prof['config']['tile']})")
print(f"profiled run      : bench.py --steps {K} --warmup {W} --no-overlap  ->  {n_train} training steps + {n_infer} inference passes in the trace")
print(f"halo GFLOP / step : training {halo_train_gf:.1f} (the line's own per-launch figures: {halo_train_gf_line:.1f}), inference {halo_infer_gf:.1f}")
print(f"halo dispatches   : {calls(main)} main (<.., 4>) in {ns(main) / 1e6:.2f} ms + {calls(tail)} tail (<.., 2>) in {ns(tail) / 1e6:.2f} ms"
      f"  (expected main: {n_train} x {rl['launches_per_step']} + {n_infer} x inference launches)")
print(f"recomputed        : {total_gf:.1f} GFLOP / {total_ns / 1e6:.2f} ms = {ach:.1f} TFLOP/s = {ach / peak:.4f} of {peak:.1f} TFLOP/s")
print(f"profiled line     : roofline.achieved {rl['achieved']} TFLOP/s, frac {rl['frac']} (live HIP events, {rl['launches_per_step']} launches, avg {rl['avg_launch_us']} us)")
lr = live["roofline"]
print(f"un-profiled line  : value {live['value']} tiles/s ({live['ms_per_step']} ms/step), infer {live.get('infer_value')}; roofline.achieved {lr['achieved']} TFLOP/s, "
      f"frac {lr['frac']} (avg launch {lr['avg_launch_us']} us); power {live.get('power')}")
print(f"agreement         : recomputed - profiled line = {ach / peak - rl['frac']:+.4f}; recomputed - un-profiled line = {ach / peak - lr['frac']:+.4f}")
