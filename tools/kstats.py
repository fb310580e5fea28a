"""Per-step kernel time table from a rocprofv3 --stats kernel_stats.csv:  python tools/kstats.py <csv> <steps-in-the-profile> [min-us]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]); lo = float(sys.argv[3]) if len(sys.argv) > 3 else 8.0
tot = 0.0
for r in rows:
    name = r["Name"].replace("(anonymous namespace)::", "")
    per = float(r["TotalDurationNs"]) / steps / 1e3
    tot += per
    if per > lo:
        print(f"{name[:84]:84s} calls/step {int(r['Calls']) / steps:5.1f}  avg {float(r['AverageNs']) / 1e3:7.1f} us  per step {per:7.1f} us")
print(f"sum of kernel time per step: {tot / 1e3:.2f} ms")
