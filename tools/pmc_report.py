"""Summarise rocprofv3 --pmc CSVs: per kernel name, mean counter value per dispatch."""
import csv, glob, sys, collections
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"gpurun_out/pmc_{tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    n = max(len(v) for v in d.values())
    if n < 3: continue
    print(f"== {k}  (dispatches {n})")
    for c, v in sorted(d.items()):
        v = v[len(v)//3:]  # skip warm-up dispatches
        print(f"   {c:32s} {sum(v)/len(v):16.1f}")
