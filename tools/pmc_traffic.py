"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (two output dirs) into profiles/<tag>_pmc_hbm_traffic.json:
per kernel mean per dispatch (KiB) and hbm_bytes_per_launch_corrected = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 reports half of
wide streaming reads, /opt/skills/guides/MI355X_MICROARCH.md, HBM section)."""
import csv, glob, json, sys, collections

fetch_dir, write_dir, out = sys.argv[1:4]


def collect(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    return agg


fe, wr = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
res = {}
for k in sorted(set(fe) | set(wr)):
    if not ("conv_" in k or "adamw" in k or "fc8" in k or "conv1a" in k):
        continue
    f, w = fe.get(k, [0.0]), wr.get(k, [0.0])
    fa, wa = sum(f) / len(f), sum(w) / len(w)
    res[k] = {"launches": len(f), "fetch_kib_avg": fa, "write_kib_avg": wa, "hbm_bytes_per_launch_corrected": (2 * fa + wa) * 1024,
              "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; KiB units; FETCH_SIZE doubled (gfx950 reports half of wide streaming reads)"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_launch_corrected"] / 1e6, 1) for k, v in res.items()}, indent=1))
