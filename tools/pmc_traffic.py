"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (two output dirs) into profiles/<tag>_pmc_hbm_traffic.json: per kernel FAMILY
(kernel name + storage type, template tiling arguments dropped) the mean HBM-side bytes per API LAUNCH:
hbm_bytes_per_launch_corrected = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 summed over every dispatch of the family / launches, where
launches = C-ABI calls: one dispatch each, except that a halo conv launch = one main dispatch <.., 4> plus, for layers with a partial last
round, one tail dispatch <.., 2>, and a stride-2 data gradient = four parity-class dispatches <.., true>.  FETCH_SIZE is doubled: gfx950 reports half of wide streaming reads
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section)."""
import collections, csv, glob, json, re, sys

fetch_dir, write_dir, out = sys.argv[1:4]


def family(name):
    stem = re.sub(r"[<(].*", "", name.replace("void ", "").replace("(anonymous namespace)::", ""))
    dt = "bf16x3" if "TraitsBF16X3" in name else "fp16x3" if "TraitsF16X3" in name else "bf16" if ("TraitsBF16" in name or "DF16b" in name) else "f16" if ("TraitsF16" in name or "IDF16_" in name) else "f32" if "TraitsF32" in name else ""
    return (stem + (f"<{dt}>" if dt else ""))[:100]


def collect(d, counter):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[family(r["Kernel_Name"])][r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


fe, wr = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
res = {}
for k in sorted(set(fe) | set(wr)):
    if not ("conv_" in k or "adamw" in k or "fc8" in k or "conv1a" in k):
        continue
    f, w = fe.get(k, {}), wr.get(k, {})
    def api_launches(d):
        n = 0.0
        for name, v in d.items():
            # template arguments: halo <traits, tile width, ring depth, WI[, queue]>, ws2 <traits, pixel tile, SPLIT[, queue]>
            if re.search(r"conv_igemm_halo_kernel<.*, \d+, \d+, 2(?:, (?:true|false))*>\(", name):
                continue                       # tail dispatch of a launch already counted through its main dispatch
            n += len(v) / 4.0 if re.search(r"conv_igemm_ws2_kernel<.*, \d+, true(?:, (?:true|false))?>\(", name) else len(v)
        return max(n, 1.0)

    launches = api_launches(f if f else w)
    ft, wt = sum(sum(v) for v in f.values()), sum(sum(v) for v in w.values())
    res[k] = {"launches": launches, "dispatches": sum(len(v) for v in f.values()), "instantiations": sorted(len(v) for v in f.values()),
              "fetch_kib_per_launch": ft / launches, "write_kib_per_launch": wt / launches,
              "hbm_bytes_per_launch_corrected": (2 * ft + wt) / launches * 1024,
              "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; KiB units; FETCH_SIZE doubled (gfx950 reports half of wide streaming reads); "
                      "summed over all dispatches of the family, per API launch"}
# which build the counters belong to: bench.py drops `roofline.traffic` when the kernel sources have changed since (csrc_sha16), and names
# the commit (passed in: the GPU box's snapshot carries no .git)
import hashlib, os
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_h = hashlib.sha256()
for _f in sorted(glob.glob(os.path.join(_root, "pistoseg_amd", "csrc", "*"))):
    _h.update(open(_f, "rb").read())
res["_build"] = {"csrc_sha16": _h.hexdigest()[:16], "git_head": sys.argv[4] if len(sys.argv) > 4 else os.environ.get("PISTOSEG_GIT_HEAD", "unknown")}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_launch_corrected"] / 1e6, 1) for k, v in res.items() if not k.startswith("_")}, indent=1))
