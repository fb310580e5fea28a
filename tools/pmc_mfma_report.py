"""MFMA-busy share per kernel from one rocprofv3 --pmc pass (SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY
GRBM_GUI_ACTIVE): mean per dispatch after the warm-up third; MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)."""
import collections, csv, glob, sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:72]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE (own pass, kernel-trace only)")
for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0]))):
    n = max(len(v) for v in d.values())
    if n < 3 or "GRBM_GUI_ACTIVE" not in d:
        continue
    m = {c: sum(v[len(v) // 3:]) / len(v[len(v) // 3:]) for c, v in d.items()}
    cyc = m["GRBM_GUI_ACTIVE"] / 8.0
    print(f"== {k}  (dispatches {n})")
    for c in sorted(m):
        print(f"   {c:28s} {m[c]:16.1f}")
    if m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0:
        print(f"   -> MFMA busy {100.0 * m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * cyc):.1f} % of SIMD-cycles, kernel {cyc / 1e3:.0f} k GPU cycles")
