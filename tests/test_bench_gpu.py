"""bench.py's one-line JSON contract (driver-facing), on a tiny configuration: keys, types, the roofline / cpu_baseline objects."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    # a child process (bench.py initialises the GPU itself; never exec over an initialised parent)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_json_contract_small_config():
    d = run_bench("--batch", "2", "--tile", "64", "--steps", "2", "--warmup", "1", "--cpu-tiles", "1")
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                 ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)):
        assert isinstance(d[k], t), (k, d[k])
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["unit"] == "tiles/s" and d["value"] > 0 and "workload" in d["config"] and "model" not in d["config"]
    assert d["final_loss"] == d["final_loss"] and 0 < d["final_loss"] < 1e3  # finite: the timed steps ran on sane numbers
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("mfma", "hbm") and r["unit"] in ("TFLOP/s", "GB/s") and r["peak"] > 0 and r["achieved"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r and "kernel" in r
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["unit"] == "tiles/s" and c["value"] > 0 and c["cores"] >= 1 and isinstance(c["sample"], str)
    assert c["cores"] in {row["threads"] for row in c["sweep"]} and c["host_threads"] >= c["cores"] and len(c["sweep"]) >= 2
    assert max(row["train_tiles_s"] for row in c["sweep"]) == c["value"]  # `value` is the best of the thread sweep, `cores` the thread count that gave it
    a = d["api_path"]  # the same step through the reference's API (SegmentationModule.training_step + configure_optimizers + loss.backward), timed beside
    assert a["train_tiles_s"] > 0 and abs(a["ratio_to_native"] - a["train_tiles_s"] / d["value"]) < 1e-3 and "ArenaAdamW" in a["what"] and 0 < a["final_loss"] < 1e3
    q = d["parity_path"]  # the parity-grade precision on the same batch, with its error against the oracle forward of the cpu_baseline leg
    assert q["dtype"] == "fp16x3" and q["train_tiles_s"] > 0 and q["infer_tiles_s"] > 0 and 0 < q["final_loss"] < 1e3
    assert q["logits_rel_err_vs_oracle"] < 1e-4 and q["argmax_agreement_vs_oracle"] > 0.999
    assert "power" in d  # board power / shader clock of an extra untimed pass; None where rocm-smi is unavailable
    if d["power"] is not None:
        assert 50 < d["power"]["mean_w"] <= d["power"]["max_w"] <= 1.05 * (d["power"]["cap_w"] or 2000) and d["power"]["samples"] >= 1


def test_bench_rfm_workload_carries_roofline_and_cpu_baseline():
    d = run_bench("--workload", "rfm", "--batch", "2", "--tile", "64", "--steps", "2", "--warmup", "1", "--cpu-tiles", "1")
    assert d["value"] > 0 and "configs[3]" in d["config"]["workload"] and d["n_gpus"] == 1
    assert all(v == v and abs(v) < 1e4 for v in d["final_losses"].values())  # finite
    r, c = d["roofline"], d["cpu_baseline"]
    assert r["bound"] == "mfma" and r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and "stage-3" in c["sample"]


def test_bench_reference_api_workloads():
    """`--workload module` (stage 5 as Lightning drives the mirrors) and `--workload rfm_api` (the stage-3 train_epoch body; eager torch loss block
    and the fused one) print the contract line, with finite losses."""
    d = run_bench("--workload", "module", "--batch", "2", "--tile", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-power")
    assert d["value"] > 0 and "reference API" in d["metric"] and "training_step" in d["config"]["workload"] and 0 < d["final_loss"] < 1e3
    assert d["roofline"]["achieved"] > 0 and "api_path" not in d
    for extra in ([], ["--fused-loss"]):
        d = run_bench("--workload", "rfm_api", "--batch", "2", "--tile", "64", "--steps", "2", "--warmup", "1", *extra)
        assert d["value"] > 0 and "train_epoch" in d["config"]["workload"] and ("rfm_loss_block" in d["config"]["workload"]) == bool(extra)
        assert all(v == v and abs(v) < 1e4 for v in d["final_losses"].values())


def test_bench_infer4_workload_runs():
    d = run_bench("--workload", "infer4", "--batch", "2", "--tile", "64", "--steps", "2", "--warmup", "1")
    assert d["value"] > 0 and "infer_revise_masks.py" in d["config"]["workload"] and d["config"]["tiles_per_gpu"] == 4
    assert d["roofline_tail"]["bound"] == "hbm" and d["roofline_tail"]["achieved"] > 0


@pytest.mark.parametrize("precision", ["fp16x3", "bf16x3"])
def test_bench_split_precisions_and_bcss_variant(precision):
    """`bench.py --precision fp16x3|bf16x3` (the split paths that carry the 1e-4 tolerance) prices its roofline against a third of the 16-bit MFMA
    peak; `--classes 4` selects the reference's BCSS CE (no ignore index, targets 0..3: models/segmentation_module.py:63-66)."""
    d = run_bench("--batch", "2", "--tile", "64", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-power", "--precision", precision, "--classes", "4")
    assert d["dtype"] == precision and d["value"] > 0
    assert abs(d["roofline"]["peak"] - 2500.0 / 3) < 0.1 and d["roofline"]["kernel"].endswith(f"<{precision}>") or "wgrad" in d["roofline"]["kernel"]
    assert "ignore_index=None" in d["config"]["workload"] and "targets 0..3" in d["config"]["workload"]
