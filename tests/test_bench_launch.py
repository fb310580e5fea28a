"""`python bench.py --gpus N` must start N ranks by itself (torch.distributed.run as a child process) or fail: it never prints a line
for fewer GPUs than were asked for (round-1 finding: it silently ran one GPU and printed n_gpus: 1)."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*argv, env=None, timeout=600):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=timeout, env=e, cwd=ROOT)


def test_more_gpus_than_visible_is_refused_without_a_json_line():
    n = torch.cuda.device_count() + 1 if torch.cuda.device_count() else 2
    r = run_bench("--gpus", str(n), "--steps", "1", "--warmup", "0")
    assert r.returncode != 0 and "n_gpus" not in r.stdout, (r.returncode, r.stdout[-500:])
    assert "refusing" in r.stderr


def test_mismatched_launcher_world_size_is_refused():
    r = run_bench("--gpus", "4", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "n_gpus" not in r.stdout and "refusing" in r.stderr


@pytest.mark.gpu
def test_bench_gpus_2_launches_two_rccl_ranks_when_the_box_has_them():
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible: RCCL refuses two ranks on one device (profiles/r02_rccl_two_ranks_one_gpu_probe.txt)")
    r = run_bench("--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8", "--no-cpu-baseline", "--no-infer")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2" and line["config"]["global_batch"] == 16


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["seg", "rfm", "infer2", "infer4", "seg+bf16+batch", "module", "rfm_api", "seg@4", "rfm@4", "infer2@4"])
def test_bench_multi_rank_control_flow_on_a_shared_gpu(workload):
    """The N > 1 control flow of bench.py -- child torch.distributed.run launch, init, barrier-bracketed timing with MAX over ranks, the
    lockstep instrumented step (it contains collectives), teardown, ONE JSON line from rank 0 -- with two ranks sharing the test GPU over gloo
    (PISTOSEG_BENCH_TEST_BACKEND: RCCL itself refuses two ranks on one device).  Checks the line's bookkeeping, not its numbers."""
    extra, ranks = [], 2
    if "@" in workload:  # four ranks (the box allows six GPU processes at once, pytest's own included): odd bucket / shard counts per rank, deeper rendezvous
        workload, ranks = workload.split("@")[0], 4
        extra = ["--batch", "1"]
    if workload == "seg+bf16+batch":  # the second wire format and the older sharing mode through the bench's own N > 1 path (the default is reserve+queue)
        workload, extra = "seg", ["--grad-payload", "bf16", "--share", "batch"]
    argv = ["--gpus", str(ranks), "--steps", "2", "--warmup", "1", "--batch", "2", "--tile", "64", "--no-cpu-baseline", "--workload", workload, *extra]
    batch = 1 if ranks == 4 else 2
    r = run_bench(*argv, env={"PISTOSEG_BENCH_TEST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == ranks and line["config"]["parallelism"] == f"dp{ranks}" and line["steps"] == 2 and line["scaling"] == "weak"
    if workload in ("infer2", "infer4"):
        assert line["config"]["tiles_per_gpu"] == 2 * batch and line["value"] > 0
    else:
        assert line["config"]["global_batch"] == ranks * batch and line["value"] > 0
    if workload == "seg" and ranks == 2 and not extra:
        assert line["config"]["share"] == "reserve+queue" and line["config"]["reserved_cus"] == 32
    if workload in ("seg", "rfm", "module"):
        assert "roofline" in line and "cpu_baseline" not in line and "test_backend" in line and "api_path" not in line
    if workload in ("seg", "rfm", "module"):  # the gradient exchange explains itself: bytes, bucket count, exposed (un-hidden) time per step
        comm = line["comm"]
        assert comm["buckets"] >= 1 and comm["bytes_per_step"] > 4e8 * (0.5 if "bf16" in " ".join(extra) else 1) and comm["exposed_ms"] >= 0 and comm["steps_measured"] == 2
    if "--grad-payload" in extra:
        assert line["config"]["grad_payload"] == "bf16" and line["config"]["share"] == "batch" and line["final_loss"] == line["final_loss"]
