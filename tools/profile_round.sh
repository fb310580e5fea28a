#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>     e.g. r01e
# (the profiled passes run with --no-overlap: weight gradients on the launch stream, so that every kernel's duration is exclusive
#  and comparable with bench.py's live per-kernel events; the un-profiled line is the default, overlapped, run)
# writes gpurun_out/prof_<tag>/..., gpurun_out/<tag>_*.json|csv|txt (copy into profiles/ afterwards)
set -o pipefail
tag=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag} -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-overlap > gpurun_out/${tag}_bench_profiled.json 2> gpurun_out/${tag}_bench_profiled.err &&
cp $(find gpurun_out/prof_${tag} -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_kernel_stats.csv &&
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${tag}_fetch -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-infer --no-overlap > /dev/null 2>&1 &&
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${tag}_write -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-infer --no-overlap > /dev/null 2>&1 &&
python tools/pmc_traffic.py gpurun_out/pmc_${tag}_fetch gpurun_out/pmc_${tag}_write gpurun_out/${tag}_pmc_hbm_traffic.json &&
python bench.py --steps 10 --warmup 3 > gpurun_out/${tag}_bench_unprofiled.json 2> gpurun_out/${tag}_bench_unprofiled.err &&
python tools/step_profile.py > gpurun_out/${tag}_step_profile_per_launch.txt 2>&1
