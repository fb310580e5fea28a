#!/bin/bash
# Whole-step A/B of the halo kernel's stream-K finish on ONE box (run from the repo root on the GPU box; results -> gpurun_out/<tag>_ab_*.json):
#   off = PISTOSEG_STREAM_K=0 (no workspace handed to the launches: static schedule + half-tile tails)
#   sk2 = the default build (stream-K also under gpu_shared, i.e. in the two-stream backward)
#   sk1 = a product build with -DPS_HALO_SK=1 (stream-K in the forward only)
# usage: bash tools/ab_stream_k.sh [tag]      (profiles/r05c_convbench_stream_k.txt holds round 5's run)
set -e
tag=${1:-sk}
python tools/ab_build.py --product sk1 PS_HALO_SK=1 > /dev/null
B="python bench.py --no-cpu-baseline --no-power --api-steps 0 --steps 30"
for rep in 1 2; do
PISTOSEG_STREAM_K=0 $B > gpurun_out/${tag}_ab_seg_off_$rep.json 2>/dev/null
$B > gpurun_out/${tag}_ab_seg_sk2_$rep.json 2>/dev/null
PISTOSEG_HIP_LIB=$PWD/pistoseg_amd/libpistoseg_hip_sk1.so $B > gpurun_out/${tag}_ab_seg_sk1_$rep.json 2>/dev/null
PISTOSEG_STREAM_K=0 $B --workload rfm --batch 32 > gpurun_out/${tag}_ab_rfm_off_$rep.json 2>/dev/null
$B --workload rfm --batch 32 > gpurun_out/${tag}_ab_rfm_sk2_$rep.json 2>/dev/null
PISTOSEG_HIP_LIB=$PWD/pistoseg_amd/libpistoseg_hip_sk1.so $B --workload rfm --batch 32 > gpurun_out/${tag}_ab_rfm_sk1_$rep.json 2>/dev/null
done
rm -f pistoseg_amd/libpistoseg_hip_sk1.so
echo done
