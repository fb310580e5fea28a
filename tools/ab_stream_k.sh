set -e
B="python bench.py --no-cpu-baseline --no-power --api-steps 0 --steps 30"
for rep in 1 2; do
PISTOSEG_STREAM_K=0 $B > gpurun_out/r05c_ab_seg_off_$rep.json 2>/dev/null
$B > gpurun_out/r05c_ab_seg_sk2_$rep.json 2>/dev/null
PISTOSEG_HIP_LIB=$PWD/pistoseg_amd/libpistoseg_hip_sk1.so $B > gpurun_out/r05c_ab_seg_sk1_$rep.json 2>/dev/null
PISTOSEG_STREAM_K=0 $B --workload rfm --batch 32 > gpurun_out/r05c_ab_rfm_off_$rep.json 2>/dev/null
$B --workload rfm --batch 32 > gpurun_out/r05c_ab_rfm_sk2_$rep.json 2>/dev/null
PISTOSEG_HIP_LIB=$PWD/pistoseg_amd/libpistoseg_hip_sk1.so $B --workload rfm --batch 32 > gpurun_out/r05c_ab_rfm_sk1_$rep.json 2>/dev/null
done
echo done
