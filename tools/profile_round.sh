#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag> [git head]     e.g. r04a $(git rev-parse --short HEAD)
# (the profiled passes run with --no-overlap: weight gradients on the launch stream, so that every kernel's duration is exclusive
#  and comparable with bench.py's live per-kernel events; the un-profiled line is the default, overlapped, run)
# writes gpurun_out/prof_<tag>/..., gpurun_out/<tag>_*.json|csv|txt (copy into profiles/ afterwards)
# PMC counters are collected in their OWN passes with --kernel-trace only (never with --stats / trace domains).
set -eo pipefail   # any failing step ends the round: a broken chain once let later steps run with an unset configuration and write mislabelled files
tag=$1
head=${2:-unknown}   # `git rev-parse --short HEAD` of the tree that was pushed (the box's snapshot has no .git)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PISTOSEG_GIT_HEAD=$head
PMC_ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-infer --no-overlap --no-power --api-steps 0"
# --- BASELINE configs[1]: bs=64 bf16 (the headline).  ONE lease: the un-profiled line first, then the kernel trace of the same command, then the
# counter passes, then the recomputation of the dominant kernel's fraction from the trace -- so that the kept line, its rocprof summary, the
# traffic figure and the board's clock / power all describe the same box
python bench.py > gpurun_out/${tag}_bench_unprofiled.json 2> gpurun_out/${tag}_bench_unprofiled.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag} -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-overlap --no-power --api-steps 0 > gpurun_out/${tag}_bench_train_bs64_bf16.json 2> gpurun_out/${tag}_bench_profiled.err
cp $(find gpurun_out/prof_${tag} -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_bench_train_bs64_bf16_kernel_stats.csv
python tools/roofline_recompute.py gpurun_out/${tag}_bench_train_bs64_bf16_kernel_stats.csv gpurun_out/${tag}_bench_train_bs64_bf16.json gpurun_out/${tag}_bench_unprofiled.json $head > gpurun_out/${tag}_roofline_recompute.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${tag}_fetch -- python bench.py $PMC_ARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${tag}_write -- python bench.py $PMC_ARGS > /dev/null 2>&1
python tools/pmc_traffic.py gpurun_out/pmc_${tag}_fetch gpurun_out/pmc_${tag}_write gpurun_out/${tag}_pmc_hbm_traffic.json $head > /dev/null
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_${tag}_mfma_1 -- python bench.py $PMC_ARGS > /dev/null 2>&1
python tools/pmc_mfma_report.py gpurun_out/pmc_${tag}_mfma_1 > gpurun_out/${tag}_pmc_mfma_busy_train_step.txt
python tools/step_profile.py > gpurun_out/${tag}_step_profile_per_launch.txt 2>&1
# --- the PARITY path: fp32 storage, exact-f32 MFMA (157 TFLOP/s matrix peak)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_fp32 -- python bench.py --precision fp32 --steps 3 --warmup 1 --no-cpu-baseline --no-overlap --no-power --api-steps 0 > gpurun_out/${tag}_bench_fp32_parity_path.json 2>> gpurun_out/${tag}_bench_profiled.err
cp $(find gpurun_out/prof_${tag}_fp32 -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_bench_fp32_parity_path_kernel_stats.csv
# --- the split paths (1e-4 logits at a third of the 16-bit rate): bench lines + kernel traces
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_x3 -- python bench.py --precision bf16x3 --steps 3 --warmup 1 --no-cpu-baseline --no-overlap --no-power --api-steps 0 > gpurun_out/${tag}_bench_bf16x3.json 2>> gpurun_out/${tag}_bench_profiled.err
cp $(find gpurun_out/prof_${tag}_x3 -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_bench_bf16x3_kernel_stats.csv
python bench.py --precision bf16x3 --steps 10 --warmup 3 --no-cpu-baseline --api-steps 0 > gpurun_out/${tag}_bench_bf16x3_unprofiled.json 2>> gpurun_out/${tag}_bench_profiled.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_hx3 -- python bench.py --precision fp16x3 --steps 3 --warmup 1 --no-cpu-baseline --no-overlap --no-power --api-steps 0 > gpurun_out/${tag}_bench_fp16x3.json 2>> gpurun_out/${tag}_bench_profiled.err
cp $(find gpurun_out/prof_${tag}_hx3 -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_bench_fp16x3_kernel_stats.csv
python bench.py --precision fp16x3 --steps 10 --warmup 3 --no-cpu-baseline --api-steps 0 > gpurun_out/${tag}_bench_fp16x3_unprofiled.json 2>> gpurun_out/${tag}_bench_profiled.err
# --- BASELINE configs[4]: BCSS 4-class, fp16 MFMA path, bs=128 -- bench line + HBM / MFMA counters
CFG5="--precision fp16 --classes 4 --batch 128"
python bench.py $CFG5 --steps 5 --warmup 2 --no-cpu-baseline --api-steps 0 > gpurun_out/${tag}_bench_cfg5_fp16_bs128.json 2>> gpurun_out/${tag}_bench_profiled.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${tag}_c5_fetch -- python bench.py $CFG5 $PMC_ARGS > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${tag}_c5_write -- python bench.py $CFG5 $PMC_ARGS > /dev/null 2>&1
python tools/pmc_traffic.py gpurun_out/pmc_${tag}_c5_fetch gpurun_out/pmc_${tag}_c5_write gpurun_out/${tag}_cfg5_fp16_bs128_pmc_hbm_traffic.json > /dev/null
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_${tag}_c5_mfma_1 -- python bench.py $CFG5 $PMC_ARGS > /dev/null 2>&1
python tools/pmc_mfma_report.py gpurun_out/pmc_${tag}_c5_mfma_1 > gpurun_out/${tag}_cfg5_fp16_bs128_pmc_mfma_busy.txt
# --- the reference-API paths (round 5): stage 5 as Lightning drives the mirrors, stage 3 as the script's train_epoch runs (eager torch loss block / fused)
python bench.py --workload module --steps 30 --no-cpu-baseline --no-power > gpurun_out/${tag}_bench_module_api.json 2>> gpurun_out/${tag}_bench_profiled.err
python bench.py --workload rfm_api --batch 32 --steps 30 > gpurun_out/${tag}_bench_rfm_api_bs32_torch_loss.json 2>> gpurun_out/${tag}_bench_profiled.err
python bench.py --workload rfm_api --fused-loss --batch 32 --steps 30 > gpurun_out/${tag}_bench_rfm_api_bs32_fused_loss.json 2>> gpurun_out/${tag}_bench_profiled.err
# --- other configs: [3] RFM stage 3, [2] stage-2 inference (with and without d4 TTA), S=256
python bench.py --workload rfm --batch 32 --steps 30 --warmup 5 > gpurun_out/${tag}_bench_cfg4_rfm_bs32.json 2>> gpurun_out/${tag}_bench_profiled.err
python bench.py --workload infer2 --steps 20 > gpurun_out/${tag}_bench_cfg3_infer2.json 2>> gpurun_out/${tag}_bench_profiled.err
python bench.py --workload infer2 --steps 20 --tta > gpurun_out/${tag}_bench_cfg3_infer2_tta.json 2>> gpurun_out/${tag}_bench_profiled.err
python bench.py --tile 256 --steps 5 --warmup 2 --no-cpu-baseline --api-steps 0 > gpurun_out/${tag}_bench_tile256.json 2>> gpurun_out/${tag}_bench_profiled.err
python bench.py --workload infer4 --tile 256 --steps 20 > gpurun_out/${tag}_bench_stage4_infer4_tile256.json 2>> gpurun_out/${tag}_bench_profiled.err
python bench.py --deterministic --no-cpu-baseline --api-steps 0 > gpurun_out/${tag}_bench_deterministic.json 2>> gpurun_out/${tag}_bench_profiled.err
