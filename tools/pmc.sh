#!/bin/bash
# usage: tools/pmc.sh <tag> <python args...>   -- three separate PMC passes (kernel-trace only), CSV into gpurun_out/pmc_<tag>_N
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmc_${tag}_1 -- python "$@" > /dev/null 2>&1 &&
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_${tag}_2 -- python "$@" > /dev/null 2>&1 &&
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TA_BUSY_avr SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d gpurun_out/pmc_${tag}_3 -- python "$@" > /dev/null 2>&1
find gpurun_out/pmc_${tag}_* -name "*counter_collection.csv" | head
