#!/bin/bash
# Stall-attribution PMC passes over the AS-norm cfg3 profile script (run via gpurun). usage: pmc_cohort.sh <tag>
TAG=${1:-pmc_cohort}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace -d $OUT/$name -o $name -- python $REPO/tools/asnorm_profile.py > /dev/null 2> $OUT/$name.err; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES
run b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU
run c TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
run d SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU
for n in a b c d; do python $REPO/tools/rocpd_summary.py $(find $OUT/$n -name "*.db" | head -1) | grep -E "counter|cohort_gemm|row_stats|cohort_fused2|cohort_finish"; done
