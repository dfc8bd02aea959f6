#!/bin/bash
# Stall-attribution PMC passes over the variant harness (run via gpurun). usage: pmc_stall.sh <tag>
TAG=${1:-r01s}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace -d $OUT/$name -o $name -- $REPO/tools/exp_fwd 20 1 150 > /dev/null 2> $OUT/$name.err; }
run a SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU
run b SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VALU_MFMA_COEXEC_CYCLES
run c SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_FLAT
for n in a b c; do python $REPO/tools/rocpd_summary.py $(find $OUT/$n -name "*.db" | head -1) | grep -E "counter|nplda_fwd_v2_kernel<10, 0, 8, false, 4>|nplda_fwd_v2_kernel<10, 0, 4, false, 4>"; done
