#!/bin/bash
# PMC passes over the variant harness (run via gpurun). usage: pmc_run.sh <tag>
TAG=${1:-r01d}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o fetch -- $REPO/tools/exp_fwd 20 1 150 > $OUT/fetch.log 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o write -- $REPO/tools/exp_fwd 20 1 150 > /dev/null 2> $OUT/write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace -d $OUT/sq -o sq -- $REPO/tools/exp_fwd 20 1 150 > /dev/null 2> $OUT/sq.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_MFMA TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $OUT/lds -o lds -- $REPO/tools/exp_fwd 20 1 150 > /dev/null 2> $OUT/lds.err
ls $OUT/*
