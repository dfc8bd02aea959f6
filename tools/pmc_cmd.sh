#!/bin/bash
# PMC passes (each in its own rocprofv3 run, --kernel-trace only) over an arbitrary command, summarised per kernel.
# usage: pmc_cmd.sh <tag> <command ...>      -> gpurun_out/<tag>/{fetch,write,sq,lds,wait}.txt
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters
  local name=$1; shift
  rocprofv3 --pmc $1 --kernel-trace -d $OUT/$name -o $name -- "${@:2}" > $OUT/$name.log 2> $OUT/$name.err
  python $REPO/tools/rocpd_summary.py $(find $OUT/$name -name "*.db" | head -1) > $OUT/$name.txt 2>&1
  rm -rf $OUT/$name
}
run fetch "FETCH_SIZE" "$@"
run write "WRITE_SIZE" "$@"
run sq "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "$@"
run lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_MFMA TCC_HIT_sum TCC_MISS_sum" "$@"
run inst "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_WAVES" "$@"
ls $OUT
