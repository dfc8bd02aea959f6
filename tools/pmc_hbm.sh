#!/bin/bash
# HBM-side traffic of the AS-norm cfg3 statistics (separate passes, MI355X_MICROARCH.md HBM section). usage: pmc_hbm.sh <tag> [fused|spill] [D]
TAG=${1:-pmc_hbm}
shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $OUT/$c -o $c -- python $REPO/tools/asnorm_profile.py "$@" > /dev/null 2> $OUT/$c.err
  python $REPO/tools/rocpd_summary.py $(find $OUT/$c -name "*.db" | head -1) | grep -E "counter|cohort_|row_stats|rows_matmul|wgrad"
done
