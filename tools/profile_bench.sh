#!/bin/bash
# Run on the GPU box (via gpurun): bench line + rocprofv3 kernel stats + HBM PMC passes (each in its own run).
# Outputs under gpurun_out/<tag>/ ; copy the summaries you want judged into profiles/.
# usage: profile_bench.sh <tag> [bench.py flags]
set -u
TAG=${1:-r02}
shift || true
EXTRA="$@"
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $REPO/bench.py $EXTRA > $OUT/bench.json 2> $OUT/bench.err
cat $OUT/bench.json
# kernel trace + stats of the same command
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $REPO/bench.py $EXTRA > $OUT/bench_traced.json 2> $OUT/trace.err
python $REPO/tools/rocpd_summary.py $(find $OUT/trace -name "*.db" | head -1) > $OUT/trace.txt
cat $OUT/trace.txt
# HBM traffic: separate PMC passes (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_$c -o $c -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline $EXTRA > /dev/null 2> $OUT/pmc_$c.err
  python $REPO/tools/rocpd_summary.py $(find $OUT/pmc_$c -name "*.db" | head -1) > $OUT/pmc_$c.txt
  grep -E "counter|nplda_fwd|cohort_|asnorm" $OUT/pmc_$c.txt | head -12
done
