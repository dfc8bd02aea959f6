#!/bin/bash
# Run on the GPU box (via gpurun): bench line + rocprofv3 kernel stats + HBM PMC passes.
# Outputs under gpurun_out/<tag>/ ; copy the summaries you want judged into profiles/.
set -u
TAG=${1:-r01}
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
find $OUT/trace -name "*kernel_stats*" | head -3
for f in $(find $OUT/trace -name "*kernel_stats.csv" | head -1); do head -12 $f; done
# HBM traffic: separate PMC passes (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2)
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o fetch -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline $EXTRA > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o write -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline $EXTRA > /dev/null 2> $OUT/pmc_write.err
find $OUT -name "*counter_collection.csv" | head
python - <<PY
import csv, glob, collections
for tag in ("fetch", "write"):
    for f in glob.glob("$OUT/pmc_%s/**/*counter_collection.csv" % tag, recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(tag, k, "n=%d mean=%.1f" % (len(v), sum(v) / len(v)))
PY
