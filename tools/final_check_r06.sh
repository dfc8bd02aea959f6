#!/bin/bash
# Last evidence of round 6 on the final library: GPU tests (serial, as the driver runs them), smoke(), the driver's bench command,
# the cfg3 lines and the cfg3 kernel trace.   usage: final_check_r06.sh <tag>
TAG=${1:-r06fin4}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
( time python -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1 ) 2>&1 | grep real
tail -1 $O/gputests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2>&1 | grep real
cp gpurun_out/bench_detail.json $O/bench_detail.json
python bench.py --workload cfg3 > $O/bench_cfg3.json 2> $O/bench_cfg3.err
python bench.py --workload cfg3 --dim 170 > $O/bench_cfg3_d170.json 2>> $O/bench_cfg3.err
python bench.py --workload cfg3 --prepared-cohort > $O/bench_cfg3_prepared.json 2>> $O/bench_cfg3.err
export NPLDA_BENCH_NO_FORM_AB=1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/tr2 -- python $R/bench.py --workload cfg3 --no-cpu-baseline --steps 20 > /tmp/tr2.log 2>&1
python $R/tools/rocpd_summary.py --drop-first --by-grid /tmp/tr2/*/*.db > $O/cfg3_trace.txt
cd $R
wc -c $O/bench.json
python - <<P
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"], d["lib"], d["alt_cfg3"]["stats_ms"], d["alt_cfg3"]["roofline"], list(d)[-6:])
P
