R=$GRAFT_REPO_ROOT
for s in 0 1 0 1; do
  NPLDA_COHORT_SPLIT=$s python $R/bench.py --workload cfg3 --steps 30 --warmup 5 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('split $s', 'stats_ms', round(d['config']['stats_ms'],4), 'sclk', round(r.get('sclk_mhz_under_kernel',0)), 'frac', round(r['frac'],4), 'prepared', round(r.get('stats_ms_prepared_cohort',0),4))"
done
(rocm-smi --showpower | grep -i power) &
NPLDA_COHORT_SPLIT=1 python $R/tools/cohort_clock.py 2>&1 | tail -3
