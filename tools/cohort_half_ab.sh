R=$GRAFT_REPO_ROOT
for n in 8 4 2; do
for h in 1 0 1 0; do
  NPLDA_COHORT_HALF=$h python $R/bench.py --workload cfg3 --prepared-cohort --steps 30 --warmup 5 --no-clock-probe --enroll $((2000 / n)) --test $((20000 / n)) --trials $((2000000 / n)) | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('N=$n half=$h stats_ms', round(d['config']['stats_ms'],4), 'step', round(d['ms_per_step'],4))"
done; done
