R=$GRAFT_REPO_ROOT
cp $R/neuralplda_amd/libnplda_hip.so /tmp/prod.so
for lib in tools/libnplda_r5.so /tmp/prod.so tools/libnplda_r5.so /tmp/prod.so; do
  cp $lib $R/neuralplda_amd/libnplda_hip.so 2>/dev/null || cp $R/$lib $R/neuralplda_amd/libnplda_hip.so
  python $R/bench.py --workload cfg3 --steps 30 --warmup 5 | python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$lib', 'stats_ms', round(d['config']['stats_ms'],4), 'sclk', round(r.get('sclk_mhz_under_kernel',0)), 'frac', round(r['frac'],4), 'prepared', round(r.get('stats_ms_prepared_cohort',0),4))"
done
cp /tmp/prod.so $R/neuralplda_amd/libnplda_hip.so
