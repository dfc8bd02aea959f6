mkdir -p gpurun_out/r03l
python -m pytest tests/test_cohort_fused_gpu.py tests/test_asnorm_gpu.py -x -q 2>&1 | tail -3
R=$PWD; cd /tmp; export TMPDIR=/tmp
for v in 0 1; do
  NPLDA_COHORT_V2=$v rocprofv3 --kernel-trace -d /tmp/p$v -- python $R/tools/asnorm_profile.py fused ${1:-150} > /tmp/log$v 2>&1
  grep cohort_stats /tmp/log$v
  python $R/tools/rocpd_summary.py --drop-first /tmp/p$v/*/*.db | grep -E "cohort|calls" | cut -c1-170
done
