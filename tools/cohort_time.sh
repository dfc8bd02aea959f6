# AS-norm cfg3 statistics: kernel times of one call (rocprofv3 --kernel-trace), D = ${1:-150}
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/ct -- python $R/tools/asnorm_profile.py fused ${1:-150} > /tmp/ct.log 2>&1
grep cohort_stats /tmp/ct.log
python $R/tools/rocpd_summary.py --drop-first /tmp/ct/*/*.db | grep -E "cohort|calls" | cut -c1-170
