#!/bin/bash
# A/B of cohort_fused2_kernel variants selected by environment, in ONE gpurun call (same box, interleaved):
#   usage: cohort_ab.sh <tag> "<ENV=.. settings A>" "<settings B>" ...   -> gpurun_out/<tag>/ab.txt
# A setting LIB=<file under tools/> runs that build of the library (copied over the SCRATCH copy of the product library on
# the GPU box for that run only; tools/exp_lib_build.sh <git rev> <name> makes one from an older revision of csrc/).
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; shift
O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
cp $R/neuralplda_amd/libnplda_hip.so /tmp/libnplda_hip_product.so
: > $O/ab.txt
for rep in 1 2; do
  i=0
  for cfg in "$@"; do
    i=$((i+1))
    rm -rf /tmp/ab_$i
    lib=$(echo "$cfg" | tr ' ' '\n' | grep '^LIB=' | cut -d= -f2)
    if [ -n "$lib" ]; then cp $R/tools/$lib $R/neuralplda_amd/libnplda_hip.so; else cp /tmp/libnplda_hip_product.so $R/neuralplda_amd/libnplda_hip.so; fi
    env $cfg rocprofv3 --kernel-trace -d /tmp/ab_$i -- python $R/tools/asnorm_profile.py fused ${DIM:-150} > /tmp/ab_$i.log 2>&1
    echo "== rep $rep [$cfg] $(grep cohort_stats /tmp/ab_$i.log)" >> $O/ab.txt
    python $R/tools/rocpd_summary.py --drop-first /tmp/ab_$i/*/*.db | grep -E "fused2|finish|rowthr|calls" | cut -c1-175 >> $O/ab.txt
  done
done
cp /tmp/libnplda_hip_product.so $R/neuralplda_amd/libnplda_hip.so
cat $O/ab.txt
