#!/bin/bash
# A/B of library builds on the training-step workloads, ONE gpurun call: ab_step.sh <tag> <lib under tools/ | product> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; shift; O=$R/gpurun_out/$TAG; mkdir -p $O; : > $O/ab.txt
cp $R/neuralplda_amd/libnplda_hip.so /tmp/libnplda_hip_product.so
for rep in 1 2; do for lib in "$@"; do
  if [ "$lib" = product ]; then cp /tmp/libnplda_hip_product.so $R/neuralplda_amd/libnplda_hip.so; else cp $R/tools/$lib $R/neuralplda_amd/libnplda_hip.so; fi
  for w in "cfg2" "cfg2 --batch 2048" "cfg5"; do
    NPLDA_BENCH_DETAIL=/tmp/bd.json python $R/bench.py --workload $w --steps 300 --warmup 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('rep $rep [$lib] $w: ms_per_step %.5f' % d['ms_per_step'])" >> $O/ab.txt
  done
done; done
cp /tmp/libnplda_hip_product.so $R/neuralplda_amd/libnplda_hip.so; cat $O/ab.txt
