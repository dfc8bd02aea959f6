#!/bin/bash
# A/B of library builds on the large-batch backward, ONE gpurun call: ab_backward.sh <tag> <lib under tools/ | product> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; shift; O=$R/gpurun_out/$TAG; mkdir -p $O; : > $O/ab.txt
cp $R/neuralplda_amd/libnplda_hip.so /tmp/libnplda_hip_product.so
for rep in 1 2; do for lib in "$@"; do
  if [ "$lib" = product ]; then cp /tmp/libnplda_hip_product.so $R/neuralplda_amd/libnplda_hip.so; else cp $R/tools/$lib $R/neuralplda_amd/libnplda_hip.so; fi
  for D in 150 170; do echo "== rep $rep [$lib]" >> $O/ab.txt; python $R/tools/backward_large.py $D 10 2>&1 | grep "^D=" >> $O/ab.txt; done
done; done
cp /tmp/libnplda_hip_product.so $R/neuralplda_amd/libnplda_hip.so; cat $O/ab.txt
