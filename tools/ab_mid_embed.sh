#!/bin/bash
# A/B of library builds on the balanced-tile kernel's modes, ONE gpurun call: ab_mid_embed.sh <tag> <lib under tools/ | product> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; shift
O=$R/gpurun_out/$TAG; mkdir -p $O
cp $R/neuralplda_amd/libnplda_hip.so /tmp/libnplda_hip_product.so
: > $O/ab.txt
for rep in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = product ]; then cp /tmp/libnplda_hip_product.so $R/neuralplda_amd/libnplda_hip.so; else cp $R/tools/$lib $R/neuralplda_amd/libnplda_hip.so; fi
    for D in 150 170; do echo "== rep $rep [$lib] D=$D" >> $O/ab.txt; python $R/tools/ab_mid_embed.py $D 2>&1 | grep "^D=" >> $O/ab.txt; done
  done
done
cp /tmp/libnplda_hip_product.so $R/neuralplda_amd/libnplda_hip.so
cat $O/ab.txt
