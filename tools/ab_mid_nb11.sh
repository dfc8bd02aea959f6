#!/bin/bash
# A/B of library builds on the embedding calls, ONE gpurun call: ab_mid_nb11.sh <tag> <D> <lib under tools/ | product> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; D=$2; shift 2; O=$R/gpurun_out/$TAG; mkdir -p $O; : > $O/ab.txt
cp $R/neuralplda_amd/libnplda_hip.so /tmp/libnplda_hip_product.so
for rep in 1 2; do for lib in "$@"; do
  if [ "$lib" = product ]; then cp /tmp/libnplda_hip_product.so $R/neuralplda_amd/libnplda_hip.so; else cp $R/tools/$lib $R/neuralplda_amd/libnplda_hip.so; fi
  echo "== rep $rep [$lib] D=$D" >> $O/ab.txt; python $R/tools/ab_mid_embed.py $D 2>&1 | grep "^D=" | grep embed >> $O/ab.txt
done; done
cp /tmp/libnplda_hip_product.so $R/neuralplda_amd/libnplda_hip.so; cat $O/ab.txt
