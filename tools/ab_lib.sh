#!/bin/bash
# A/B of library builds under ONE command, ONE gpurun call: ab_lib.sh <tag> "<command>" <lib under tools/ | product> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; CMD=$2; shift 2; O=$R/gpurun_out/$TAG; mkdir -p $O; : > $O/ab.txt
cp $R/neuralplda_amd/libnplda_hip.so /tmp/libnplda_hip_product.so
for rep in 1 2; do for lib in "$@"; do
  if [ "$lib" = product ]; then cp /tmp/libnplda_hip_product.so $R/neuralplda_amd/libnplda_hip.so; else cp $R/tools/$lib $R/neuralplda_amd/libnplda_hip.so; fi
  echo "== rep $rep [$lib]" >> $O/ab.txt; (cd $R && eval "$CMD") 2>&1 | grep -v "amdgpu.ids" >> $O/ab.txt
done; done
cp /tmp/libnplda_hip_product.so $R/neuralplda_amd/libnplda_hip.so; cat $O/ab.txt
