#!/bin/bash
# On the GPU box: swap the ABLATE library in (scratch copy only) and run tools/exp_cohort_stamps.py.  usage: exp_cohort_run.sh <tag> <D> <abl...>
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; shift
mkdir -p $R/gpurun_out/$TAG
cp $R/neuralplda_amd/libnplda_hip.so /tmp/libnplda_hip_product.so
cp $R/tools/${ABL_LIB:-libnplda_hip_abl.so} $R/neuralplda_amd/libnplda_hip.so
python $R/tools/exp_cohort_stamps.py "$@" 2>&1 | grep -v amdgpu.ids | tee -a $R/gpurun_out/$TAG/stamps.txt
cp /tmp/libnplda_hip_product.so $R/neuralplda_amd/libnplda_hip.so
