#!/bin/bash
# Experiment build of the library with -DNPLDA_COHORT_ABLATE (timing ablations + per-tile cycle stamps in
# cohort_fused2_kernel) -> tools/libnplda_hip_abl.so.  NOT the product library: tools/exp_cohort_run.sh copies it over the
# scratch copy of neuralplda_amd/libnplda_hip.so on the GPU box only.  Run after `python -m neuralplda_amd.build`.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); B=$R/neuralplda_amd/build
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DNPLDA_COHORT_ABLATE ${EXTRA_DEFS} \
      -c $R/neuralplda_amd/csrc/nplda_cohort_fused.hip -o /tmp/nplda_cohort_fused_abl.o
OBJS=$(ls $B/*.o | grep -v nplda_cohort_fused.o)
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$R/neuralplda_amd/csrc/libnplda_hip.map \
      -o $R/tools/${OUT:-libnplda_hip_abl.so} $OBJS /tmp/nplda_cohort_fused_abl.o
ls -la $R/tools/${OUT:-libnplda_hip_abl.so}
