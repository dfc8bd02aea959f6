#!/bin/bash
# Experiment build of the library with ONE translation unit recompiled with extra flags -> tools/<out>.so (same-box A/B runs:
# tools/ab_mid_embed.sh, tools/cohort_ab.sh LIB=).  Run after `python -m neuralplda_amd.build` (the other objects come from
# neuralplda_amd/build/).   usage: exp_obj_build.sh <file under csrc/> <out name> <flags...>
set -e
R=$(cd "$(dirname "$0")/.." && pwd); B=$R/neuralplda_amd/build; F=$1; OUT=$2; shift 2
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden "$@" -c $R/neuralplda_amd/csrc/$F -o /tmp/exp_${OUT}.o
OBJS=$(ls $B/*.o | grep -v "/${F%.*}.o")
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$R/neuralplda_amd/csrc/libnplda_hip.map -o $R/tools/$OUT.so $OBJS /tmp/exp_${OUT}.o
ls -la $R/tools/$OUT.so
