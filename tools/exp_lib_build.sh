#!/bin/bash
# Build the library from the csrc/ of an older git revision -> tools/<name>.so (for same-box A/B runs: tools/cohort_ab.sh LIB=).
#   usage: exp_lib_build.sh <git rev> <name>
set -e
R=$(cd "$(dirname "$0")/.." && pwd); REV=$1; NAME=$2
W=/tmp/explib_$NAME; rm -rf $W; mkdir -p $W
git -C $R archive $REV neuralplda_amd/csrc include | tar -x -C $W
cd $W/neuralplda_amd/csrc
for f in *.hip *.cpp; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DNPLDA_SRC_SHA=\"$NAME\" -c $f -o $W/${f%.*}.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=libnplda_hip.map -o $R/tools/$NAME.so $W/*.o
ls -la $R/tools/$NAME.so
