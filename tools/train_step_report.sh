#!/bin/bash
# The one-call training step, measured (run on the GPU box via gpurun): wall per step for the ways of feeding it, the
# kernel times of the graph's launches (rocprofv3 --kernel-trace), and the in-kernel phase stamps of its two big kernels.
# usage: tools/train_step_report.sh <tag>   -> gpurun_out/<tag>/train_step.txt
TAG=${1:-r02m}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $REPO/tools/fused_step_profile.py 150 graph inplace 4096 > /dev/null 2>&1  # (the first process on a fresh box reads 3x slow)
{
  echo "== wall per step, HIP-graph replay, 200 steps each (tools/fused_step_profile.py; one process per line)"
  for D in 150 170; do
    for B in 4096 2048; do
      for mode in inplace records rows1; do python $REPO/tools/fused_step_profile.py $D graph $mode $B 2>/dev/null | grep feed; done
    done
  done
  for D in 150 170; do
    echo
    echo "== kernel times inside the graph, D = $D, B = 4096, begin_epoch / step_record (rocprofv3 --kernel-trace)"
    rm -rf $OUT/trace$D
    rocprofv3 --kernel-trace -d $OUT/trace$D -o t -- python $REPO/tools/fused_step_profile.py $D graph records 4096 > /dev/null 2>&1
    python $REPO/tools/rocpd_summary.py $(find $OUT/trace$D -name "*.db" | head -1) | head -7
  done
  echo
  echo "== phase stamps of train_fb_small_kernel, one wave of the middle block (tools/exp_fb.hip), D = 150 then D = 170"
  [ -x $REPO/tools/exp_fb ] && { $REPO/tools/exp_fb 150 | head -4; $REPO/tools/exp_fb 170 | head -4; }
  echo
  echo "== wgrad_fm_kernel on random operands: time per launch back to back, slab sums against fp64, phase stamps (tools/exp_wgrad.hip)"
  [ -x $REPO/tools/exp_wgrad ] && { $REPO/tools/exp_wgrad 150; $REPO/tools/exp_wgrad 170 | sed -n 2,3p; }
} > $OUT/train_step.txt 2>&1
cat $OUT/train_step.txt
