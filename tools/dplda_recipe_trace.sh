#!/bin/bash
# kernel trace of the DPlda recipe step at B = 256 (BCE) and B = 2048 (BCE, SoftCdet) -> gpurun_out/<tag>/dplda_step.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
: > $O/dplda_step.txt
for cfg in "256 crossentropy" "2048 crossentropy" "2048 SoftCdet"; do
  rm -rf /tmp/dp_tr
  python $R/tools/dplda_recipe_profile.py $cfg >> $O/dplda_step.txt 2>/dev/null
  rocprofv3 --kernel-trace -d /tmp/dp_tr -- python $R/tools/dplda_recipe_profile.py $cfg > /tmp/dp_tr.log 2>&1
  echo "== B, loss = $cfg (under the profiler: $(grep 'per step' /tmp/dp_tr.log))" >> $O/dplda_step.txt
  python $R/tools/rocpd_summary.py --drop-first /tmp/dp_tr/*/*.db | head -12 | cut -c1-170 >> $O/dplda_step.txt
done
cat $O/dplda_step.txt
