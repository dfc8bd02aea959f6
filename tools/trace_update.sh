#!/bin/bash
# kernel times of the cfg2 step (4096 / 2048 pairs) for library builds: trace_update.sh <tag> <lib under tools/ | product> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; shift; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; : > $O/trace.txt
cp $R/neuralplda_amd/libnplda_hip.so /tmp/libnplda_hip_product.so
for lib in "$@"; do
  if [ "$lib" = product ]; then cp /tmp/libnplda_hip_product.so $R/neuralplda_amd/libnplda_hip.so; else cp $R/tools/$lib $R/neuralplda_amd/libnplda_hip.so; fi
  for b in 4096 2048; do
    rm -rf /tmp/tru_$b
    rocprofv3 --kernel-trace -d /tmp/tru_$b -- python $R/bench.py --workload cfg2 --batch $b --no-cpu-baseline --steps 400 > /tmp/tru.log 2>&1
    echo "== [$lib] cfg2 batch $b: $(python -c "import json;d=json.loads([l for l in open("/tmp/tru.log") if l.startswith("{")][-1]);print(d['ms_per_step'])")" >> $O/trace.txt
    python $R/tools/rocpd_summary.py --drop-first /tmp/tru_$b/*/*.db | grep -E "nplda|update|calls" | grep -v "at::" | head -8 | cut -c1-170 >> $O/trace.txt
  done
done
cp /tmp/libnplda_hip_product.so $R/neuralplda_amd/libnplda_hip.so; cat $O/trace.txt
