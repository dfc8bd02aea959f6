#!/bin/bash
# kernel trace of bench.py --workload cfg3 -> gpurun_out/r05q/trace_$1.txt
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r05q; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/trq_$1 -- python $R/bench.py --workload cfg3 --no-cpu-baseline --steps 30 > /tmp/trq.log 2>&1
python $R/tools/rocpd_summary.py --drop-first /tmp/trq_$1/*/*.db > $R/gpurun_out/r05q/trace_$1.txt
grep -E "fused2|finish" $R/gpurun_out/r05q/trace_$1.txt | cut -c1-150
