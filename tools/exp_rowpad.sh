#!/bin/bash
# A/B of the padding behind a row's candidate-list region (NPLDA_COHORT_ROWPAD floats, csrc/nplda_cohort_fused.hip
# fused_row_pad): cfg3 step / statistics time per padding (interleaved), then WRITE_SIZE / FETCH_SIZE and a kernel trace.
#   usage: exp_rowpad.sh "<pads for timing>" "<pads for PMC>"      -> gpurun_out/r05q/
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05q; mkdir -p $O
cd $R
for rep in 1 2; do
for pad in $1; do
  NPLDA_COHORT_ROWPAD=$pad python bench.py --workload cfg3 --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('pad $pad step', round(d['ms_per_step'],4), 'stats', round(r['kernel_ms'],4), 'prepared', round(r.get('stats_ms_prepared_cohort') or 0,4), 'MHz', round(r.get('sclk_mhz_under_kernel') or 0))" >> $O/rowpad_times.txt
done
done
for pad in $2; do
  NPLDA_COHORT_ROWPAD=$pad bash tools/pmc_traffic.sh r05q/pmc_pad$pad python $R/bench.py --workload cfg3 --steps 6 --warmup 2 --no-clock-probe --no-cpu-baseline
  echo "pad $pad: $(grep -E 'fused2' $O/pmc_pad$pad/write.txt | awk '{print "fused2 WRITE_KB", $6, "us", $9}') $(grep -E 'finish' $O/pmc_pad$pad/fetch.txt | awk '{print "finish FETCH_KB(x2)", $4, "us", $7}')" >> $O/rowpad_pmc.txt
done
