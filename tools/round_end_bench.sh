#!/bin/bash
# Round-end evidence in ONE gpurun call: the driver's bench line, the --workload lines, kernel traces (with the per-launch
# series of the dominant kernel: the cold start is several launches long), PMC passes of every headline kernel, and
# profiles/traffic.json REGENERATED from those passes (tools/traffic_from_pmc.py).   usage: round_end_bench.sh <tag, e.g. r05z>
TAG=${1:-r05s}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python bench.py > $O/bench.json 2> $O/bench.err
cp $R/gpurun_out/bench_detail.json $O/bench_detail.json  # the line's full-precision objects + prose (bench.emit)
for w in cfg2 cfg3 cfg5; do python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err; done
python bench.py --workload cfg3 --prepared-cohort > $O/bench_cfg3_prepared.json 2>> $O/bench_cfg3.err
python bench.py --workload cfg2 --batch 2048 > $O/bench_cfg2_b2048.json 2>> $O/bench_cfg2.err
python bench.py --workload cfg5 --batch 2048 > $O/bench_cfg5_b2048.json 2>> $O/bench_cfg5.err
python bench.py --workload secondary > $O/extra.json 2> $O/extra.err
python tools/size_sweep.py 150 > $O/size_sweep.txt 2>&1
python tools/size_sweep.py 170 >> $O/size_sweep.txt 2>&1
export NPLDA_BENCH_NO_FORM_AB=1   # cfg3 passes below run ONE form of the fused cohort kernel (the default: split)
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/tr1 -- python $R/bench.py --no-cpu-baseline --no-alt --no-clock-probe --steps 60 > /tmp/tr1.log 2>&1
python $R/tools/rocpd_summary.py --series 12 /tmp/tr1/*/*.db > $O/bench_trace.txt
rocprofv3 --kernel-trace -d /tmp/tr2 -- python $R/bench.py --workload cfg3 --no-cpu-baseline --steps 20 > /tmp/tr2.log 2>&1
python $R/tools/rocpd_summary.py --drop-first --by-grid /tmp/tr2/*/*.db > $O/cfg3_trace.txt
rocprofv3 --kernel-trace -d /tmp/tr3 -- python $R/bench.py --workload cfg2 --no-cpu-baseline --steps 200 > /tmp/tr3.log 2>&1
python $R/tools/rocpd_summary.py --drop-first /tmp/tr3/*/*.db > $O/cfg2_trace.txt
cd $R
# PMC passes (each counter group in its own rocprofv3 run, --kernel-trace only)
bash tools/pmc_cmd.sh $TAG/pmc_fwd150 python $R/bench.py --no-cpu-baseline --no-alt --no-clock-probe --steps 20
bash tools/pmc_cmd.sh $TAG/pmc_fwd170 python $R/bench.py --no-cpu-baseline --no-alt --no-clock-probe --steps 20 --dim 170
bash tools/pmc_cmd.sh $TAG/pmc_cfg2 python $R/bench.py --workload cfg2 --steps 50 --warmup 10
bash tools/pmc_cmd.sh $TAG/pmc_cfg3 python $R/bench.py --workload cfg3 --steps 10 --warmup 3
bash tools/pmc_cmd.sh $TAG/pmc_cfg5 python $R/bench.py --workload cfg5 --steps 50 --warmup 10
bash tools/pmc_cmd.sh $TAG/pmc_regimeB150 python $R/bench.py --workload secondary --part regimeB --dim 150
bash tools/pmc_cmd.sh $TAG/pmc_regimeB170 python $R/bench.py --workload secondary --part regimeB --dim 170
bash tools/pmc_cmd.sh $TAG/pmc_gb python $R/bench.py --workload secondary --part gb
# the shard shapes of the N > 1 line (alt_cfg1_strong, alt_cfg3): rank N - 1 of N, emulated on this GPU
# (as plain one-GPU runs of the shard's shape: an --emulate-rank run also computes the other ranks' statistics once, at full size)
for n in 2 4 8; do
  bash tools/pmc_traffic.sh $TAG/pmc_cfg1_s$n python $R/bench.py --no-cpu-baseline --no-alt --no-clock-probe --steps 20 --pairs $((1048576 / n))
  bash tools/pmc_traffic.sh $TAG/pmc_cfg3_s$n python $R/bench.py --workload cfg3 --steps 10 --warmup 3 --no-clock-probe \
       --enroll $((2000 / n)) --test $((20000 / n)) --trials $((2000000 / n))
done
cp $R/profiles/traffic.json $O/traffic.json
python tools/traffic_from_pmc.py $O/traffic.json \
  score_pairs_D150_B1048576=$O/pmc_fwd150:nplda_fwd_v6_kernel \
  score_pairs_D170_B1048576=$O/pmc_fwd170:nplda_fwd_v5_kernel \
  train_step_D150_B4096=$O/pmc_cfg2:train_fb_small_kernel+wgrad_fm_kernel+train_update_kernel \
  head_step_dx_D150_B4096=$O/pmc_cfg5:train_fb_small_kernel+wgrad_fm_kernel+train_update_kernel \
  cohort_stats_D150_R22000_M10000=$O/pmc_cfg3:cohort_fused2_kernel+cohort_finish_kernel+wgrad_kernel+cohort_rowthr_kernel+cohort_prep_kernel+cohort_split_kernel+cohort_fallback_kernel \
  "score_indexed_D150_B1048576_N1200000=$O/pmc_regimeB150:score_indexed_kernel<false" \
  "score_indexed_D170_B1048576_N1200000=$O/pmc_regimeB170:score_indexed_kernel<false" \
  "score_indexed_self_D150_B1048576_N1200000=$O/pmc_regimeB150:score_indexed_kernel<true" \
  "score_indexed_self_D170_B1048576_N1200000=$O/pmc_regimeB170:score_indexed_kernel<true" \
  gb_score_D170_B524288=$O/pmc_gb:nplda_fwd_kernel \
  score_pairs_D150_B524288=$O/pmc_cfg1_s2:nplda_fwd_v6_kernel score_pairs_D150_B262144=$O/pmc_cfg1_s4:nplda_fwd_v6_kernel \
  score_pairs_D150_B131072=$O/pmc_cfg1_s8:nplda_fwd_v6_kernel \
  cohort_stats_D150_R11000_M10000=$O/pmc_cfg3_s2:cohort_fused2_kernel+cohort_finish_kernel+wgrad_kernel+cohort_rowthr_kernel+cohort_prep_kernel+cohort_split_kernel+cohort_fallback_kernel \
  cohort_stats_D150_R5500_M10000=$O/pmc_cfg3_s4:cohort_fused2_kernel+cohort_finish_kernel+wgrad_kernel+cohort_rowthr_kernel+cohort_prep_kernel+cohort_split_kernel+cohort_fallback_kernel \
  cohort_stats_D150_R2750_M10000=$O/pmc_cfg3_s8:cohort_fused2_kernel+cohort_finish_kernel+wgrad_kernel+cohort_rowthr_kernel+cohort_prep_kernel+cohort_split_kernel+cohort_fallback_kernel \
  > $O/traffic.log 2>&1
tail -3 $O/traffic.log
python tools/emulate_ranks.py > $O/emulated_ranks.txt 2>&1
tail -c 400 $O/bench.json
