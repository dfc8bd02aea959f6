mkdir -p gpurun_out/r03m
python bench.py > gpurun_out/r03m/bench.json 2> gpurun_out/r03m/bench.err
for w in cfg2 cfg3 cfg5; do python bench.py --workload $w > gpurun_out/r03m/bench_$w.json 2> gpurun_out/r03m/bench_$w.err; done
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/c3 -- python $R/bench.py --workload cfg3 --no-cpu-baseline --steps 20 > /tmp/c3.log 2>&1
python $R/tools/rocpd_summary.py --drop-first --by-grid /tmp/c3/*/*.db > $R/gpurun_out/r03m/cfg3_trace.txt
tail -c 600 $R/gpurun_out/r03m/bench.json
