#!/bin/bash
# The driver's SCALE command with EIGHT ranks on a ONE-GPU box: torch.distributed over gloo (NPLDA_BENCH_BACKEND=gloo), the
# ranks sharing the device — real shards, real collectives (through the host), real kernels; timings meaningless.  Checks the
# N = 8 control path: shard bounds with a short last rank, allgather_bytes, the all-reduce payload, ranks_in_group, and the
# exit status of a failed alt workload under NPLDA_BENCH_ALT_FAIL_RC.     usage: gloo8_dryrun.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p $O; cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0 NPLDA_BENCH_BACKEND=gloo
python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29571 bench.py \
    --gpus 8 --steps 3 --warmup 1 > $O/bench_gloo8_dryrun.json 2> $O/bench_gloo8_dryrun.err
echo "rc $?" >> $O/bench_gloo8_dryrun.err
cp gpurun_out/bench_detail.json $O/bench_gloo8_dryrun_detail.json
# an odd row / trial / batch count: the last rank's shard is short
python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29573 bench.py \
    --gpus 8 --steps 2 --warmup 1 --workload cfg3 --enroll 2001 --test 20002 --trials 2000003 > $O/bench_gloo8_cfg3_short_last.json 2>> $O/bench_gloo8_dryrun.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29575 bench.py \
    --gpus 8 --steps 5 --warmup 2 --workload cfg2 --scaling strong --batch 4099 --table 200000 > $O/bench_gloo8_cfg2_short_last.json 2>> $O/bench_gloo8_dryrun.err
# a hung alt workload must be able to FAIL the launcher: watchdog after 0.2 s, exit status 3
NPLDA_BENCH_ALT_SECONDS=0.2 NPLDA_BENCH_ALT_FAIL_RC=3 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
    --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 8 --steps 2 --warmup 1 > $O/bench_gloo8_watchdog.json 2> $O/bench_gloo8_watchdog.err
echo "watchdog run exit status: $?" | tee $O/bench_gloo8_watchdog_rc.txt
wc -c $O/*.json; tail -c 600 $O/bench_gloo8_dryrun.json
