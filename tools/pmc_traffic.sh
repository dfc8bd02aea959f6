#!/bin/bash
# FETCH_SIZE and WRITE_SIZE passes only (each in its own rocprofv3 run, --kernel-trace only) over a command: the two files
# tools/traffic_from_pmc.py reads.      usage: pmc_traffic.sh <tag> <command ...>      -> gpurun_out/<tag>/{fetch,write}.txt
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for name in fetch write; do
  ctr=FETCH_SIZE; [ $name = write ] && ctr=WRITE_SIZE
  rocprofv3 --pmc $ctr --kernel-trace -d $OUT/$name -o $name -- "$@" > $OUT/$name.log 2> $OUT/$name.err
  python $REPO/tools/rocpd_summary.py $(find $OUT/$name -name "*.db" | head -1) > $OUT/$name.txt 2>&1
  rm -rf $OUT/$name
done
