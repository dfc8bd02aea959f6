R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06p; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for b in 2048 4096; do
rocprofv3 --kernel-trace -d /tmp/trb_$b -- python $R/bench.py --workload cfg2 --batch $b --no-cpu-baseline --steps 400 > /tmp/trb.log 2>&1
echo "== cfg2 batch $b: $(python -c "import json;d=json.loads(open('/tmp/trb.log').read().strip().splitlines()[-1]);print(d['ms_per_step'])")" >> $O/cfg2_traces.txt
python $R/tools/rocpd_summary.py --drop-first /tmp/trb_$b/*/*.db | head -8 | cut -c1-170 >> $O/cfg2_traces.txt
done
cat $O/cfg2_traces.txt
