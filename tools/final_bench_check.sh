mkdir -p gpurun_out/r06fin
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06fin/bench.json 2> gpurun_out/r06fin/bench.err ) 2>&1 | grep real
wc -c gpurun_out/r06fin/bench.json
python - <<'P'
import json
d=json.loads(open("gpurun_out/r06fin/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["frac"], d["lib"], d["alt_cfg3"]["stats_ms"], d["alt_dplda"]["B256_bce"]["ms_per_step"], list(d)[-6:])
P
