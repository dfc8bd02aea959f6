#!/usr/bin/env python3
"""Does one HIP-graph replay of a rank's AS-norm step (rows embed -> prepared cohort statistics -> apply) beat the eager
launches?  An 8-way row shard of cfg3: 2 750 rows, 250 000 trials, cohort 10 000 prepared once.   usage: exp_cfg3_graph.py [R=2750]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from neuralplda_amd import ops

R = int(sys.argv[1]) if len(sys.argv) > 1 else 2750
dev = torch.device("cuda:0")
prm, _ = bench.make_params(150, dev)
pk = ops.pack_params(*prm)
g = torch.Generator(device=dev).manual_seed(1)
xr = torch.randn(R, 512, device=dev, generator=g)
xc = torch.randn(10000, 512, device=dev, generator=g)
T = 2000000 * R // 22000
ie = torch.randint(0, R // 10, (T,), device=dev, generator=g)
it = torch.randint(R // 10, R, (T,), device=dev, generator=g)
raw = torch.randn(T, device=dev, generator=g, dtype=torch.float64)
zc, qc = ops.embed(xc, pk)
prep = ops.cohort_prepare(zc, qc, pk, topn=500)


WHICH = os.environ.get("WHICH", "all")
if WHICH == "no_embed":
    zr0, qr0 = ops.embed(xr, pk)


def step():
    if WHICH == "no_embed":
        return ops.asnorm_apply(raw, ie, it, ops.cohort_stats(zr0, qr0, zc, qc, pk, topn=500, prepared=prep))
    zr, qr = ops.embed(xr, pk)
    st = ops.cohort_stats(zr, qr, zc, qc, pk, topn=500, prepared=prep if WHICH != "no_prep" else None)
    if WHICH == "no_apply":
        return st
    return ops.asnorm_apply(raw, ie, it, st)


def timeit(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize(); print('  warm 20 ok', flush=True)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / n * 1e3); print('  batch ok', ts[-1], flush=True)
    return sorted(ts)[2]


ref = step().clone()
torch.cuda.synchronize(); print('one eager step ok', flush=True)
eager = timeit(step)
print('eager timing ok', eager, flush=True)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize(); print('side-stream warm-up ok', flush=True)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    out = step()
print('captured', flush=True)
gr.replay()
torch.cuda.synchronize()
print('replayed', flush=True)
assert torch.equal(out, ref); print('equal ok', flush=True)
replay = timeit(gr.replay)
print(f"R={R} T={T}: eager {eager:.4f} ms per step, one graph replay {replay:.4f} ms (same bits)")
