import os, sys, torch
sys.path.insert(0, os.getcwd())
import bench
from neuralplda_amd import ops
dev = torch.device("cuda:0")
for D in (150,):
    prm, _ = bench.make_params(D, dev)
    pk = ops.pack_params(*prm)
    f = 2 * 512 * D + 2 * D * D
    for N in (22000, 100000, 200000):
        x = torch.randn(N, 512, device=dev)
        x2 = torch.randn(N // 2, 512, device=dev)
        for name, fn in (("embed q", lambda: ops.embed(x, pk)), ("embed noq", lambda: ops.embed(x, pk, want_q=False)),
                         ("pairs N/2", lambda: ops.score_pairs(x[:N // 2], x2, pk))):
            ms, _ = bench.kernel_ms_of(fn, reps=10)
            print(f"D={D} N={N:8d} {name:10s}: {ms*1e3:9.1f} us  frac {N*f/(ms*1e-3)/1e12/157.3:.3f}", flush=True)
