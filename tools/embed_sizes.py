#!/usr/bin/env python3
"""extract_plda_embeddings rate against the row count (balanced-tile kernel vs the streaming schedule: NPLDA_FWD_NO_MID=1 forces the latter)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from neuralplda_amd import ops
dev = torch.device("cuda:0")
for D in (150, 170):
    prm, _ = bench.make_params(D, dev)
    pk = ops.pack_params(*prm)
    f = 2 * 512 * D + 2 * D * D
    for N in (22000, 50000, 100000, 200000, 400000, 1200000):
        x = torch.randn(N, 512, device=dev)
        ms, _ = bench.kernel_ms_of(lambda: ops.embed(x, pk), reps=10)
        print(f"NO_MID={os.environ.get('NPLDA_FWD_NO_MID','0')} D={D} N={N:8d}: {ms*1e3:9.1f} us  {N/ms*1e3:.3e} rows/s  frac {N*f/(ms*1e-3)/1e12/157.3:.3f}", flush=True)
        del x
