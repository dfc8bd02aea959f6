#!/usr/bin/env python3
"""Balanced-tile kernel (csrc/nplda_fwd_mid.h) by mode: embedding N rows, embedding rows named by index, scoring B pairs — us per
call for one build of the library (A/B: tools/ab_mid_embed.sh swaps builds on the same box).  usage: ab_mid_embed.py [D=150]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from neuralplda_amd import ops

D = int(sys.argv[1]) if len(sys.argv) > 1 else 150
dev = torch.device("cuda:0")
prm, _ = bench.make_params(D, dev)
pk = ops.pack_params(*prm)
f = bench.algorithmic_flops_per_pair(512, D, D) / 2
tab = torch.randn(1200000, 512, device=dev)
g = torch.Generator(device="cpu").manual_seed(1)
for N in (12304, 32000, 65536, 100000, 131072, 262144, 1200000):
    x = tab[:N]
    ms, _ = bench.kernel_ms_of(lambda: ops.embed(x, pk), reps=30)
    print(f"D={D} embed      N={N:7d}: {ms * 1e3:8.1f} us  frac {N * f / (ms * 1e-3) / 1e12 / 157.3:.3f}", flush=True)
for N in (32000, 100000):
    rows = torch.randint(0, tab.shape[0], (N,), generator=g).to(dev)
    ms, _ = bench.kernel_ms_of(lambda: ops.embed_rows(tab, rows, pk), reps=30)
    print(f"D={D} embed_rows N={N:7d}: {ms * 1e3:8.1f} us  frac {N * f / (ms * 1e-3) / 1e12 / 157.3:.3f}", flush=True)
for B in (10240, 20000, 32768, 40000, 49152):
    x1 = tab[:B]; x2 = tab[100000:100000 + B]
    ms, _ = bench.kernel_ms_of(lambda: ops.score_pairs(x1, x2, pk), reps=30)
    print(f"D={D} pairs      B={B:7d}: {ms * 1e3:8.1f} us  frac {B * 2 * f / (ms * 1e-3) / 1e12 / 157.3:.3f}", flush=True)
