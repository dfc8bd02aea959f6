#!/usr/bin/env python3
"""Scoring rate against the batch size (the small-batch kernel up to 16 384 pairs, the persistent kernels above): is there
a cliff between the regimes?  usage: size_sweep.py [D=150]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from neuralplda_amd import ops

D = int(sys.argv[1]) if len(sys.argv) > 1 else 150
dev = torch.device("cuda:0")
prm, _ = bench.make_params(D, dev)
pk = ops.pack_params(*prm)
f = bench.algorithmic_flops_per_pair(512, D, D)
for B in (4096, 8192, 16384, 16385, 20000, 32768, 65536, 131072, 262144, 524288, 1048576):
    x1 = torch.randn(B, 512, device=dev); x2 = torch.randn(B, 512, device=dev)
    ms, _ = bench.kernel_ms_of(lambda: ops.score_pairs(x1, x2, pk), reps=20)
    print(f"D={D} B={B:8d}: {ms * 1e3:9.1f} us  {B / ms * 1e3:.3e} pairs/s  frac {B * f / (ms * 1e-3) / 1e12 / 157.3:.3f}", flush=True)
