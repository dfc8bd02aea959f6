#!/usr/bin/env python3
"""Scoring rate against the batch size (small-batch kernel up to one tile per CU, balanced-tile kernel, streaming kernels:
csrc/nplda_fwd_dispatch.h picks by modelled time): is there a cliff between the regimes?  usage: size_sweep.py [D=150]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from neuralplda_amd import _lib, ops

D = int(sys.argv[1]) if len(sys.argv) > 1 else 150
dev = torch.device("cuda:0")
prm, _ = bench.make_params(D, dev)
pk = ops.pack_params(*prm)
f = bench.algorithmic_flops_per_pair(512, D, D)
for B in (4096, 8192, 10240, 12288, 16384, 16385, 20000, 20480, 24577, 32768, 40000, 49152, 65536, 70000, 81920, 100000, 131072,
          150000, 200000, 262144, 300000, 524288, 1048576):
    x1 = torch.randn(B, 512, device=dev); x2 = torch.randn(B, 512, device=dev)
    ms, _ = bench.kernel_ms_of(lambda: ops.score_pairs(x1, x2, pk), reps=20)
    name = _lib.load().nplda_score_pairs_kernel_name(B, 512, D, D).decode().split(" ")[0]
    print(f"D={D} B={B:8d}: {ms * 1e3:9.1f} us  {B / ms * 1e3:.3e} pairs/s  frac {B * f / (ms * 1e-3) / 1e12 / 157.3:.3f}  {name}", flush=True)
