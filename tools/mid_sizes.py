#!/usr/bin/env python3
"""Scoring time in steps of 1024 pairs between 8 192 and 20 480 (D = 150): the staircase of one 16-pair unit per CU
(4 096 pairs per step of ~12 us).  NPLDA_FWD_NO_MID=1 shows the small / streaming kernels at the same sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from neuralplda_amd import _lib, ops
D=150; dev=torch.device("cuda:0")
prm,_=bench.make_params(D,dev); pk=ops.pack_params(*prm)
f=bench.algorithmic_flops_per_pair(512,D,D)
for B in (8192, 9216, 10240, 11264, 12288, 14336, 16384, 18432, 20480):
    x1=torch.randn(B,512,device=dev); x2=torch.randn(B,512,device=dev)
    ms,_=bench.kernel_ms_of(lambda: ops.score_pairs(x1,x2,pk), reps=20)
    name=_lib.load().nplda_score_pairs_kernel_name(B,512,D,D).decode().split(" ")[0]
    print(f"B={B:6d}: {ms*1e3:7.1f} us frac {B*f/(ms*1e-3)/1e12/157.3:.3f} {name}", flush=True)
