#!/usr/bin/env python3
"""Score 1 048 576 pairs in a loop for ~14 s (argv[1] = fp32 | bf16x3): the load tools/power_probe.sh samples rocm-smi under."""
import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from neuralplda_amd import ops
prec = sys.argv[1]
prm, _ = bench.make_params(150, torch.device("cuda:0"))
pk = ops.pack_params(*prm, precision=prec)
B = 1048576
x1 = torch.randn(B, 512, device="cuda"); x2 = torch.randn(B, 512, device="cuda")
t0 = time.time()
while time.time() - t0 < 14:
    for _ in range(50):
        ops.score_pairs(x1, x2, pk)
    torch.cuda.synchronize()
