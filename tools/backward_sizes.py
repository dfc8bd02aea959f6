#!/usr/bin/env python3
"""forward(train) + backward at several minibatch sizes, for `rocprofv3 --kernel-trace` + `rocpd_summary.py --by-grid`:
separates the fixed cost of the small-batch kernels from their per-row cost."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from neuralplda_amd import ops

D = int(sys.argv[1]) if len(sys.argv) > 1 else 150
prm, _ = bench.make_params(D, torch.device("cuda:0"))
packed = ops.pack_params(*prm)
for B in (512, 1024, 2048, 4096, 8192, 16384):
    x1 = torch.randn(B, 512, device="cuda"); x2 = torch.randn(B, 512, device="cuda")
    g = torch.randn(B, device="cuda") * 1e-3
    for _ in range(12):
        s, saved = ops.forward_train(x1, x2, packed)
        flat = ops.backward(saved, g, packed, prm[4])
    torch.cuda.synchronize()
