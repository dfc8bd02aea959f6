#!/usr/bin/env python3
"""forward(train) + backward over the minibatch AND the streaming sizes: time per call (HIP events) and fraction of the
fp32 MFMA peak on the backward's algorithmic work — weight gradients of both layers (two rows per pair) + the layer-2 data
gradient dy = dz . W2: 488 400 FLOP/pair at D = 150, 580 720 at D = 170.  Also usable under `rocprofv3 --kernel-trace` +
`rocpd_summary.py --by-grid --drop-first` to separate the fixed cost of the small-batch kernels from their per-row cost.
usage: backward_sizes.py [D=150] [reps=10] [B ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from neuralplda_amd import ops

D = int(sys.argv[1]) if len(sys.argv) > 1 else 150
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
sizes = [int(v) for v in sys.argv[3:]] or [512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144]
prm, _ = bench.make_params(D, torch.device("cuda:0"))
packed = ops.pack_params(*prm)
flop = 2 * (2 * 512 * D + 2 * D * D) + 2 * (2 * D * D)
for B in sizes:
    x1 = torch.randn(B, 512, device="cuda"); x2 = torch.randn(B, 512, device="cuda")
    g = torch.randn(B, device="cuda") * 1e-3
    s, saved = ops.forward_train(x1, x2, packed)
    t_end = time.perf_counter() + 0.05  # >= 50 ms of warm-up: the first launches after idle run at a lower clock
    while time.perf_counter() < t_end:
        flat = ops.backward(saved, g, packed, prm[4])
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        flat = ops.backward(saved, g, packed, prm[4])
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"D={D} B={B:7d} backward: {ms * 1e3:9.1f} us  {B * flop / (ms * 1e-3) / 1e12:6.1f} TFLOP/s  frac {B * flop / (ms * 1e-3) / 1e12 / 157.3:.3f}",
          flush=True)
