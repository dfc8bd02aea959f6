#!/usr/bin/env python3
"""forward(train) + backward at streaming batch sizes (the large-batch kernels: bwd_data_kernel, wgrad, reduce), for
`rocprofv3 --kernel-trace` / `--pmc` + `rocpd_summary.py --by-grid --drop-first`, and as a stand-alone timing:
prints the backward's time and its fraction of the fp32 MFMA peak on 488 400 FLOP/pair (D = 150; weight gradients
398 400 + layer-2 data gradient 90 000) / 580 720 (D = 170).   usage: backward_large.py [D=150] [reps=10] [B ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from neuralplda_amd import ops

D = int(sys.argv[1]) if len(sys.argv) > 1 else 150
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
sizes = [int(v) for v in sys.argv[3:]] or [32768, 65536, 131072, 262144]
prm, _ = bench.make_params(D, torch.device("cuda:0"))
packed = ops.pack_params(*prm)
flop = 2 * (2 * 512 * D + 2 * D * D) + 2 * (2 * D * D)  # wgrad of both layers (two rows per pair) + dy = dz . W2
for B in sizes:
    x1 = torch.randn(B, 512, device="cuda"); x2 = torch.randn(B, 512, device="cuda")
    g = torch.randn(B, device="cuda") * 1e-3
    s, saved = ops.forward_train(x1, x2, packed)
    for want_dx in (False, True):
        t_end = __import__("time").perf_counter() + 0.05  # >= 50 ms of warm-up: the first launches after idle run at a lower clock
        while __import__("time").perf_counter() < t_end:
            out = ops.backward(saved, g, packed, prm[4], want_dx=want_dx)
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            out = ops.backward(saved, g, packed, prm[4], want_dx=want_dx)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        f = flop + (2 * 2 * 512 * D if want_dx else 0)
        print(f"D={D} B={B:7d} backward{' + dx' if want_dx else '     '}: {ms * 1e3:9.1f} us  {B * f / (ms * 1e-3) / 1e12:6.1f} TFLOP/s  "
              f"frac {B * f / (ms * 1e-3) / 1e12 / 157.3:.3f}", flush=True)
