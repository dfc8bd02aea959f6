#!/usr/bin/env python3
"""GaussianBackend.forward at streaming batch sizes (the GB mode of nplda_fwd_kernel.h) — ms per call for one build of the library
(A/B: tools/ab_lib.sh).  usage: ab_gb.py [D1=170]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from neuralplda_amd import ops

D1 = int(sys.argv[1]) if len(sys.argv) > 1 else 170
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
W1 = torch.randn(D1, 512, generator=g) * 0.05; b1 = torch.randn(D1, generator=g) * 0.1
A = torch.randn(2 * D1, 2 * D1, generator=g)
Lt = A @ A.T / (2 * D1) + torch.eye(2 * D1); Ln = A.T @ A / (2 * D1) + 0.5 * torch.eye(2 * D1)
mt = torch.randn(2 * D1, generator=g) * 0.1; mn = torch.randn(2 * D1, generator=g) * 0.1
gpk = ops.gb_pack(*[t.to(dev) for t in (W1, b1, mt, Lt, mn, Ln)])
fg = 2 * 2 * 512 * D1 + 2 * (2 * D1) ** 2
for B in (65536, 131072, 524288):
    x1 = torch.randn(B, 512, device=dev); x2 = torch.randn(B, 512, device=dev)
    ms, s = bench.kernel_ms_of(lambda: ops._gb_call(x1, x2, gpk, True, False)[0], reps=8)
    print(f"D1={D1} GB forward B={B:7d}: {ms * 1e3:9.1f} us  frac {B * fg / (ms * 1e-3) / 1e12 / 157.3:.3f}  checksum {float(s.double().sum()):.6e}", flush=True)
