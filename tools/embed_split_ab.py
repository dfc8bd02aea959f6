#!/usr/bin/env python3
"""extract_plda_embeddings on the fp32-input kernels against the split-bf16 image (precision="bf16x3": csrc/nplda_fwd_bf16x3.h,
MODE_EMBED) at table sizes: time per call and worst error of z against the fp64 oracle on 2 048 rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from neuralplda_amd import ops
from oracle import nplda_oracle as orc

for D in (150, 170):
    prm, _ = bench.make_params(D, torch.device("cuda:0"))
    pk = {p: ops.pack_params(*prm, precision=p) for p in ("fp32", "bf16x3")}
    P = orc.Params(*[t.detach().cpu().numpy() for t in prm])
    for n in (22000, 32000, 100000, 262144):
        x = torch.randn(n, 512, device="cuda")
        ref = orc.extract_plda_embeddings(x[:2048].cpu().numpy(), P, np.float64)
        out = []
        for p in ("fp32", "bf16x3"):
            ms, _ = bench.kernel_ms_of(lambda: ops.embed(x, pk[p]), reps=20, batches=3, warm=3)
            z, q = ops.embed(x[:2048].contiguous(), pk[p])
            zz = z[:, :D].cpu().numpy().astype(np.float64)
            out.append((ms, np.abs(zz - ref).max()))
        fl = 2.0 * (512 * D + D * D) * n
        print(f"D={D} rows={n:7d}: fp32 {out[0][0]*1e3:8.1f} us (frac {fl/(out[0][0]*1e-3)/1e12/157.3:.3f}, max|dz| {out[0][1]:.2e})   "
              f"bf16x3 {out[1][0]*1e3:8.1f} us (x{out[0][0]/out[1][0]:.2f}, max|dz| {out[1][1]:.2e})")
