#!/usr/bin/env python3
"""nplda_cohort_stats_f32 at cfg3 (22 000 rows x 10 000 cohort, D = 150) in a loop for ~14 s with a prepared cohort — the load
tools/power_probe.sh samples rocm-smi under.  NPLDA_COHORT_SPLIT=0 for the fp32-input form."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neuralplda_amd import models, ops


class NC:
    xvector_dim, layer1_LDA_dim, layer2_PLDA_spkfactor_dim = 512, 150, 150
    beta, alpha, device, loss = [99.0], 15.0, "cuda", "SoftCdet"


torch.manual_seed(3)
m = models.NeuralPlda(NC()).cuda()
packed = ops.pack_params(*[p.detach() for p in m._params()])
zr, qr = ops.embed(torch.randn(22000, 512, device="cuda"), packed)
zc, qc = ops.embed(torch.randn(10000, 512, device="cuda"), packed)
prep = ops.cohort_prepare(zc, qc, packed, topn=500)
t0 = time.time()
n = 0
while time.time() - t0 < 14:
    for _ in range(100):
        ops.cohort_stats(zr, qr, zc, qc, packed, topn=500, prepared=prep)
    torch.cuda.synchronize()
    n += 100
print(f"{n} calls, {(time.time() - t0) / n * 1e3:.4f} ms per call")
