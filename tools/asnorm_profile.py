#!/usr/bin/env python3
"""AS-norm cfg3 statistics on one GPU for rocprofv3 --kernel-trace / --pmc.
usage: asnorm_profile.py [fused|spill] [D=170]   (spill = the general path: score matrix to HBM + row statistics)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neuralplda_amd import models, ops

spill = len(sys.argv) > 1 and sys.argv[1] == "spill"
D = int(sys.argv[2]) if len(sys.argv) > 2 else 170


class NC:
    xvector_dim, layer1_LDA_dim, layer2_PLDA_spkfactor_dim = 512, D, D
    beta, alpha, device, loss = [99.0], 15.0, "cuda", "SoftCdet"

torch.manual_seed(3)
m = models.NeuralPlda(NC()).cuda()
packed = ops.pack_params(*[p.detach() for p in m._params()])
R, M = 22000, 10000
zr, qr = ops.embed(torch.randn(R, 512, device="cuda"), packed)
zc, qc = ops.embed(torch.randn(M, 512, device="cuda"), packed)
for _ in range(3):
    st = ops.cohort_stats(zr, qr, zc, qc, packed, topn=500, force_spill=spill)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    st = ops.cohort_stats(zr, qr, zc, qc, packed, topn=500, force_spill=spill)
torch.cuda.synchronize()
print(f"cohort_stats ({'spill' if spill else 'fused'}, D={D}) R={R} M={M}: {(time.perf_counter()-t0)/5*1e3:.3f} ms")
