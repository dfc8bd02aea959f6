#!/usr/bin/env python3
"""Shader clock held under the cohort pipeline (nplda_clock_probe on a side stream next to repeated cohort_stats calls)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neuralplda_amd import _lib, models, ops

class NC:
    xvector_dim, layer1_LDA_dim, layer2_PLDA_spkfactor_dim = 512, int(os.environ.get("D", "170")), int(os.environ.get("D", "170"))
    beta, alpha, device, loss = [99.0], 15.0, "cuda", "SoftCdet"

torch.manual_seed(3)
m = models.NeuralPlda(NC()).cuda()
packed = ops.pack_params(*[p.detach() for p in m._params()])
R, M = 22000, 10000
zr, qr = ops.embed(torch.randn(R, 512, device="cuda"), packed)
zc, qc = ops.embed(torch.randn(M, 512, device="cuda"), packed)
lib = _lib.load()
for _ in range(3):
    ops.cohort_stats(zr, qr, zc, qc, packed, topn=500)
torch.cuda.synchronize()
ticks = torch.zeros(2, dtype=torch.int64, device="cuda")
side = torch.cuda.Stream()
reps = 14
for _ in range(2):
    ops.cohort_stats(zr, qr, zc, qc, packed, topn=500)
with torch.cuda.stream(side):
    _lib.check(lib.nplda_clock_probe(_lib.ptr(ticks), 8000, _lib.current_stream()), "probe")
for _ in range(reps):
    ops.cohort_stats(zr, qr, zc, qc, packed, topn=500)
torch.cuda.synchronize()
tk = ticks.cpu().numpy()
print(f"sclk under cohort_stats: {100.0 * tk[0] / tk[1]:.0f} MHz")
