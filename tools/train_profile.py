#!/usr/bin/env python3
"""Eager training steps at B = 4096 (BASELINE cfg2) for rocprofv3 --kernel-trace: per-kernel breakdown."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neuralplda_amd import models

class NC:
    xvector_dim, layer1_LDA_dim, layer2_PLDA_spkfactor_dim = 512, 150, 150
    beta, alpha, device, loss = [99.0, 199.0], 15.0, "cuda", "SoftCdet"

D = int(sys.argv[1]) if len(sys.argv) > 1 else 150
NC.layer1_LDA_dim = NC.layer2_PLDA_spkfactor_dim = D
torch.manual_seed(0)
m = models.NeuralPlda(NC()).cuda()
B = 4096
x1 = torch.randn(B, 512, device="cuda"); x2 = torch.randn(B, 512, device="cuda")
t = (torch.rand(B, device="cuda") < 0.1).float()
opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=1e-5)
def step():
    opt.zero_grad()
    o = m(x1, x2); L = m.loss(o, t); L.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 50
for _ in range(n): step()
torch.cuda.synchronize()
print(f"D={D} B={B}: {(time.perf_counter()-t0)/n*1e3:.3f} ms/step (eager, wall)")

from neuralplda_amd import train
torch.manual_seed(0)
m2 = models.NeuralPlda(NC()).cuda()
opt2 = train.make_optimizer(m2, 1e-4, capturable=True)
gs = train.GraphedTrainStep(m2, opt2, B)
for _ in range(5): gs(x1, x2, t)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n): gs(x1, x2, t)
torch.cuda.synchronize()
print(f"D={D} B={B}: {(time.perf_counter()-t0)/n*1e3:.3f} ms/step (HIP graph replay, wall, incl. batch copy-in) -> {B*n/(time.perf_counter()-t0):.3e} pairs/s")

torch.manual_seed(0)
m3 = models.NeuralPlda(NC()).cuda()
fs = train.FusedTrainStep(m3, 1e-4, batch_size=B, graph=True)
for _ in range(5): fs(x1, x2, t)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n): fs(x1, x2, t)
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"D={D} B={B}: {el/n*1e3:.3f} ms/step (FusedTrainStep, HIP graph, wall, incl. batch copy-in) -> {B*n/el:.3e} pairs/s")
fe = train.FusedTrainStep(m3, 1e-4, batch_size=B, graph=False)
for _ in range(5): fe(x1, x2, t)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n): fe(x1, x2, t)
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"D={D} B={B}: {el/n*1e3:.3f} ms/step (FusedTrainStep, eager launches) -> {B*n/el:.3e} pairs/s")
