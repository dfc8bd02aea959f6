#!/usr/bin/env python3
"""FusedTrainStep replays at B = 4096 (BASELINE cfg2) for rocprofv3 --kernel-trace: what the graph's launches cost."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neuralplda_amd import models, train


class NC:
    xvector_dim, layer1_LDA_dim, layer2_PLDA_spkfactor_dim = 512, 150, 150
    beta, alpha, device, loss = [99.0, 199.0], 15.0, "cuda", "SoftCdet"


D = int(sys.argv[1]) if len(sys.argv) > 1 else 150
graph = (sys.argv[2] != "eager") if len(sys.argv) > 2 else True
NC.layer1_LDA_dim = NC.layer2_PLDA_spkfactor_dim = D
torch.manual_seed(0)
m = models.NeuralPlda(NC()).cuda()
B = 4096
x1 = torch.randn(B, 512, device="cuda"); x2 = torch.randn(B, 512, device="cuda")
t = (torch.rand(B, device="cuda") < 0.1).float()
step = train.FusedTrainStep(m, 1e-4, weight_decay=1e-5, batch_size=B, graph=graph)
for _ in range(5):
    step(x1, x2, t)
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for _ in range(n):
    step(x1, x2, t)
torch.cuda.synchronize()
print(f"D={D} B={B} graph={graph}: {(time.perf_counter() - t0) / n * 1e3:.4f} ms/step (wall)")
