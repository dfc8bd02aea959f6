#!/usr/bin/env python3
"""Secondary kernels for rocprofv3 --kernel-trace: GaussianBackend / DPlda scoring, weighted moments (two-class
statistics pass and DPlda gradient), the detection-cost sweep."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neuralplda_amd import models, ops


class NC:
    xvector_dim, layer1_LDA_dim, layer2_PLDA_spkfactor_dim = 512, 170, 170
    beta, alpha, device, loss = [99.0, 199.0], 15.0, "cuda", "SoftCdet"


torch.manual_seed(5)
dev = torch.device("cuda")
B = 1 << 19
x1, x2 = torch.randn(B, 512, device=dev), torch.randn(B, 512, device=dev)
t = (torch.rand(B, device=dev) < 0.1).float()
gb = models.GaussianBackend(NC()).to(dev)
dp = models.DPlda(NC()).to(dev)
for prm in dp.centering_and_LDA.parameters():
    prm.requires_grad = False
opt = torch.optim.Adam([p for p in dp.parameters() if p.requires_grad], lr=1e-4, weight_decay=1e-5)
for it in range(4):
    with torch.no_grad():
        stats = gb.accumulate_statistics(x1, x2, t)          # gb paired rows + two-class moments
        gb.fit_statistics(stats)
        s = gb(x1, x2)                                        # GB scoring
        sd = dp(x1, x2)                                       # DPlda scoring
        ops.detcost_sweep(s, t, [99.0, 199.0])                # minc, reference semantics
        ops.detcost_sweep(s, t, [99.0, 199.0], exact=True, want_eer=True)
    opt.zero_grad()
    L = dp.loss(dp(x1[:2048], x2[:2048]), t[:2048])           # DPlda recipe step at B = 2048
    L.backward()
    opt.step()
torch.cuda.synchronize()
print("ok", float(L))
