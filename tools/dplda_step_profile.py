#!/usr/bin/env python3
"""FusedDPldaStep(train_lda=True, want_dx=True) at B = 2048 under rocprofv3 --kernel-trace: which launches make up the
joint fine-tune step of the DPlda head.  usage: dplda_step_profile.py [D=150]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neuralplda_amd import models, train

D = int(sys.argv[1]) if len(sys.argv) > 1 else 150


class NC:
    xvector_dim, layer1_LDA_dim, layer2_PLDA_spkfactor_dim = 512, D, D
    beta, alpha, device, loss = [99.0, 199.0], 15.0, "cuda", "SoftCdet"


torch.manual_seed(0)
d = models.DPlda(NC()).cuda()
B = 2048
x1, x2 = torch.randn(B, 512, device="cuda"), torch.randn(B, 512, device="cuda")
t = (torch.rand(B, device="cuda") < 0.2).float()
fs = train.FusedDPldaStep(d, 1e-4, batch_size=B, graph=True, train_lda=True, want_dx=True)
for _ in range(60):
    fs(x1, x2, t)
torch.cuda.synchronize()
