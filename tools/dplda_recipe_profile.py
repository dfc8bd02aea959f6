#!/usr/bin/env python3
"""The step xvector_DPlda_pytorch.py:35-43 runs (DPlda.forward -> loss -> backward -> Adam on the linear unit, LDA frozen,
512 -> 170) as train.FusedDPldaStep replays it, under rocprofv3 --kernel-trace: which launches make it up at the script's
batch (256, BCE: conf/voices_config_dplda.cfg:25-29) and at 2048.   usage: dplda_recipe_profile.py [B=256] [loss=crossentropy] [D1=170]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neuralplda_amd import models, train

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lossname = sys.argv[2] if len(sys.argv) > 2 else "crossentropy"
D1 = int(sys.argv[3]) if len(sys.argv) > 3 else 170


class NC:
    xvector_dim, layer1_LDA_dim, layer2_PLDA_spkfactor_dim = 512, D1, D1
    beta, alpha, device = [99.0, 199.0], 15.0, "cuda"
    loss = lossname


torch.manual_seed(5)
dp = models.DPlda(NC()).cuda()
for prm in dp.centering_and_LDA.parameters():
    prm.requires_grad = False
x1, x2 = torch.randn(B, 512, device="cuda"), torch.randn(B, 512, device="cuda")
t = (torch.rand(B, device="cuda") < 0.1).float()
fs = train.FusedDPldaStep(dp, 1e-4, weight_decay=1e-5, batch_size=B, graph=True)
fs(x1, x2, t)
for _ in range(200):
    fs(fs.x1, fs.x2, fs.t)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500):
    fs(fs.x1, fs.x2, fs.t)
torch.cuda.synchronize()
print(f"DPlda recipe step B={B} {lossname} D1={D1}: {(time.perf_counter() - t0) / 500 * 1e3:.4f} ms per step ({fs.launches_per_step})")
