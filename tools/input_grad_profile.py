#!/usr/bin/env python3
"""Input-side backward under rocprofv3 --kernel-trace: pair backward with and without dx (B = 4096 and 262 144),
embedding backward, DPlda with a trainable LDA.  Prints wall times per call (HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neuralplda_amd import models, ops


def ms(fn, reps=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


class NC:
    def __init__(self, D):
        self.xvector_dim, self.layer1_LDA_dim, self.layer2_PLDA_spkfactor_dim = 512, D, D
        self.beta, self.alpha, self.device, self.loss = [99.0, 199.0], 15.0, "cuda", "SoftCdet"


for D in (150, 170):
    torch.manual_seed(0)
    m = models.NeuralPlda(NC(D)).cuda()
    prm = [p.detach() for p in m._params()]
    packed = ops.pack_params(*prm)
    for B in (4096, 262144):
        x1, x2 = torch.randn(B, 512, device="cuda"), torch.randn(B, 512, device="cuda")
        g = torch.randn(B, device="cuda") / B
        s, saved = ops.forward_train(x1, x2, packed)
        t0 = ms(lambda: ops.backward(saved, g, packed, prm[4]))
        t1 = ms(lambda: ops.backward(saved, g, packed, prm[4], want_dx=True))
        flop_dx = 2.0 * 2 * B * 512 * D
        print(f"D={D} B={B}: backward {t0*1e3:.1f} us, with dx {t1*1e3:.1f} us (dx GEMM {(t1-t0)*1e3:.1f} us = "
              f"{flop_dx/((t1-t0)*1e-3)/1e12:.1f} TFLOP/s algorithmic)")
        z, es = ops.embed_train(x1, packed)
        gz = torch.randn(B, D, device="cuda") / B
        t2 = ms(lambda: ops.embed_backward(es, gz, packed, want_dx=True))
        print(f"D={D} N={B}: embed_train+backward(dx) {t2*1e3:.1f} us")
    d = models.DPlda(NC(D)).cuda()
    B = 2048
    x1, x2 = torch.randn(B, 512, device="cuda", requires_grad=True), torch.randn(B, 512, device="cuda", requires_grad=True)
    t = (torch.rand(B, device="cuda") < 0.2).float()

    def step():
        d.zero_grad()
        d.loss(d(x1, x2), t).backward()
    print(f"D={D} DPlda B={B} autograd step with LDA + x gradients: {ms(step, 10)*1e3:.1f} us")
    from neuralplda_amd import train
    fs = train.FusedDPldaStep(d, 1e-4, batch_size=B, graph=True, train_lda=True, want_dx=True)
    xx1, xx2 = x1.detach(), x2.detach()
    for _ in range(20):
        fs(xx1, xx2, t)
    print(f"D={D} DPlda B={B} FusedDPldaStep(train_lda, want_dx), graph replay incl. Adam: {ms(lambda: fs(xx1, xx2, t), 50)*1e3:.1f} us")
    fs.x1.copy_(xx1); fs.x2.copy_(xx2); fs.t.copy_(t)
    print(f"D={D} DPlda B={B} ... with the inputs already in the step's buffers (no staging copies): "
          f"{ms(lambda: fs(fs.x1, fs.x2, fs.t), 50)*1e3:.1f} us")
