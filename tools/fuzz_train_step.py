#!/usr/bin/env python3
"""Random shapes (D = 145 .. 176, B = 1 .. 4200) through the ONE-CALL training step — batch rows, table rows (the gather
folded in), bf16 rows with dL/dx — against the fp64 oracle: loss 1e-5 relative, applied flat gradient and dL/dtheta 1e-4 of
the per-tensor max-abs, dL/dx to bf16 rounding.  B <= 2048 runs the 8-pair half-tile kernel (nplda_train_fb_half.h), above
that the 16-pair kernel.   usage: fuzz_train_step.py [iterations=60] [seed=5]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import nplda_oracle as orc
from neuralplda_amd import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
for it in range(iters):
    D = int(rng.integers(145, 177))
    B = int(rng.choice([1, 2, 7, 8, 9, 15, 16, 17, 100, 1003, 2047, 2048, 2049, 4096, int(rng.integers(1, 4201))]))
    form = str(rng.choice(["batch", "rows", "dx_f32", "dx_bf16"]))
    kind = ops.LOSS_SOFTCDET if rng.random() < 0.6 else ops.LOSS_BCE
    k1, k2 = 1 / np.sqrt(512), 1 / np.sqrt(D)
    p = orc.Params(rng.uniform(-k1, k1, (D, 512)).astype(np.float32), rng.uniform(-k1, k1, D).astype(np.float32),
                   rng.uniform(-k2, k2, (D, D)).astype(np.float32), rng.uniform(-k2, k2, D).astype(np.float32),
                   rng.uniform(0, 1, D).astype(np.float32), rng.uniform(0, 1, D).astype(np.float32))
    prm = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in p.tensors()]
    packed = ops.pack_params(*prm)
    x1 = rng.standard_normal((B, 512)).astype(np.float32)
    x2 = rng.standard_normal((B, 512)).astype(np.float32)
    if form == "dx_bf16":  # the oracle sees the bf16-rounded rows
        x1 = torch.from_numpy(x1).bfloat16().float().numpy()
        x2 = torch.from_numpy(x2).bfloat16().float().numpy()
    t = (rng.random(B) < 0.3).astype(np.float32)
    if B > 1:
        t[0], t[-1] = 1.0, 0.0
    theta = ([-0.4, -0.2] if kind == ops.LOSS_SOFTCDET else [0.1])
    betas, alpha = ([99.0, 199.0], 15.0) if kind == ops.LOSS_SOFTCDET else ([], 0.0)
    ths = [torch.tensor([v], device="cuda") for v in theta]
    n, K = int(sum(q.numel() for q in prm)), len(ths)
    m, v, step = torch.zeros(n + K, device="cuda"), torch.zeros(n + K, device="cuda"), torch.zeros(2, device="cuda")
    out, lbuf = torch.zeros(n + K, device="cuda"), torch.zeros((), device="cuda")
    T = torch.from_numpy(t).cuda()
    dxs = None
    adam = (1e-3, 0.9, 0.999, 1e-8, 1e-5)
    if form == "batch":
        ws = ops.train_step_workspace(B, packed)
        ops.train_step(torch.from_numpy(x1).cuda(), torch.from_numpy(x2).cuda(), T, prm, ths, betas, alpha, kind, m, v, step, *adam,
                       packed, ws, lbuf, grad_out=out)
    elif form == "rows":
        N = 3000
        tab = rng.standard_normal((N, 512)).astype(np.float32)
        r1, r2 = rng.integers(0, N, B), rng.integers(0, N, B)
        x1, x2 = tab[r1], tab[r2]
        ws = ops.train_step_workspace(B, packed, rows=True)
        ops.train_step_rows(torch.from_numpy(tab).cuda(), torch.from_numpy(r1).cuda(), torch.from_numpy(r2).cuda(), T, prm, ths, betas,
                            alpha, kind, m, v, step, *adam, packed, ws, lbuf, grad_out=out)
    else:
        dt = torch.bfloat16 if form == "dx_bf16" else torch.float32
        X1, X2 = torch.from_numpy(x1).cuda().to(dt), torch.from_numpy(x2).cuda().to(dt)
        dxs = (torch.zeros(B, 512, device="cuda", dtype=dt), torch.zeros(B, 512, device="cuda", dtype=dt))
        ws = ops.train_step_dx_workspace(B, packed, dt == torch.bfloat16)
        ops.train_step_dx(X1, X2, T, prm, ths, betas, alpha, kind, m, v, step, *adam, packed, ws, lbuf, dxs[0], dxs[1])
        out = None
    s_ref = orc.forward(x1, x2, p, np.float64)
    if kind == ops.LOSS_SOFTCDET:
        L_ref = orc.softcdet(s_ref, t, theta, betas, alpha, np.float64)
        g_ref, dth_ref = orc.softcdet_grad(s_ref, t, theta, betas, alpha)
    else:
        L_ref = orc.crossentropy(s_ref, t, theta[0], np.float64)
        g_ref, dth_ref = orc.crossentropy_grad(s_ref, t, theta[0])
    ok_cls = 0 < t.sum() < B or kind == ops.LOSS_BCE  # SoftCdet with one class only divides by zero in the reference too
    eL = abs(lbuf.item() - L_ref) / max(abs(L_ref), 1e-30) if ok_cls else 0.0
    eg = ed = ex = 0.0
    if out is not None and ok_cls:
        ref = orc.backward(x1, x2, g_ref, p)
        got = [a.cpu().numpy() for a in ops.split_flat_grad(out[:n], 512, D, D)]
        eg = max(np.abs(a - ref[k]).max() / max(np.abs(ref[k]).max(), 1e-30) for k, a in zip(("W1", "b1", "W2", "b2", "P_sqrt", "Q"), got))
        ed = float(np.abs(out[n:].cpu().numpy() - np.atleast_1d(dth_ref)).max() / max(np.abs(dth_ref).max(), 1e-30))
    if dxs is not None and ok_cls:
        d1, d2 = orc.input_grads(x1, x2, g_ref, p)
        for got, ref in ((dxs[0], d1), (dxs[1], d2)):
            tol = (1e-4 if form == "dx_f32" else 1e-2) * np.abs(ref).max()
            ex = max(ex, float(np.abs(got.float().cpu().numpy() - ref).max() / max(tol, 1e-30)))
    print(f"D={D} B={B} {form} kind={kind}: loss {eL:.1e} grad {eg:.1e} dtheta {ed:.1e} dx(tol units) {ex:.2f}", flush=True)
    assert np.isfinite(lbuf.item()) or not ok_cls
    assert eL <= 1e-5 and eg <= 1e-4 and ed <= 2e-4 and ex <= 1.0, (D, B, form, kind)
print("ok")
