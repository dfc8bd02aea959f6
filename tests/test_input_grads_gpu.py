"""GPU parity of the input-side backward (SURVEY.md f4 / BASELINE cfg 5): dL/dx of NeuralPlda.forward, the backward of
extract_plda_embeddings and forward_from_plda_embeddings on their own, DPlda with a trainable LDA, and the E2E
composition (a torch extractor under the head, utils/models.py:251-268) — against the fp64 oracle and against the
reference's own autograd (golden G11, tests/golden/make_golden_r2.py).  Everything goes through the C ABI.

Tolerance: gradients within 1e-4 of the per-tensor max-abs of the fp64 value (the reference's fp32 autograd is itself
only ~1e-2 accurate there, tests/test_oracle_golden.py::test_g11_*)."""
import os

import numpy as np
import pytest
import torch

from oracle import nplda_oracle as orc
from tests.test_train_gpu import NC, model_from, rand_params, relmax

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("D0,D1,D2,B", [(512, 150, 150, 300), (512, 170, 170, 1000), (64, 24, 20, 200),
                                        (128, 40, 100, 33), (512, 150, 150, 20037), (512, 170, 170, 17001)])
def test_backward_ex_matches_oracle(hip_lib, D0, D1, D2, B):
    """nplda_backward_ex_f32: flat parameter gradient + dx1, dx2.  The two largest cases take the streaming
    bwd_data_kernel / v2 train-mode forward (B > 16 384), which the other training tests never reach."""
    from neuralplda_amd import ops
    rng = np.random.default_rng(D1 * 7 + B)
    p = rand_params(rng, D0, D1, D2)
    x1 = rng.standard_normal((B, D0)).astype(np.float32)
    x2 = rng.standard_normal((B, D0)).astype(np.float32)
    g = (rng.standard_normal(B) / B).astype(np.float32)
    dev = [cu(a) for a in p.tensors()]
    packed = ops.pack_params(*dev)
    s, saved = ops.forward_train(cu(x1), cu(x2), packed)
    ref_s = orc.forward(x1, x2, p, np.float64)
    assert np.all(np.abs(s.cpu().numpy() - ref_s) <= 2e-5 + 1e-5 * np.abs(ref_s))
    # saved activations of the big path: y, z, rn
    z1ref, (_, y1ref, n1) = orc.extract_plda_embeddings(x1, p, np.float64, True)
    z2ref, (_, y2ref, n2) = orc.extract_plda_embeddings(x2, p, np.float64, True)
    y, z, rn = saved[3].cpu().numpy(), saved[4].cpu().numpy(), saved[5].cpu().numpy()
    np.testing.assert_allclose(y[:B, :D1], y1ref, atol=2e-6)
    np.testing.assert_allclose(y[B:, :D1], y2ref, atol=2e-6)
    np.testing.assert_allclose(z[:B, :D2], z1ref, atol=5e-6, rtol=1e-5)
    np.testing.assert_allclose(z[B:, :D2], z2ref, atol=5e-6, rtol=1e-5)
    np.testing.assert_allclose(rn[:B], 1 / n1, rtol=1e-5)
    assert np.all(y[:, D1:] == 0) and np.all(z[:, D2:] == 0)
    flat, dx1, dx2 = ops.backward(saved, cu(g), packed, dev[4], want_dx=True)
    ref = orc.backward(x1, x2, g, p)
    for name, got in zip(("W1", "b1", "W2", "b2", "P_sqrt", "Q"), ops.split_flat_grad(flat, D0, D1, D2)):
        assert relmax(got.cpu().numpy(), ref[name]) <= 1e-4, (name, relmax(got.cpu().numpy(), ref[name]))
    rx1, rx2 = orc.input_grads(x1, x2, g, p)
    assert relmax(dx1.cpu().numpy(), rx1) <= 1e-4 and relmax(dx2.cpu().numpy(), rx2) <= 1e-4
    # the flat gradient does not depend on whether dx was asked for; everything is deterministic
    flat_b = ops.backward(saved, cu(g), packed, dev[4])
    assert torch.equal(flat, flat_b)
    _, dx1b, _ = ops.backward(saved, cu(g), packed, dev[4], want_dx=True)
    assert torch.equal(dx1, dx1b)


@pytest.mark.parametrize("D0,D1,D2,N", [(512, 150, 150, 257), (512, 170, 170, 1), (64, 24, 20, 200),
                                        (512, 150, 150, 40001)])
def test_embed_backward_matches_oracle(hip_lib, D0, D1, D2, N):
    """extract_plda_embeddings with its own backward (odd row counts, one row, > 32 768 rows: the streaming kernels)."""
    from neuralplda_amd import ops
    rng = np.random.default_rng(N + D1)
    p = rand_params(rng, D0, D1, D2)
    x = rng.standard_normal((N, D0)).astype(np.float32)
    gz = (rng.standard_normal((N, D2)) / N).astype(np.float32)
    dev = [cu(a) for a in p.tensors()]
    packed = ops.pack_params(*dev)
    z, saved = ops.embed_train(cu(x), packed)
    zref = orc.extract_plda_embeddings(x, p, np.float64)
    np.testing.assert_allclose(z[:, :D2].cpu().numpy(), zref, atol=5e-6, rtol=1e-5)
    z_plain, _ = ops.embed(cu(x), packed, want_q=False)
    # plain rows of these counts take the balanced-tile kernel (round 5: also up to one 16-row half tile per CU, as lone halves)
    if D0 == 512 and 145 <= D1 <= 176 and D1 == D2 and (N > 8192 or N <= 4096):
        np.testing.assert_allclose(z.cpu().numpy(), z_plain.cpu().numpy(), atol=2e-6, rtol=1e-5)
    else:
        assert torch.equal(z, z_plain)  # the saving variant computes the same bits
    flat, dx = ops.embed_backward(saved, cu(gz), packed, want_dx=True)
    ref = orc.embed_backward(x, gz, p)
    dW1, db1, dW2, db2, dP, dQ = ops.split_flat_grad(flat, D0, D1, D2)
    for name, got in (("W1", dW1), ("b1", db1), ("W2", dW2), ("b2", db2), ("x", dx)):
        assert relmax(got.cpu().numpy(), ref[name]) <= 1e-4, (name, relmax(got.cpu().numpy(), ref[name]))
    assert float(dP.abs().max()) == 0 and float(dQ.abs().max()) == 0
    # a strided upstream gradient (columns of a wider tensor) is consumed in place
    wide = torch.zeros(N, D2 + 8, device="cuda")
    wide[:, :D2] = cu(gz)
    flat2, _ = ops.embed_backward(saved, wide[:, :D2], packed)
    assert torch.equal(flat, flat2)


@pytest.mark.parametrize("D2,B", [(150, 1000), (170, 4096), (20, 7), (192, 70000)])
def test_score_embeddings_backward(hip_lib, D2, B):
    from neuralplda_amd import ops
    rng = np.random.default_rng(D2 + B)
    z1 = rng.standard_normal((B, D2)).astype(np.float32)
    z2 = rng.standard_normal((B, D2)).astype(np.float32)
    g = (rng.standard_normal(B) / B).astype(np.float32)
    p = orc.Params(np.zeros((1, 4), np.float32), np.zeros(1, np.float32), np.zeros((D2, 1), np.float32),
                   np.zeros(D2, np.float32), rng.uniform(0, 1, D2).astype(np.float32),
                   rng.uniform(-1, 1, D2).astype(np.float32))
    dz1, dz2, dP, dQ = ops.score_embeddings_bwd(cu(z1), cu(z2), cu(p.P_sqrt), cu(p.Q), cu(g))
    ref = orc.embscore_backward(z1, z2, g, p)
    for name, got in (("z1", dz1), ("z2", dz2), ("P_sqrt", dP), ("Q", dQ)):
        assert relmax(got.cpu().numpy(), ref[name]) <= 1e-4, name
    again = ops.score_embeddings_bwd(cu(z1), cu(z2), cu(p.P_sqrt), cu(p.Q), cu(g))
    assert torch.equal(dP, again[2]) and torch.equal(dQ, again[3])


@pytest.mark.parametrize("R,K,N,mode", [(1000, 160, 512, 0), (77, 340, 340, 2), (300, 48, 64, 1), (5, 4, 4, 0),
                                        (129, 512, 172, 0)])
def test_rows_matmul(hip_lib, R, K, N, mode):
    """The resident-matrix GEMM behind every input gradient, with bias and row scale."""
    from neuralplda_amd import ops
    rng = np.random.default_rng(R + K + N)
    src = rng.standard_normal((N, K) if mode == 1 else (K, N)).astype(np.float32)
    Wm = {0: src, 1: src.T, 2: None}[mode]
    if mode == 2:
        Wm = src + src.T
    rows = rng.standard_normal((R, K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    rs = rng.standard_normal(R).astype(np.float32)
    pm = ops.pack_matrix(cu(src), mode)
    out = ops.rows_matmul(cu(rows), pm, bias=cu(bias), rowscale=cu(rs)).cpu().numpy()
    ref = rs[:, None].astype(np.float64) * (rows.astype(np.float64) @ Wm.astype(np.float64) + bias)
    assert relmax(out, ref) <= 2e-6
    plain = ops.rows_matmul(cu(rows), pm).cpu().numpy()
    assert relmax(plain, rows.astype(np.float64) @ Wm.astype(np.float64)) <= 2e-6


# ---- module level, against the reference's autograd (G11) ----------------------------------------------------------

def test_module_input_grads_golden_small(hip_lib):
    g = np.load(os.path.join(G, "g11_input_grads_small.npz"))
    p = orc.Params(g["W1"], g["b1"], g["W2"], g["b2"], g["P_sqrt"], g["Q"])
    for lossname in ("SoftCdet", "crossentropy"):
        m = model_from(p, NC(64, 24, 20, loss=lossname), thetas=g["theta"], theta_xent=float(g["theta_xent"]))
        x1, x2 = cu(g["x1"]).requires_grad_(True), cu(g["x2"]).requires_grad_(True)
        L = m.loss(m(x1, x2), cu(g["t"]))
        L.backward()
        assert abs(L.item() - float(g[f"{lossname}_f64_L"])) <= 1e-5 * abs(float(g[f"{lossname}_f64_L"]))
        assert relmax(x1.grad.cpu().numpy(), g[f"{lossname}_f64_dx1"]) <= 1e-4
        assert relmax(x2.grad.cpu().numpy(), g[f"{lossname}_f64_dx2"]) <= 1e-4
        assert relmax(m.centering_and_LDA.weight.grad.cpu().numpy(), g[f"{lossname}_f64_dW1"]) <= 1e-4
        assert relmax(x1.grad.cpu().numpy(), g[f"{lossname}_f32_dx1"]) <= 2e-2  # the reference's fp32 is the noisy side
    # extract_plda_embeddings alone
    m = model_from(p, NC(64, 24, 20), thetas=g["theta"], theta_xent=float(g["theta_xent"]))
    x = cu(g["x1"]).requires_grad_(True)
    z = m.extract_plda_embeddings(x)
    np.testing.assert_allclose(z.detach().cpu().numpy(), g["embed_f64_z"], atol=5e-6, rtol=1e-5)
    (z * cu(g["Gz"])).sum().backward()
    for name, t in (("dx", x.grad), ("dW1", m.centering_and_LDA.weight.grad), ("db1", m.centering_and_LDA.bias.grad),
                    ("dW2", m.centering_and_wccn_plda.weight.grad), ("db2", m.centering_and_wccn_plda.bias.grad)):
        assert relmax(t.cpu().numpy(), g[f"embed_f64_{name}"]) <= 1e-4, name
    assert m.P_sqrt.grad is None and m.Q.grad is None
    # forward_from_plda_embeddings alone
    m.zero_grad()
    z1, z2 = cu(g["z1"]).requires_grad_(True), cu(g["z2"]).requires_grad_(True)
    s = m.forward_from_plda_embeddings(z1, z2)
    np.testing.assert_allclose(s.detach().cpu().numpy(), g["embscore_f64_s"], atol=2e-5, rtol=1e-5)
    (s * cu(g["gs"])).sum().backward()
    for name, t in (("dz1", z1.grad), ("dz2", z2.grad), ("dP_sqrt", m.P_sqrt.grad), ("dQ", m.Q.grad)):
        assert relmax(t.cpu().numpy(), g[f"embscore_f64_{name}"]) <= 1e-4, name
    # the composition extract -> extract -> from_embeddings is the same function as forward(): same gradients
    m.zero_grad()
    xa, xb = cu(g["x1"]).requires_grad_(True), cu(g["x2"]).requires_grad_(True)
    L2 = m.loss(m.forward_from_plda_embeddings(m.extract_plda_embeddings(xa), m.extract_plda_embeddings(xb)), cu(g["t"]))
    L2.backward()
    assert relmax(xa.grad.cpu().numpy(), g["SoftCdet_f64_dx1"]) <= 1e-4
    assert relmax(m.centering_and_LDA.weight.grad.cpu().numpy(), g["SoftCdet_f64_dW1"]) <= 1e-4


def test_module_input_grads_golden_kaldi170(hip_lib):
    from tests.test_scorefile_gpu import kaldi_model
    g = np.load(os.path.join(G, "g11_input_grads_kaldi170.npz"))
    g2 = np.load(os.path.join(G, "g2_forward_kaldi170.npz"))
    g3 = np.load(os.path.join(G, "g3_loss_kaldi170.npz"))
    m, _ = kaldi_model()
    with torch.no_grad():
        m.threshold[99.0].fill_(-0.9)
        m.threshold[199.0].fill_(-0.8)
    m.lossfn = "SoftCdet"
    x1, x2 = cu(g2["x1"]).requires_grad_(True), cu(g2["x2"]).requires_grad_(True)
    L = m.loss(m(x1, x2), cu(g3["t"]))
    L.backward()
    assert abs(L.item() - float(g["L64"])) <= 1e-5 * abs(float(g["L64"]))
    assert relmax(x1.grad.cpu().numpy(), g["dx1_64"]) <= 1e-4
    assert relmax(x2.grad.cpu().numpy(), g["dx2_64"]) <= 1e-4
    assert relmax(x1.grad.cpu().numpy(), g["dx1"]) <= 2e-2


def test_dplda_input_and_lda_grads_golden(hip_lib):
    from tests.test_dplda_gpu import make
    g = np.load(os.path.join(G, "g11_dplda_input_grads.npz"))
    for lossname in ("SoftCdet", "crossentropy"):
        m = make(64, 24, g["W1"], g["b1"], g["wlr"], g["blr"])
        m.lossfn = lossname
        with torch.no_grad():
            m.threshold[99.0].fill_(float(g["theta"][0]))
            m.threshold[199.0].fill_(float(g["theta"][1]))
        x1, x2 = cu(g["x1"]).requires_grad_(True), cu(g["x2"]).requires_grad_(True)
        L = m.loss(m(x1, x2), cu(g["t"]))
        L.backward()
        assert abs(L.item() - float(g[f"{lossname}_f64_L"])) <= 2e-5 * abs(float(g[f"{lossname}_f64_L"]))
        for name, t in (("dx1", x1.grad), ("dx2", x2.grad), ("dW1", m.centering_and_LDA.weight.grad),
                        ("db1", m.centering_and_LDA.bias.grad), ("dwlr", m.logistic_regres.weight.grad),
                        ("dblr", m.logistic_regres.bias.grad)):
            assert relmax(t.cpu().numpy(), g[f"{lossname}_f64_{name}"]) <= 2e-4, (lossname, name)


@pytest.mark.parametrize("D1,B", [(170, 3000), (40, 20000)])
def test_dplda_lda_backward_matches_oracle(hip_lib, D1, B):
    from tests.test_dplda_gpu import make
    rg = np.random.default_rng(D1 + B)
    D0 = 512 if D1 == 170 else 128
    W1 = (rg.standard_normal((D1, D0)) / np.sqrt(D0)).astype(np.float32)
    b1 = (0.1 * rg.standard_normal(D1)).astype(np.float32)
    wlr = (0.05 * rg.standard_normal((1, 2 * D1 * D1 + D1))).astype(np.float32)
    m = make(D0, D1, W1, b1, wlr, [0.1])
    x1n, x2n = rg.standard_normal((B, D0)).astype(np.float32), rg.standard_normal((B, D0)).astype(np.float32)
    gs = (rg.standard_normal(B) / B).astype(np.float32)
    x1, x2 = cu(x1n).requires_grad_(True), cu(x2n).requires_grad_(True)
    (m(x1, x2) * cu(gs)).sum().backward()
    ref = orc.dplda_lda_backward(x1n, x2n, gs, W1, b1, wlr)
    for name, t in (("x1", x1.grad), ("x2", x2.grad), ("W1", m.centering_and_LDA.weight.grad),
                    ("b1", m.centering_and_LDA.bias.grad)):
        assert relmax(t.cpu().numpy(), ref[name]) <= 2e-4, name


# ---- behaviour ----------------------------------------------------------------------------------------------------

def test_double_backward_and_cpu_inputs(hip_lib):
    """retain_graph + a second backward works (activations live in save_for_backward, not on the ctx), an in-place edit of
    a saved input is detected, and CPU leaf inputs get CPU gradients."""
    rng = np.random.default_rng(5)
    p = rand_params(rng, 64, 24, 20)
    m = model_from(p, NC(64, 24, 20))
    x1, x2 = cu(rng.standard_normal((40, 64)).astype(np.float32)).requires_grad_(True), \
        cu(rng.standard_normal((40, 64)).astype(np.float32))
    t = cu((rng.random(40) < 0.3).astype(np.float32))
    L = m.loss(m(x1, x2), t)
    L.backward(retain_graph=True)
    g1 = x1.grad.clone()
    L.backward()
    assert torch.allclose(x1.grad, 2 * g1)
    s = m(x1, x2)
    with torch.no_grad():
        x2.add_(1.0)
    with pytest.raises(RuntimeError):
        s.sum().backward()
    xc = torch.from_numpy(rng.standard_normal((40, 64)).astype(np.float32)).requires_grad_(True)
    m(xc, xc.detach()).sum().backward()
    assert xc.grad is not None and xc.grad.device.type == "cpu" and torch.isfinite(xc.grad).all()


def test_bf16_inputs_and_e2e_extractor(hip_lib):
    """BASELINE cfg 5 in miniature: a torch 'extractor' (autocast bf16) under the HIP head, joint backward.  The head
    computes in fp32 whatever the input dtype; its input gradient comes back in the input's dtype and — like the
    extractor's weight gradients — equals what the head restated in torch FLOAT64 ops gives on the same bf16 activations
    (float64, because torch's fp32 sigma' = sigma (1 - sigma) loses its digits where the sigmoid saturates)."""
    rng = np.random.default_rng(6)
    D0, D1, D2, B, F = 512, 150, 150, 512, 96
    p = rand_params(rng, D0, D1, D2)
    m = model_from(p, NC(D0, D1, D2))
    ext = torch.nn.Sequential(torch.nn.Linear(F, 256), torch.nn.ReLU(), torch.nn.Linear(256, D0)).cuda()
    f1, f2 = torch.randn(B, F, device="cuda"), torch.randn(B, F, device="cuda")
    t = (torch.rand(B, device="cuda") < 0.2).float()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        s0 = m(ext(f1), ext(f2))
    th = [float(torch.quantile(s0, 0.5)), float(torch.quantile(s0, 0.6))]  # thresholds inside the score range
    with torch.no_grad():
        m.threshold[99.0].fill_(th[0])
        m.threshold[199.0].fill_(th[1])

    def head_torch(x1, x2):
        W1, b1, W2, b2, Ps, Q = [q.detach().double() for q in m._params()]
        z1 = torch.nn.functional.normalize(x1.double() @ W1.T + b1) @ W2.T + b2
        z2 = torch.nn.functional.normalize(x2.double() @ W1.T + b1) @ W2.T + b2
        s = (z1 * Q * z1).sum(1) + (z2 * Q * z2).sum(1) + 2 * (z1 * Ps * Ps * z2).sum(1)
        sig, td = torch.sigmoid, t.double()
        return sum((sig(15.0 * (h - s)) * td).sum() / td.sum() + b * (sig(15.0 * (s - h)) * (1 - td)).sum() / (1 - td).sum()
                   for h, b in zip((float(np.float32(th[0])), float(np.float32(th[1]))), (99.0, 199.0))) / 2

    res = []
    for head in ("hip", "torch"):
        ext.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            x1, x2 = ext(f1), ext(f2)
        assert x1.dtype == torch.bfloat16
        x1.retain_grad()
        L = m.loss(m(x1, x2), t) if head == "hip" else head_torch(x1, x2)
        L.backward()
        assert x1.grad.dtype == torch.bfloat16
        res.append((float(L.detach()), x1.grad.float().clone(), ext[0].weight.grad.clone(), ext[2].weight.grad.clone()))
    assert abs(res[0][0] - res[1][0]) <= 1e-4 * abs(res[1][0])
    dxa, dxb = res[0][1], res[1][1]
    # dL/dx in bf16: the two fp32/fp64 values round to the same or to neighbouring bf16 numbers (2^-8 relative)
    assert float((dxa - dxb).abs().max()) <= 2.0 ** -7 * float(dxb.abs().max())
    for a, b in zip(res[0][2:], res[1][2:]):  # extractor gradients: bf16 GEMMs of those rows
        assert float((a - b).abs().max()) <= 2e-2 * float(b.abs().max())
    # the head's own parameters received gradients in the joint step
    assert m.centering_and_LDA.weight.grad is not None and torch.isfinite(m.centering_and_LDA.weight.grad).all()


def test_packed_image_cache_follows_parameter_updates(hip_lib):
    """The packed parameter image is cached per parameter version: unchanged parameters reuse it, an optimizer step, a
    load_state_dict, the Kaldi loader and the fused training step all invalidate it."""
    from neuralplda_amd import train
    rng = np.random.default_rng(7)
    p = rand_params(rng, 512, 150, 150)
    m = model_from(p, NC(512, 150, 150))
    x1, x2 = torch.randn(300, 512, device="cuda"), torch.randn(300, 512, device="cuda")
    with torch.no_grad():
        s0 = m(x1, x2)
        buf0 = m._pack_cache["packed"].buf
        s0b = m(x1, x2)
        assert m._pack_cache["packed"].buf is buf0 and torch.equal(s0, s0b)
    opt = torch.optim.SGD(m.parameters(), lr=0.05)
    t = (torch.rand(300, device="cuda") < 0.3).float()
    m.loss(m(x1, x2), t).backward()
    opt.step()
    with torch.no_grad():
        s1 = m(x1, x2)
    pn = orc.Params(*[q.detach().cpu().numpy() for q in m._params()])
    ref = orc.forward(x1.cpu().numpy(), x2.cpu().numpy(), pn, np.float64)
    assert np.all(np.abs(s1.cpu().numpy() - ref) <= 2e-5 + 1e-5 * np.abs(ref)) and not torch.equal(s0, s1)
    # fused step (raw Adam kernel): the next forward must see the new parameters
    step = train.FusedTrainStep(m, 1e-2, batch_size=300, graph=True)
    for _ in range(3):
        step(x1, x2, t)
    with torch.no_grad():
        s2 = m(x1, x2)
    pn = orc.Params(*[q.detach().cpu().numpy() for q in m._params()])
    ref = orc.forward(x1.cpu().numpy(), x2.cpu().numpy(), pn, np.float64)
    assert np.all(np.abs(s2.cpu().numpy() - ref) <= 2e-5 + 1e-5 * np.abs(ref)) and not torch.equal(s1, s2)
    # load_state_dict back to the start
    m.load_state_dict(model_from(p, NC(512, 150, 150)).state_dict())
    with torch.no_grad():
        assert torch.equal(m(x1, x2), s0)


def test_fused_dplda_step_crossentropy(hip_lib):
    """FusedDPldaStep under BCE (DPlda has no threshold_Xent: utils/models.py:503-506) follows autograd + torch Adam."""
    from neuralplda_amd import train
    from tests.test_dplda_gpu import _freeze_lda, make
    rg = np.random.default_rng(9)
    D0, D1, B = 128, 40, 512
    W1 = (rg.standard_normal((D1, D0)) / np.sqrt(D0)).astype(np.float32)
    b1 = (0.1 * rg.standard_normal(D1)).astype(np.float32)
    wlr = (0.05 * rg.standard_normal((1, 2 * D1 * D1 + D1))).astype(np.float32)
    batches = [(cu(rg.standard_normal((B, D0)).astype(np.float32)), cu(rg.standard_normal((B, D0)).astype(np.float32)),
                cu((rg.random(B) < 0.2).astype(np.float32))) for _ in range(4)]
    m_ref = make(D0, D1, W1, b1, wlr, [0.0])
    m_ref.lossfn = "crossentropy"
    _freeze_lda(m_ref)
    trainable = [m_ref.logistic_regres.weight, m_ref.logistic_regres.bias]
    opt = torch.optim.Adam(trainable, lr=1e-3, weight_decay=1e-5)
    ref_losses = []
    for x1, x2, t in batches:
        opt.zero_grad()
        L = m_ref.loss(m_ref(x1, x2), t)
        L.backward()
        opt.step()
        ref_losses.append(float(L))
    for graph in (False, True):
        m = make(D0, D1, W1, b1, wlr, [0.0])
        m.lossfn = "crossentropy"
        _freeze_lda(m)
        step = train.FusedDPldaStep(m, 1e-3, weight_decay=1e-5, batch_size=B, graph=graph)
        losses = [float(step(x1, x2, t)) for x1, x2, t in batches]
        np.testing.assert_allclose(losses, ref_losses, rtol=2e-5)
        np.testing.assert_allclose(m.logistic_regres.weight.detach().cpu().numpy(),
                                   m_ref.logistic_regres.weight.detach().cpu().numpy(), rtol=2e-4, atol=2e-6)
        for b in m.beta:  # thresholds do not train under BCE
            assert float(m.threshold[b]) == 0.0


def test_head_step_with_input_grads_follows_autograd(hip_lib):
    """train.HeadStepWithInputGrads (BASELINE cfg 5, the head's share): bf16 x-vectors in, the head's Adam step, dL/dx back
    in bf16 — against loss.backward() + torch.optim.Adam on the same model with x.float() requiring grad."""
    from neuralplda_amd import train
    from tests.test_train_gpu import NC, model_from, rand_params
    rng = np.random.default_rng(77)
    D, B = 150, 1024
    p = rand_params(rng, 512, D, D)
    batches = [(torch.from_numpy(rng.standard_normal((B, 512)).astype(np.float32)).cuda().to(torch.bfloat16),
                torch.from_numpy(rng.standard_normal((B, 512)).astype(np.float32)).cuda().to(torch.bfloat16),
                torch.from_numpy((rng.random(B) < 0.2).astype(np.float32)).cuda()) for _ in range(3)]
    for graph in (False, True):
        nc = NC(512, D, D)
        m_a, m_b = model_from(p, nc, thetas=[-0.5, -0.3]), model_from(p, nc, thetas=[-0.5, -0.3])
        step = train.HeadStepWithInputGrads(m_a, 1e-3, weight_decay=1e-5, batch_size=B, graph=graph)
        opt = train.make_optimizer(m_b, 1e-3)
        for x1, x2, t in batches:
            loss, dx1, dx2 = step(x1, x2, t)
            a1, a2 = x1.float().requires_grad_(True), x2.float().requires_grad_(True)
            opt.zero_grad()
            L = m_b.loss(m_b(a1, a2), t)
            L.backward()
            opt.step()
            assert dx1.dtype == torch.bfloat16 and dx1.shape == (B, 512)
            assert abs(loss.item() - L.item()) <= 1e-5 * abs(L.item())
            for got, want in ((dx1, a1.grad), (dx2, a2.grad)):
                assert float((got.float() - want).abs().max()) <= 2.0 ** -7 * float(want.abs().max())
        for (k, a), (_, b) in zip(m_a.state_dict().items(), m_b.state_dict().items()):
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-4, atol=2e-6, err_msg=k)


def test_fused_dplda_step_with_lda_and_input_grads(hip_lib):
    """FusedDPldaStep(train_lda=True, want_dx=True): the joint fine-tune step of the DPlda head (linear unit + LDA + dL/dx)
    as one captured set of direct launches, against loss.backward() + torch.optim.Adam on the same model."""
    from neuralplda_amd import models, train
    from tests.test_train_gpu import NC
    rng = np.random.default_rng(5)
    D, B = 150, 512
    batches = [(torch.from_numpy(rng.standard_normal((B, 512)).astype(np.float32)).cuda(),
                torch.from_numpy(rng.standard_normal((B, 512)).astype(np.float32)).cuda(),
                torch.from_numpy((rng.random(B) < 0.2).astype(np.float32)).cuda()) for _ in range(3)]
    for graph in (False, True):
        torch.manual_seed(3)
        m_a = models.DPlda(NC(512, D, D)).cuda()
        m_b = models.DPlda(NC(512, D, D)).cuda()
        m_b.load_state_dict(m_a.state_dict())
        step = train.FusedDPldaStep(m_a, 1e-3, weight_decay=1e-5, batch_size=B, graph=graph, train_lda=True, want_dx=True)
        opt = train.make_optimizer(m_b, 1e-3)
        for x1, x2, t in batches:
            loss, dx1, dx2 = step(x1, x2, t)
            a1, a2 = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
            opt.zero_grad()
            L = m_b.loss(m_b(a1, a2), t)
            L.backward()
            opt.step()
            assert abs(loss.item() - L.item()) <= 1e-5 * abs(L.item())
            for got, want in ((dx1, a1.grad), (dx2, a2.grad)):
                assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max())
        for (k, a), (_, b) in zip(m_a.state_dict().items(), m_b.state_dict().items()):
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-4, atol=2e-6, err_msg=k)


def test_head_step_data_parallel_form(hip_lib):
    """HeadStepWithInputGrads on a data-parallel model (the head under an extractor's DDP, BASELINE configs[4]): one rank
    with identity reductions = the plain fused head step, bit for bit in parameters and dL/dx (eager and graph replay, bf16
    and fp32 rows, incl. the separate-launch fall-back for an unsupported shape); two unequal shards of one minibatch with
    the GLOBAL counts give the dL/dx rows of the whole-batch step."""
    from neuralplda_amd import ops, train
    from tests.test_train_gpu import NC, model_from, rand_params
    rng = np.random.default_rng(123)
    D, B = 150, 1024
    p = rand_params(rng, 512, D, D)
    mk = lambda dt: [(torch.from_numpy(rng.standard_normal((B, 512)).astype(np.float32)).cuda().to(dt),
                      torch.from_numpy(rng.standard_normal((B, 512)).astype(np.float32)).cuda().to(dt),
                      torch.from_numpy((rng.random(B) < 0.2).astype(np.float32)).cuda()) for _ in range(3)]
    for dt in (torch.bfloat16, torch.float32):
        batches = mk(dt)
        for graph in (False, True):
            nc = NC(512, D, D)
            m_a, m_b = model_from(p, nc, thetas=[-0.5, -0.3]), model_from(p, nc, thetas=[-0.5, -0.3])
            calls = []
            m_b._reduce_sums = lambda t: t
            m_b._reduce_flat = lambda t: (calls.append(t.numel()), t)[1]
            sa = train.HeadStepWithInputGrads(m_a, 1e-3, weight_decay=1e-5, batch_size=B, graph=graph, dtype=dt)
            sb = train.HeadStepWithInputGrads(m_b, 1e-3, weight_decay=1e-5, batch_size=B, graph=graph, dtype=dt)
            assert sb._dp_head and not sb._fused_ok(batches[0][0]) and sb._dp_ok(batches[0][0])
            for i, (x1, x2, t) in enumerate(batches):
                nt = float(t.sum().item())
                la, da1, da2 = sa(x1, x2, t)
                lb, db1, db2 = sb(x1, x2, t, global_counts=(nt, B - nt)) if i else sb(x1, x2, t)
                assert abs(la.item() - lb.item()) <= 1e-6 * abs(la.item())
                assert torch.equal(da1, db1) and torch.equal(da2, db2)
                for (k, a), (_, b) in zip(m_a.state_dict().items(), m_b.state_dict().items()):
                    if k.startswith("Th"):
                        assert torch.allclose(a, b, rtol=1e-6, atol=1e-9), (dt, graph, i, k)
                    else:
                        assert torch.equal(a, b), (dt, graph, i, k)
            assert calls and all(n == ops.train_step_flat_floats(sb._packed) for n in calls)
    # the fall-back (40-dimensional head: no fused form for bf16 rows) also goes through the two reductions
    nc = NC(512, 40, 40)
    p40 = rand_params(rng, 512, 40, 40)
    m_a, m_b = model_from(p40, nc, thetas=[-0.5, -0.3]), model_from(p40, nc, thetas=[-0.5, -0.3])
    seen = []
    m_b._reduce_sums = lambda t: (seen.append("sums"), t)[1]
    m_b._reduce_flat = lambda t: (seen.append("flat"), t)[1]
    x1, x2, t = mk(torch.bfloat16)[0]
    sa = train.HeadStepWithInputGrads(m_a, 1e-3, batch_size=B, graph=False)
    sb = train.HeadStepWithInputGrads(m_b, 1e-3, batch_size=B, graph=False)
    assert not sb._dp_ok(x1)
    la, da1, _ = sa(x1, x2, t)
    lb, db1, _ = sb(x1, x2, t)
    assert seen == ["sums", "flat"] and abs(la.item() - lb.item()) <= 1e-6 * abs(la.item())
    assert float((da1.float() - db1.float()).abs().max()) <= 2.0 ** -7 * float(da1.float().abs().max())
    # two shards, global counts: each shard's dL/dx = its rows of the whole-batch dL/dx
    nc = NC(512, D, D)
    x1, x2, t = mk(torch.float32)[0]
    m_a, m_b = model_from(p, nc, thetas=[-0.5, -0.3]), model_from(p, nc, thetas=[-0.5, -0.3])
    sa = train.HeadStepWithInputGrads(m_a, 1e-3, batch_size=B, graph=False, dtype=torch.float32)
    _, da1, da2 = sa(x1, x2, t)
    m_b._reduce_sums = lambda v: v
    m_b._reduce_flat = lambda v: v
    nt = float(t.sum().item())
    for lo, hi in ((0, 300), (300, B)):
        m_c = model_from(p, nc, thetas=[-0.5, -0.3])
        m_c._reduce_sums, m_c._reduce_flat = m_b._reduce_sums, m_b._reduce_flat
        sc = train.HeadStepWithInputGrads(m_c, 1e-3, batch_size=hi - lo, graph=False, dtype=torch.float32)
        _, dc1, dc2 = sc(x1[lo:hi].contiguous(), x2[lo:hi].contiguous(), t[lo:hi].contiguous(), global_counts=(nt, B - nt))
        for got, want in ((dc1, da1[lo:hi]), (dc2, da2[lo:hi])):
            assert float((got - want).abs().max()) <= 2e-6 * float(want.abs().max()) + 1e-12
