"""GPU parity of the training path (loss kernels, hand-derived backward, module autograd) against the
CPU oracle (fp64) and the reference-generated golden vectors.  Everything goes through the C ABI.

Tolerances: loss rtol 1e-5; g / grads: 1e-4 of the per-tensor max-abs against the fp64
oracle (the reference's own fp32 autograd is ~1e-2 noisy there, see test_oracle_golden.py).
"""
import os

import numpy as np
import pytest
import torch

from oracle import nplda_oracle as orc

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class NC:
    def __init__(self, D0=512, D1=170, D2=170, beta=(99.0, 199.0), alpha=15.0, loss="SoftCdet"):
        self.xvector_dim, self.layer1_LDA_dim, self.layer2_PLDA_spkfactor_dim = D0, D1, D2
        self.beta, self.alpha, self.device, self.loss = list(beta), alpha, "cuda", loss


def rand_params(rng, D0, D1, D2):
    k1, k2 = 1 / np.sqrt(D0), 1 / np.sqrt(D1)
    return orc.Params(rng.uniform(-k1, k1, (D1, D0)).astype(np.float32), rng.uniform(-k1, k1, D1).astype(np.float32),
                      rng.uniform(-k2, k2, (D2, D1)).astype(np.float32), rng.uniform(-k2, k2, D2).astype(np.float32),
                      rng.uniform(0, 1, D2).astype(np.float32), rng.uniform(0, 1, D2).astype(np.float32))


def model_from(p, nc, thetas=None, theta_xent=None):
    from neuralplda_amd import models
    m = models.NeuralPlda(nc)
    sd = m.state_dict()
    sd["centering_and_LDA.weight"].copy_(torch.from_numpy(p.W1))
    sd["centering_and_LDA.bias"].copy_(torch.from_numpy(p.b1))
    sd["centering_and_wccn_plda.weight"].copy_(torch.from_numpy(p.W2))
    sd["centering_and_wccn_plda.bias"].copy_(torch.from_numpy(p.b2))
    sd["P_sqrt"].copy_(torch.from_numpy(p.P_sqrt))
    sd["Q"].copy_(torch.from_numpy(p.Q))
    if thetas is not None:
        for b, th in zip(nc.beta, thetas):
            sd["Th{}".format(int(b))].fill_(float(th))
    if theta_xent is not None:
        sd["threshold_Xent"].fill_(float(theta_xent))
    return m.cuda()


def relmax(a, b):
    return np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize("K", [1, 2, 3])
@pytest.mark.parametrize("B", [7, 512, 4096])
def test_softcdet_kernels(hip_lib, K, B):
    from neuralplda_amd import ops
    rng = np.random.default_rng(B + K)
    s = (rng.standard_normal(B) * 0.5 - 1).astype(np.float32)
    t = (rng.random(B) < 0.15).astype(np.float32)
    t[0], t[1] = 1, 0
    theta = [-0.8, -0.6, -1.1][:K]
    beta = [99.0, 199.0, 9.9][:K]
    alpha = 15.0
    S, T = torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()
    ths = [torch.tensor([th], dtype=torch.float32, device="cuda") for th in theta]
    sums = ops.loss_sums(S, T, ths, alpha, ops.LOSS_SOFTCDET)
    loss, g, dth = ops.loss_finish(S, T, ths, beta, alpha, ops.LOSS_SOFTCDET, sums)
    th32 = [float(np.float32(x)) for x in theta]
    Lref = orc.softcdet(s, t, th32, beta, alpha, np.float64)
    gref, dthref = orc.softcdet_grad(s, t, th32, beta, alpha)
    assert abs(loss.item() - Lref) <= 1e-5 * abs(Lref)
    assert relmax(g.cpu().numpy(), gref) <= 1e-4
    assert relmax(dth.cpu().numpy(), dthref) <= 1e-4
    assert abs(sums[0].item() - t.sum()) == 0 and abs(sums[1].item() - (1 - t).sum()) == 0
    # hard cdet at the same thresholds
    hs = ops.loss_sums(S, T, ths, 0.0, ops.LOSS_HARD_CDET)
    hl, _, _ = ops.loss_finish(S, T, ths, beta, 0.0, ops.LOSS_HARD_CDET, hs, want_grad=False)
    assert abs(hl.item() - orc.cdet(s, t, th32, beta, np.float64)) <= 1e-6 * max(1.0, abs(hl.item()))


@pytest.mark.parametrize("B", [5, 1000])
def test_bce_kernels(hip_lib, B):
    from neuralplda_amd import ops
    rng = np.random.default_rng(B)
    s = (rng.standard_normal(B) * 2).astype(np.float32)
    t = (rng.random(B) < 0.3).astype(np.float32)
    S, T = torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()
    th = [torch.tensor([0.25], dtype=torch.float32, device="cuda")]
    sums = ops.loss_sums(S, T, th, 0.0, ops.LOSS_BCE)
    loss, g, dth = ops.loss_finish(S, T, th, [], 0.0, ops.LOSS_BCE, sums)
    assert abs(loss.item() - orc.crossentropy(s, t, 0.25, np.float64)) <= 1e-5 * abs(loss.item())
    gref, dref = orc.crossentropy_grad(s, t, 0.25)
    assert relmax(g.cpu().numpy(), gref) <= 1e-5
    assert relmax(dth.cpu().numpy(), dref) <= 1e-4


# (500, 150, 160) / (400, 180, 192): the full-M weight-gradient kernel (NB = 10 / 12) with a last 32-column tile that is
# only partly there (500 = 15 x 32 + 20 columns; 400 = 12 x 32 + 16); B = 3 and 4097 rows odd: its 64 x 64 fall-back
@pytest.mark.parametrize("D0,D1,D2", [(512, 150, 150), (512, 170, 170), (64, 24, 20), (128, 40, 100), (500, 150, 160),
                                      (400, 180, 192)])
@pytest.mark.parametrize("B", [3, 100, 4096, 4097])
def test_backward_matches_oracle(hip_lib, D0, D1, D2, B):
    from neuralplda_amd import ops
    rng = np.random.default_rng(D1 * 7 + B)
    p = rand_params(rng, D0, D1, D2)
    x1 = rng.standard_normal((B, D0)).astype(np.float32)
    x2 = rng.standard_normal((B, D0)).astype(np.float32)
    g = (rng.standard_normal(B) / B).astype(np.float32)
    dev = [torch.from_numpy(a).cuda() for a in p.tensors()]
    packed = ops.pack_params(*dev)
    s, saved = ops.forward_train(torch.from_numpy(x1).cuda(), torch.from_numpy(x2).cuda(), packed)
    ref_s = orc.forward(x1, x2, p, np.float64)
    assert np.all(np.abs(s.cpu().numpy() - ref_s) <= 2e-5 + 1e-5 * np.abs(ref_s))
    # saved activations
    z1ref, (u1, y1ref, n1) = orc.extract_plda_embeddings(x1, p, np.float64, True)
    y = saved[3].cpu().numpy()
    np.testing.assert_allclose(y[:B, :D1], y1ref, atol=2e-6)
    assert np.all(y[:, D1:] == 0)
    flat = ops.backward(saved, torch.from_numpy(g).cuda(), packed, dev[4])
    grads = [t.cpu().numpy() for t in ops.split_flat_grad(flat, D0, D1, D2)]
    ref = orc.backward(x1, x2, g, p)
    for name, got in zip(("W1", "b1", "W2", "b2", "P_sqrt", "Q"), grads):
        assert got.shape == ref[name].shape
        assert relmax(got, ref[name]) <= 1e-4, (name, relmax(got, ref[name]))
    # deterministic: same inputs -> bit-identical gradients
    flat2 = ops.backward(saved, torch.from_numpy(g).cuda(), packed, dev[4])
    assert torch.equal(flat, flat2)


def test_backward_eps_branch(hip_lib):
    """A row with ||u|| == 0: normalize's clamp branch in the backward (du = dy / eps) must not produce NaN/inf
    in the other rows' gradients (the reference's autograd gives huge-but-finite values there too)."""
    from neuralplda_amd import ops
    rng = np.random.default_rng(11)
    p = rand_params(rng, 64, 24, 20)
    p.b1[:] = 0
    x1 = rng.standard_normal((50, 64)).astype(np.float32)
    x2 = rng.standard_normal((50, 64)).astype(np.float32)
    g = (rng.standard_normal(50) / 50).astype(np.float32)
    g[9] = 0  # keep the 1/eps row out of the sums so that values stay comparable
    x1[9] = 0
    dev = [torch.from_numpy(a).cuda() for a in p.tensors()]
    packed = ops.pack_params(*dev)
    s, saved = ops.forward_train(torch.from_numpy(x1).cuda(), torch.from_numpy(x2).cuda(), packed)
    flat = ops.backward(saved, torch.from_numpy(g).cuda(), packed, dev[4])
    assert torch.isfinite(flat).all()
    ref = orc.backward(x1, x2, g, p)
    got = ops.split_flat_grad(flat, 64, 24, 20)
    assert relmax(got[0].cpu().numpy(), ref["W1"]) <= 1e-4


def test_module_golden_g3_small(hip_lib):
    """NeuralPlda module (autograd bridges) vs the reference's own outputs (fp64 autograd) on G3."""
    g = np.load(os.path.join(G, "g3_loss_grad_small.npz"))
    p = orc.Params(g["W1"], g["b1"], g["W2"], g["b2"], g["P_sqrt"], g["Q"])
    X1, X2, T = (torch.from_numpy(g[k]).cuda() for k in ("x1", "x2", "t"))
    names = {"centering_and_LDA.weight": "W1", "centering_and_LDA.bias": "b1", "centering_and_wccn_plda.weight": "W2",
             "centering_and_wccn_plda.bias": "b2", "P_sqrt": "P_sqrt", "Q": "Q"}
    for lossname, tag in (("SoftCdet", "SoftCdet64"), ("softCdet", "SoftCdet64"), ("crossentropy", "crossentropy")):
        m = model_from(p, NC(64, 24, 20, loss=lossname), thetas=g["theta"], theta_xent=float(g["theta_xent"]))
        out = m(X1, X2)
        L = m.loss(out, T)
        L.backward()
        np.testing.assert_allclose(out.detach().cpu().numpy(), g[tag.replace("64", "") + "_s"], atol=2e-5)
        assert abs(L.item() - float(g[tag + "_L"])) <= 2e-5 * abs(float(g[tag + "_L"]))
        tol = 1e-4 if tag == "SoftCdet64" else 2e-2  # crossentropy golden is the reference's fp32 autograd
        for k, prm in m.named_parameters():
            key = f"{tag}_grad_{k}"
            if key not in g.files:
                continue
            ref = g[key]
            if prm.grad is None:
                assert np.all(ref == 0), k
                continue
            assert relmax(prm.grad.cpu().numpy(), ref) <= tol, (lossname, k, relmax(prm.grad.cpu().numpy(), ref))


def test_module_golden_kaldi170(hip_lib):
    g1 = np.load(os.path.join(G, "g1_kaldi_params.npz"))
    f = np.load(os.path.join(G, "g2_forward_kaldi170.npz"))
    gl = np.load(os.path.join(G, "g3_loss_kaldi170.npz"))
    p = orc.Params(g1["W1"], g1["b1"], g1["W2"], g1["b2"], g1["P_sqrt"], g1["Q"])
    m = model_from(p, NC(), thetas=gl["theta"])
    X1, X2 = torch.from_numpy(f["x1"]).cuda(), torch.from_numpy(f["x2"]).cuda()
    with torch.no_grad():
        s = m(X1, X2).cpu().numpy()
        z1 = m.extract_plda_embeddings(X1).cpu().numpy()
        sz = m.forward_from_plda_embeddings(torch.from_numpy(f["z1"]).cuda(), torch.from_numpy(f["z2"]).cuda())
    assert np.all(np.abs(s - f["s"]) <= 2e-5 + 1e-5 * np.abs(f["s"]))
    assert np.all(np.abs(s - f["s64"]) <= 2e-5 + 1e-5 * np.abs(f["s64"]))
    np.testing.assert_allclose(z1, f["z1"], atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(sz.cpu().numpy(), f["s_from_z"], atol=2e-6, rtol=1e-5)
    out = m(X1, X2)
    L = m.loss(out, torch.from_numpy(gl["t"]).cuda())
    L.backward()
    assert abs(L.item() - float(gl["L"])) <= 1e-4 * abs(float(gl["L"]))
    # against the fp64 oracle (the golden here is the reference's noisy fp32 autograd: loose check only)
    gs, dth = orc.softcdet_grad(f["s64"], gl["t"], gl["theta"].astype(np.float32).astype(np.float64), gl["beta"], 15.0)
    ref = orc.backward(f["x1"], f["x2"], gs, p)
    assert relmax(m.centering_and_LDA.weight.grad.cpu().numpy(), ref["W1"]) <= 2e-4
    assert relmax(m.centering_and_wccn_plda.weight.grad.cpu().numpy(), ref["W2"]) <= 2e-4
    assert relmax(m.Q.grad.cpu().numpy(), ref["Q"]) <= 2e-4
    assert relmax(m.P_sqrt.grad.cpu().numpy(), ref["P_sqrt"]) <= 2e-4
    assert relmax(m.Q.grad.cpu().numpy(), gl["grad_Q"]) <= 2e-2
    assert relmax([m.Th99.grad.item(), m.Th199.grad.item()], dth) <= 2e-4


def test_adam_trajectory_g4(hip_lib):
    """Three optimiser steps exactly as xvector_NeuralPlda_pytorch.py:35-43 takes them."""
    from neuralplda_amd import models
    g = np.load(os.path.join(G, "g4_adam_small.npz"))
    d = np.load(os.path.join(G, "g3_loss_grad_small.npz"))
    m = models.NeuralPlda(NC(64, 24, 20))
    sd = m.state_dict()
    for k in g["keys"]:
        sd[str(k)].copy_(torch.from_numpy(g["p0_" + str(k)]))
    m = m.cuda()
    opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=1e-5)
    losses = []
    for step in range(3):
        opt.zero_grad()
        lo, hi = step * 128, (step + 1) * 128
        o = m(torch.from_numpy(d["x1"][lo:hi]).cuda(), torch.from_numpy(d["x2"][lo:hi]).cuda())
        L = m.loss(o, torch.from_numpy(d["t"][lo:hi]).cuda())
        losses.append(L.item())
        L.backward()
        opt.step()
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-5)
    g64 = np.load(os.path.join(G, "g4_adam_small_f64.npz"))  # the same three steps by the reference in float64
    np.testing.assert_allclose(losses, g64["losses"], rtol=2e-5)
    for k in g["keys"]:
        k = str(k)
        got = m.state_dict()[k].cpu().numpy().astype(np.float64)
        # Adam's first steps move every weight by ~lr whatever the gradient scale, so compare the DISPLACEMENT.
        # Against the float64 trajectory what is left is this build's own fp32 noise: the update m_hat / sqrt(v_hat) of
        # an element changes by the relative error of its gradients (<= 1e-4 here), i.e. by <= ~1e-3 of the 3 lr
        # displacement once elements whose gradient changes sign between steps are allowed for, and by far less on
        # average; float32 storage of the parameters adds 6e-8 |p| per step.
        p0 = g["p0_" + k].astype(np.float64)
        disp_ref, disp = g64["p3_" + k] - p0, got - p0
        scale = max(np.abs(disp_ref).max(), 1e-12)
        err = np.abs(disp - disp_ref)
        assert err.max() <= 4e-3 * scale + 3 * 6e-8 * np.abs(p0).max(), (k, err.max() / scale)
        assert err.mean() <= 5e-4 * scale + 3 * 6e-8 * np.abs(p0).max(), (k, err.mean() / scale)
        # and the reference's own fp32 run stays within ITS noise of us (measured 2.2 % of the displacement: its sigma'
        # is formed from a saturated fp32 sigmoid, tests/test_oracle_golden.py)
        assert np.abs(disp - (g["p3_" + k] - g["p0_" + k])).max() <= 0.03 * scale + 1e-7, k


@pytest.mark.parametrize("D", [150, 170])
def test_score_indexed_and_gather(hip_lib, D):
    from neuralplda_amd import ops
    rng = np.random.default_rng(D)
    p = rand_params(rng, 512, D, D)
    N, B = 1000, 5000
    x = rng.standard_normal((N, 512)).astype(np.float32)
    i1 = rng.integers(0, N, B)
    i2 = rng.integers(0, N, B)
    packed = ops.pack_params(*[torch.from_numpy(a).cuda() for a in p.tensors()])
    X = torch.from_numpy(x).cuda()
    z, q = ops.embed(X, packed)
    s = ops.score_indexed(z, q, torch.from_numpy(i1), torch.from_numpy(i2), packed).cpu().numpy()
    ref = orc.forward(x[i1], x[i2], p, np.float64)
    assert np.all(np.abs(s - ref) <= 2e-5 + 1e-5 * np.abs(ref))
    # indexed == dense on the gathered rows (same kernels' arithmetic up to summation order)
    g1 = ops.gather_rows(X, torch.from_numpy(i1).cuda())
    assert torch.equal(g1.cpu(), torch.from_numpy(x[i1]))
    dense = ops.score_pairs(g1, ops.gather_rows(X, torch.from_numpy(i2).cuda()), packed).cpu().numpy()
    np.testing.assert_allclose(s, dense, atol=1e-5, rtol=1e-5)
    # out-of-range index -> NaN, not a fault
    bad = ops.score_indexed(z, q, torch.tensor([0, N]), torch.tensor([1, 2]), packed).cpu().numpy()
    assert np.isfinite(bad[0]) and np.isnan(bad[1])
    # q=None: the self terms formed from the rows themselves — the same scores to the fp32 tolerance
    s_self = ops.score_indexed(z, None, torch.from_numpy(i1), torch.from_numpy(i2), packed).cpu().numpy()
    assert np.all(np.abs(s_self - ref) <= 2e-5 + 1e-5 * np.abs(ref))
    bad = ops.score_indexed(z, None, torch.tensor([0, -1]), torch.tensor([1, 2]), packed).cpu().numpy()
    assert np.isfinite(bad[0]) and np.isnan(bad[1])


@pytest.mark.parametrize("D1,D2", [(16, 16), (40, 24), (100, 70), (128, 128), (160, 160), (192, 180)])
def test_score_indexed_other_widths(hip_lib, D1, D2):
    """Every column count of the indexed kernel (1 .. 6 float4 columns per lane), with and without the q table, ragged B."""
    from neuralplda_amd import ops
    rng = np.random.default_rng(D1 + D2)
    p = rand_params(rng, 64, D1, D2)
    N, B = 300, 1237
    x = rng.standard_normal((N, 64)).astype(np.float32)
    i1, i2 = rng.integers(0, N, B), rng.integers(0, N, B)
    packed = ops.pack_params(*[torch.from_numpy(a).cuda() for a in p.tensors()])
    z, q = ops.embed(torch.from_numpy(x).cuda(), packed)
    ref = orc.forward(x[i1], x[i2], p, np.float64)
    for qq in (q, None):
        s = ops.score_indexed(z, qq, torch.from_numpy(i1), torch.from_numpy(i2), packed).cpu().numpy()
        assert np.all(np.abs(s - ref) <= 2e-5 + 1e-5 * np.abs(ref)), (D1, D2, qq is None)


def test_cpu_tensors_are_staged_through_the_device(hip_lib):
    """utils/scorefile_generator.py:25-34 moves model and data to CPU before calling forward(): our module
    must still produce the HIP result (and return it on the CPU), never compute on the host."""
    rng = np.random.default_rng(2)
    p = rand_params(rng, 512, 150, 150)
    m = model_from(p, NC(512, 150, 150)).to(torch.device("cpu"))
    x1 = rng.standard_normal((33, 512)).astype(np.float32)
    x2 = rng.standard_normal((33, 512)).astype(np.float32)
    with torch.no_grad():
        s = m.forward(torch.from_numpy(x1), torch.from_numpy(x2))
    assert s.device.type == "cpu"
    ref = orc.forward(x1, x2, p, np.float64)
    assert np.all(np.abs(s.numpy() - ref) <= 2e-5 + 1e-5 * np.abs(ref))
    assert m.forward(torch.tensor([]), torch.tensor([])).shape == (0,)


def test_metrics_on_device_match_golden(hip_lib):
    g = np.load(os.path.join(G, "g5_metrics.npz"))
    p = rand_params(np.random.default_rng(0), 64, 24, 20)
    m = model_from(p, NC(64, 24, 20), thetas=g["theta"])
    S, T = torch.from_numpy(g["s"]).cuda(), torch.from_numpy(g["t"]).cuda()
    with torch.no_grad():
        assert abs(m.cdet(S, T).item() - float(g["cdet"])) <= 1e-6 * float(g["cdet"])
        assert abs(m.softcdet(S, T).item() - float(g["softcdet"])) <= 2e-5 * float(g["softcdet"])
        assert abs(m.crossentropy(S, T).item() - float(g["xent"])) <= 2e-5 * float(g["xent"])
        mc, th = m.minc(S, T)
        assert abs(mc.item() - float(g["minc"])) <= 1e-6
        assert th[99.0].item() == np.float32(g["minc_th"][0]) and th[199.0].item() == np.float32(g["minc_th"][1])
        m.minc(S, T, update_thresholds=True)
        assert m.Th99.item() == np.float32(g["minc_th"][0]) and m.threshold[199.0].item() == np.float32(g["minc_th"][1])
        mcs, _ = m.minc(torch.from_numpy(g["s_sep"]).cuda(), T)
        assert abs(mcs.item() - float(g["minc_sep"])) <= 1e-7
        assert m.minc(torch.from_numpy(g["s_sep"]).cuda(), T, exact=True)[0].item() == 0.0


def test_graphed_step_matches_eager(hip_lib):
    """A HIP-graph replay of the whole optimisation step gives the same trajectory as the eager step."""
    from neuralplda_amd import models, train
    rng = np.random.default_rng(21)
    p = rand_params(rng, 512, 150, 150)
    B = 512
    xs = [(torch.from_numpy(rng.standard_normal((B, 512)).astype(np.float32)).cuda(),
           torch.from_numpy(rng.standard_normal((B, 512)).astype(np.float32)).cuda(),
           torch.from_numpy((rng.random(B) < 0.2).astype(np.float32)).cuda()) for _ in range(4)]

    def run(graphed):
        m = model_from(p, NC(512, 150, 150), thetas=[-0.5, -0.3])
        opt = train.make_optimizer(m, 1e-3, capturable=graphed)
        losses = []
        if graphed:
            step = train.GraphedTrainStep(m, opt, B)
            for x1, x2, t in xs:
                losses.append(step(x1, x2, t).item())
        else:
            for x1, x2, t in xs:
                opt.zero_grad()
                L = m.loss(m(x1, x2), t)
                L.backward()
                opt.step()
                losses.append(L.item())
        return losses, {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}

    le, pe = run(False)
    lg, pg = run(True)
    np.testing.assert_allclose(lg, le, rtol=1e-5)
    for k in pe:
        np.testing.assert_allclose(pg[k], pe[k], rtol=1e-4, atol=1e-6, err_msg=k)


def test_forward_after_graph_replay_scores_with_the_new_weights(hip_lib):
    """A graph replay rewrites the parameters without Python seeing it; the model's packed-image cache must notice
    (GraphedTrainStep bumps the version counters): replay -> forward -> replay -> forward, each forward equal to an
    eager model stepped the same way."""
    from neuralplda_amd import train
    rng = np.random.default_rng(22)
    p = rand_params(rng, 512, 150, 150)
    B = 256
    xs = [(torch.from_numpy(rng.standard_normal((B, 512)).astype(np.float32)).cuda(),
           torch.from_numpy(rng.standard_normal((B, 512)).astype(np.float32)).cuda(),
           torch.from_numpy((rng.random(B) < 0.2).astype(np.float32)).cuda()) for _ in range(3)]
    v1 = torch.from_numpy(rng.standard_normal((300, 512)).astype(np.float32)).cuda()
    v2 = torch.from_numpy(rng.standard_normal((300, 512)).astype(np.float32)).cuda()
    mg = model_from(p, NC(512, 150, 150), thetas=[-0.5, -0.3])
    me = model_from(p, NC(512, 150, 150), thetas=[-0.5, -0.3])
    og = train.make_optimizer(mg, 1e-2, capturable=True)
    oe = train.make_optimizer(me, 1e-2, capturable=False)
    step = train.GraphedTrainStep(mg, og, B)
    with torch.no_grad():
        s_prev = mg(v1, v2).clone()  # fills the cache with the initial image
    for x1, x2, t in xs:
        step(x1, x2, t)
        oe.zero_grad()
        me.loss(me(x1, x2), t).backward()
        oe.step()
        with torch.no_grad():
            sg, se = mg(v1, v2), me(v1, v2)
        assert (sg - s_prev).abs().max().item() > 1e-4, "forward after a replay returned the previous weights' scores"
        np.testing.assert_allclose(sg.cpu().numpy(), se.cpu().numpy(), rtol=1e-4, atol=2e-5)
        s_prev = sg.clone()


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("lossname", ["SoftCdet", "crossentropy"])
def test_fused_train_step_matches_torch_adam(hip_lib, graph, lossname):
    """FusedTrainStep (direct C-ABI launches + one-launch Adam) follows the autograd + torch.optim.Adam
    trajectory of the reference's training loop (xvector_NeuralPlda_pytorch.py:35-43, :139)."""
    from neuralplda_amd import train
    rng = np.random.default_rng(31)
    p = rand_params(rng, 512, 170, 170)
    B = 256
    xs = [(torch.from_numpy(rng.standard_normal((B, 512)).astype(np.float32)).cuda(),
           torch.from_numpy(rng.standard_normal((B, 512)).astype(np.float32)).cuda(),
           torch.from_numpy((rng.random(B) < 0.2).astype(np.float32)).cuda()) for _ in range(5)]
    m_ref = model_from(p, NC(loss=lossname), thetas=[-0.5, -0.3], theta_xent=0.1)
    opt = train.make_optimizer(m_ref, 1e-3)
    ref_losses = []
    for x1, x2, t in xs:
        opt.zero_grad()
        L = m_ref.loss(m_ref(x1, x2), t)
        L.backward()
        opt.step()
        ref_losses.append(L.item())
    m = model_from(p, NC(loss=lossname), thetas=[-0.5, -0.3], theta_xent=0.1)
    step = train.FusedTrainStep(m, 1e-3, weight_decay=1e-5, batch_size=B, graph=graph)
    losses = [step(x1, x2, t).item() for x1, x2, t in xs]
    np.testing.assert_allclose(losses, ref_losses, rtol=2e-5)
    for (k, a), (_, b) in zip(m.state_dict().items(), m_ref.state_dict().items()):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-4, atol=2e-6, err_msg=k)
    assert step.step_count[0].item() == 5 and step.step_count[1].item() == 0


def _same_step(a, b, D1, what, scale=1.0):
    """Two tensors left by the one-call step and by the separate launches: the same step to rounding.  (Round 5: the one-call
    step's first kernel works on 8-pair half tiles at the recipe shapes, the separate forward / backward on 16-pair tiles —
    the cross-wave and cross-pair sums associate differently, so the bits may differ; parity proper is against the fp64
    oracle and autograd + torch.optim.Adam, test_fused_train_step_matches_torch_adam and the golden trajectories.)"""
    a, b = a.double(), b.double()
    tol = 2e-6 + 2e-4 * b.abs()
    assert bool(((a - b).abs() <= tol).all()), (what, (a - b).abs().max().item())


@pytest.mark.parametrize("lossname,B,D", [("SoftCdet", 4096, 150), ("SoftCdet", 1003, 170), ("crossentropy", 250, 150),
                                          ("SoftCdet", 16, 40)])
def test_one_call_step_equals_the_separate_launches(hip_lib, lossname, B, D):
    """nplda_train_step_f32 (loss folded into the data-gradient kernel, slab sums + Adam + re-pack in one launch) against
    the same step as separate C-ABI calls: the SAME parameter bits after every step (dL/ds, the slabs and Adam's update
    are the same arithmetic); the loss scalar may differ in its last bit (fp64 sums taken per
    block of 16 pairs)."""
    from neuralplda_amd import ops, train
    rng = np.random.default_rng(77)
    p = rand_params(rng, 512, D, D)
    xs = [(torch.from_numpy(rng.standard_normal((B, 512)).astype(np.float32)).cuda(),
           torch.from_numpy(rng.standard_normal((B, 512)).astype(np.float32)).cuda(),
           torch.from_numpy((rng.random(B) < 0.2).astype(np.float32)).cuda()) for _ in range(4)]
    nc = NC(512, D, D, loss=lossname)
    m_a = model_from(p, nc, thetas=[-0.5, -0.3], theta_xent=0.1)
    m_b = model_from(p, nc, thetas=[-0.5, -0.3], theta_xent=0.1)
    sa = train.FusedTrainStep(m_a, 1e-3, weight_decay=1e-5, batch_size=B, graph=False)
    sb = train.FusedTrainStep(m_b, 1e-3, weight_decay=1e-5, batch_size=B, graph=False)
    assert sa._one_call
    sb._one_call = False
    for i, (x1, x2, t) in enumerate(xs):
        la, lb = sa(x1, x2, t), sb(x1, x2, t)
        assert abs(la.item() - lb.item()) <= 2e-7 * abs(lb.item())
        for (k, a), (_, b) in zip(m_a.state_dict().items(), m_b.state_dict().items()):
            _same_step(a, b, D, (i, k), scale=0)
        _same_step(sa.m, sb.m, D, (i, "m"))
        _same_step(sa.v, sb.v, D, (i, "v"))
        # the image the step carries along is the image of the updated parameters
        fresh = ops.pack_params(*[q.detach() for q in sa.params])
        assert torch.equal(sa._packed.buf, fresh.buf), i
        if i == 1:  # somebody else rewrites a parameter between steps: the step notices and re-packs
            with torch.no_grad():
                for mm in (m_a, m_b):
                    list(mm._params())[1].mul_(0.5)
    assert sa.step_count[0].item() == 4 and sa.step_count[1].item() == 0


@pytest.mark.parametrize("D0,D1,D2,betas,B", [(512, 128, 128, (99.0,), 333), (512, 192, 192, (9.9, 99.0, 199.0), 2048),
                                               (256, 150, 100, (99.0, 199.0), 640), (72, 24, 20, (99.0, 199.0), 100)])
def test_one_call_step_other_shapes(hip_lib, D0, D1, D2, betas, B):
    """The remaining kernel instantiations of the one-call step (NB = 8, 12; D1 != D2; x-vector dims other than 512, one
    not a multiple of 16; 1 and 3 thresholds): same parameter bits as the separate launches, which are pinned by the oracle."""
    from neuralplda_amd import train
    rng = np.random.default_rng(79)
    p = rand_params(rng, D0, D1, D2)
    xs = [(torch.from_numpy(rng.standard_normal((B, D0)).astype(np.float32)).cuda(),
           torch.from_numpy(rng.standard_normal((B, D0)).astype(np.float32)).cuda(),
           torch.from_numpy((rng.random(B) < 0.25).astype(np.float32)).cuda()) for _ in range(3)]
    nc = NC(D0, D1, D2, beta=betas)
    ths = [-0.5, -0.3, -0.1][:len(betas)]
    m_a, m_b = model_from(p, nc, thetas=ths), model_from(p, nc, thetas=ths)
    sa = train.FusedTrainStep(m_a, 1e-3, weight_decay=1e-5, batch_size=B, graph=False)
    sb = train.FusedTrainStep(m_b, 1e-3, weight_decay=1e-5, batch_size=B, graph=False)
    assert sa._one_call
    sb._one_call = False
    for x1, x2, t in xs:
        la, lb = sa(x1, x2, t), sb(x1, x2, t)
        assert abs(la.item() - lb.item()) <= 2e-7 * abs(lb.item())
        for (k, a), (_, b) in zip(m_a.state_dict().items(), m_b.state_dict().items()):
            _same_step(a, b, D1, k, scale=0)


@pytest.mark.parametrize("graph", [False, True])
def test_step_rows_gathers_inside_the_step(hip_lib, graph):
    """step_rows (pairs named by rows of the resident x-vector table; the step's first kernel gathers them itself) gives
    the same parameter bits as gathering with gather_rows and calling the step on the (B, D0) tensors."""
    from neuralplda_amd import ops, train
    rng = np.random.default_rng(91)
    D, B, N = 170, 777 if not graph else 1024, 5000
    p = rand_params(rng, 512, D, D)
    table = torch.from_numpy(rng.standard_normal((N, 512)).astype(np.float32)).cuda()
    batches = [(torch.from_numpy(rng.integers(0, N, B)).cuda(), torch.from_numpy(rng.integers(0, N, B)).cuda(),
                torch.from_numpy((rng.random(B) < 0.2).astype(np.float32)).cuda()) for _ in range(3)]
    nc = NC(512, D, D)
    m_a, m_b = model_from(p, nc, thetas=[-0.5, -0.3]), model_from(p, nc, thetas=[-0.5, -0.3])
    sa = train.FusedTrainStep(m_a, 1e-3, weight_decay=1e-5, batch_size=B, graph=graph)
    sb = train.FusedTrainStep(m_b, 1e-3, weight_decay=1e-5, batch_size=B, graph=False)
    for r1, r2, t in batches:
        la = sa.step_rows(table, r1, r2, t)
        lb = sb(ops.gather_rows(table, r1), ops.gather_rows(table, r2), t)
        assert la.item() == lb.item()
        for (k, a), (_, b) in zip(m_a.state_dict().items(), m_b.state_dict().items()):
            assert torch.equal(a, b), k
    assert sa.step_count[0].item() == 3


def test_step_record_walks_a_device_resident_epoch(hip_lib):
    """begin_epoch / step_record (nplda_train_step_records_f32: the captured step trains on its staging record and its own
    last kernel copies the epoch's next record there, counted by a device cursor) leaves the same parameter bits as
    step_rows fed the same batches one by one; the cursor ends on the record count; one record too many raises."""
    from neuralplda_amd import train
    rng = np.random.default_rng(91)
    B, nb, N = 256, 5, 3000
    p = rand_params(rng, 512, 150, 150)
    nc = NC(D1=150, D2=150, loss="SoftCdet")
    table = torch.from_numpy(rng.standard_normal((N, 512)).astype(np.float32)).cuda()
    r1 = torch.from_numpy(rng.integers(0, N, (nb, B))).cuda()
    r2 = torch.from_numpy(rng.integers(0, N, (nb, B))).cuda()
    lab = torch.from_numpy((rng.random((nb, B)) < 0.2).astype(np.float32)).cuda()
    records = torch.empty((nb, 20 * B), dtype=torch.uint8, device="cuda")
    records[:, :8 * B].view(torch.int64).copy_(r1)
    records[:, 8 * B:16 * B].view(torch.int64).copy_(r2)
    records[:, 16 * B:].view(torch.float32).copy_(lab)
    m_a, m_b = model_from(p, nc, thetas=[-0.5, -0.3]), model_from(p, nc, thetas=[-0.5, -0.3])
    sa = train.FusedTrainStep(m_a, 1e-3, weight_decay=1e-5, batch_size=B, graph=True)
    sb = train.FusedTrainStep(m_b, 1e-3, weight_decay=1e-5, batch_size=B, graph=True)
    assert sa.records_ok(table, records) and not sa.records_ok(table, records[:, :-4])
    sa.begin_epoch(table, records)
    for k in range(nb):
        la = sa.step_record().item()
        lb = sb.step_rows(table, r1[k], r2[k], lab[k]).item()
        assert la == lb, k
        for (key, a), (_, b) in zip(m_a.state_dict().items(), m_b.state_dict().items()):
            assert torch.equal(a, b), (k, key)
    assert sa._cursor[1].item() == nb and sa.step_count[0].item() == nb
    assert abs(sa.pop_loss_mean() - sb.pop_loss_mean()) < 1e-12
    with pytest.raises(RuntimeError):
        sa.step_record()
    sa.begin_epoch(table, records[1:3])  # a second epoch on the same graph: only the cursor moves
    sa.step_record()
    sb.step_rows(table, r1[1], r2[1], lab[1])
    assert torch.equal(next(iter(m_a.state_dict().values())), next(iter(m_b.state_dict().values())))


def test_one_call_step_reports_the_applied_gradient(hip_lib):
    """grad_out of nplda_train_step_f32 = flat gradient of the separate backward + dtheta of the separate loss."""
    from neuralplda_amd import ops
    rng = np.random.default_rng(78)
    D, B = 150, 777
    rp = rand_params(rng, 512, D, D)
    prm = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (rp.W1, rp.b1, rp.W2, rp.b2, rp.P_sqrt, rp.Q)]
    x1 = torch.from_numpy(rng.standard_normal((B, 512)).astype(np.float32)).cuda()
    x2 = torch.from_numpy(rng.standard_normal((B, 512)).astype(np.float32)).cuda()
    t = torch.from_numpy((rng.random(B) < 0.3).astype(np.float32)).cuda()
    ths = [torch.tensor([-0.4], device="cuda"), torch.tensor([-0.2], device="cuda")]
    betas, alpha = [99.0, 199.0], 15.0
    packed = ops.pack_params(*prm)
    s, saved = ops.forward_train(x1, x2, packed)
    loss, g, dth, _ = ops.loss_fwd_bwd(s, t, ths, betas, alpha, ops.LOSS_SOFTCDET)
    flat = ops.backward(saved, g, packed, prm[4])
    n = flat.numel()
    m, v, step = torch.zeros(n + 2, device="cuda"), torch.zeros(n + 2, device="cuda"), torch.zeros(2, device="cuda")
    out, lbuf = torch.zeros(n + 2, device="cuda"), torch.zeros((), device="cuda")
    ws = ops.train_step_workspace(B, packed)
    ops.train_step(x1, x2, t, prm, ths, betas, alpha, ops.LOSS_SOFTCDET, m, v, step, 1e-3, 0.9, 0.999, 1e-8, 1e-5, packed,
                   ws, lbuf, grad_out=out, loss_sum=(acc := torch.full((1,), 2.5, dtype=torch.float64, device="cuda")))
    assert acc.item() == 2.5 + float(lbuf.item())  # the running loss sum of the training log: += loss, in fp64
    _same_step(out[:n], flat, D, "flat gradient")
    np.testing.assert_allclose(out[n:].cpu().numpy(), dth.cpu().numpy(), rtol=1e-6)
    assert abs(lbuf.item() - loss.item()) <= 2e-7 * abs(loss.item())
    assert ops.train_step_workspace(16385, packed) is None


def test_cfg2_full_size_minibatch_from_a_voxceleb_scale_table(hip_lib):
    """BASELINE cfg2 at full size: 4096-pair minibatches gathered from a 1.2 M-utterance resident table (2.4 GB),
    512 -> 150 -> 150, SoftCdet, three fused optimiser steps.  Properties: the device gather equals plain indexing bit
    for bit, the loss and gradient of the first step agree with the fp64 oracle, and the fused step follows
    autograd + torch.optim.Adam."""
    from neuralplda_amd import ops, train
    rng = np.random.default_rng(77)
    N, B, D = 1_200_000, 4096, 150
    gen = torch.Generator(device="cuda").manual_seed(3)
    table = torch.randn(N, 512, device="cuda", generator=gen)
    p = rand_params(rng, 512, D, D)
    batches = []
    for _ in range(3):
        i1 = torch.randint(0, N, (B,), device="cuda", generator=gen)
        i2 = torch.randint(0, N, (B,), device="cuda", generator=gen)
        t = (torch.rand(B, device="cuda", generator=gen) < 0.1).float()
        x1, x2 = ops.gather_rows(table, i1), ops.gather_rows(table, i2)
        assert torch.equal(x1, table[i1]) and torch.equal(x2, table[i2])
        batches.append((x1, x2, t))
    m_ref = model_from(p, NC(D1=D, D2=D, loss="SoftCdet"), thetas=[-0.5, -0.3], theta_xent=0.1)
    x1, x2, t = batches[0]
    L = m_ref.loss(m_ref(x1, x2), t)
    L.backward()
    sref = orc.forward(x1.cpu().numpy(), x2.cpu().numpy(), p, np.float64)
    Lref = orc.softcdet(sref, t.cpu().numpy(), [-0.5, -0.3], [99.0, 199.0], 15.0, np.float64)
    gs, _ = orc.softcdet_grad(sref, t.cpu().numpy(), [-0.5, -0.3], [99.0, 199.0], 15.0)
    gref = orc.backward(x1.cpu().numpy(), x2.cpu().numpy(), gs, p)
    assert abs(L.item() - float(Lref)) <= 1e-4 * abs(float(Lref))
    for name, prm in (("W1", m_ref.centering_and_LDA.weight), ("W2", m_ref.centering_and_wccn_plda.weight),
                      ("Q", m_ref.Q), ("P_sqrt", m_ref.P_sqrt)):
        np.testing.assert_allclose(prm.grad.cpu().numpy(), gref[name], atol=1e-4 * np.abs(gref[name]).max(), rtol=1e-3)
    m_ref.zero_grad()
    opt = train.make_optimizer(m_ref, 1e-4)
    ref_losses = []
    for x1, x2, t in batches:
        opt.zero_grad()
        L = m_ref.loss(m_ref(x1, x2), t)
        L.backward()
        opt.step()
        ref_losses.append(L.item())
    m = model_from(p, NC(D1=D, D2=D, loss="SoftCdet"), thetas=[-0.5, -0.3], theta_xent=0.1)
    step = train.FusedTrainStep(m, 1e-4, weight_decay=1e-5, batch_size=B, graph=True)
    losses = [step(x1, x2, t).item() for x1, x2, t in batches]
    np.testing.assert_allclose(losses, ref_losses, rtol=2e-5)
    for (k, a), (_, b) in zip(m.state_dict().items(), m_ref.state_dict().items()):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-4, atol=2e-6, err_msg=k)


def _tiny_trial_set(tmp_path, rng, B):
    """300 utterances, 1000 trials in a TSV, the vectorised loader over it (ragged last batch at B = 128)."""
    from neuralplda_amd import sv_trials_loaders as svl
    n_utt, n_trials = 300, 1000
    ids = [f"spk{u // 3:03d}-utt{u:04d}" for u in range(n_utt)]
    xv = rng.standard_normal((n_utt, 512)).astype(np.float32)
    mega = {u: xv[i] for i, u in enumerate(ids)}
    num_to_id = dict(enumerate(ids))
    id_to_num = {u: i for i, u in num_to_id.items()}
    a, b = rng.integers(0, n_utt, n_trials), rng.integers(0, n_utt, n_trials)
    lab = (a // 3 == b // 3).astype(int)
    lab[rng.random(n_trials) < 0.15] = 1
    tf = tmp_path / "train.tsv"
    tf.write_text("\n".join(f"{ids[i]}\t{ids[j]}\t{l}" for i, j, l in zip(a, b, lab)) + "\n")
    loader = svl.combine_trials_and_get_loader([str(tf)], id_to_num, subsample_factors=[1.01], batch_size=B)
    return mega, num_to_id, loader


def test_device_resident_epoch_equals_the_generic_loop(hip_lib, tmp_path):
    """train() with the vectorised loader + FusedTrainStep keeps the epoch's index arrays on the device and gathers inside
    the captured step (TrialLoader.device_batches, FusedTrainStep.step_rows).  It must visit the same batches and leave
    the same parameters as the generic loop (host batches -> load_xvec_trials_from_numbatch -> step), bit for bit,
    including the ragged last batch (eager fall-back of the graph path)."""
    import contextlib
    import io
    from neuralplda_amd import sv_trials_loaders as svl
    from neuralplda_amd import train
    rng = np.random.default_rng(5)
    B = 128
    mega, num_to_id, loader = _tiny_trial_set(tmp_path, rng, B)
    p = rand_params(rng, 512, 150, 150)
    nc = NC(D1=150, D2=150, loss="SoftCdet")
    nc.log_interval = 3

    def run(fast):
        m = model_from(p, nc, thetas=[-0.5, -0.3])
        step = train.FusedTrainStep(m, 1e-3, weight_decay=1e-5, batch_size=B, graph=True)
        torch.manual_seed(11)
        out = io.StringIO()
        with contextlib.redirect_stdout(out):
            if fast:
                train.train(nc, m, torch.device("cuda"), loader, mega, num_to_id, None, 1, step_fn=step)
            else:  # the same batches handed over as a plain iterable: generic path
                batches = list(loader)

                class Plain(list):
                    dataset = loader.dataset
                train.train(nc, m, torch.device("cuda"), Plain(batches), mega, num_to_id, None, 1, step_fn=step)
        return {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}, out.getvalue(), step

    sd_fast, log_fast, step_fast = run(True)
    sd_gen, log_gen, _ = run(False)
    assert step_fast._graph_rec is not None and step_fast.step_count[0].item() == 8  # 7 replays + 1 ragged eager step
    assert log_fast == log_gen and log_fast.count("Train Epoch") == 3
    # the figure on each progress line is the MEAN of the losses since the previous line (xvector_NeuralPlda_pytorch.py:
    # 41-47), also when the step is a replayed graph whose loss tensor is rewritten in place by every replay
    m = model_from(p, nc, thetas=[-0.5, -0.3])
    step = train.FusedTrainStep(m, 1e-3, weight_decay=1e-5, graph=False)
    torch.manual_seed(11)
    per_step = []
    for d1, d2, t in loader:
        x1, x2 = svl.load_xvec_trials_from_numbatch(mega, num_to_id, d1, d2, "cuda")
        per_step.append(float(step(x1, x2, t.cuda())))
    want, acc = [], []
    for i, l in enumerate(per_step):
        acc.append(l)
        if i % nc.log_interval == 0:
            want.append("{:.6f}".format(sum(acc) / len(acc)))
            acc = []
    got = [ln.rsplit(" ", 1)[1] for ln in log_fast.strip().splitlines()]
    assert got == want and "{:.6f}".format(per_step[6]) != want[2]  # (a mean, not the interval's last loss)
    for k in sd_gen:
        assert np.array_equal(sd_fast[k], sd_gen[k]), k
    # an index that maps to no utterance raises like the reference-shaped gather
    bad = dict(num_to_id)
    bad[0] = "nobody"
    with pytest.raises(KeyError):
        m = model_from(p, nc, thetas=[-0.5, -0.3])
        step = train.FusedTrainStep(m, 1e-3, batch_size=B, graph=True)
        with contextlib.redirect_stdout(io.StringIO()):
            train.train(nc, m, torch.device("cuda"), loader, mega, bad, None, 1, step_fn=step)


@pytest.mark.parametrize("B", [129, 130, 131, 250, 1000, 2000])
def test_device_resident_epoch_other_batch_sizes(hip_lib, tmp_path, B):
    """The forms of train()'s device-resident epoch the main test does not reach: a batch size that is not a multiple of 4
    (no packed records on the cursor path: one record copy per step instead), an ODD batch size (no packed record at all: three
    index copies per step), an epoch of exactly full batches (no ragged
    tail), and an epoch shorter than one batch (only the tail).  Same parameters and log lines as the generic loop."""
    import contextlib
    import io
    from neuralplda_amd import train
    rng = np.random.default_rng(6)
    mega, num_to_id, loader = _tiny_trial_set(tmp_path, rng, B)
    p = rand_params(rng, 512, 150, 150)
    nc = NC(D1=150, D2=150, loss="SoftCdet")
    nc.log_interval = 2

    def run(fast):
        m = model_from(p, nc, thetas=[-0.5, -0.3])
        step = train.FusedTrainStep(m, 1e-3, weight_decay=1e-5, batch_size=B, graph=True)
        torch.manual_seed(13)
        out = io.StringIO()
        with contextlib.redirect_stdout(out):
            if fast:
                train.train(nc, m, torch.device("cuda"), loader, mega, num_to_id, None, 1, step_fn=step)
            else:
                class Plain(list):
                    dataset = loader.dataset
                train.train(nc, m, torch.device("cuda"), Plain(list(loader)), mega, num_to_id, None, 1, step_fn=step)
        return {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}, out.getvalue(), step

    sd_fast, log_fast, step_fast = run(True)
    sd_gen, log_gen, _ = run(False)
    n = len(loader.dataset)
    assert step_fast.step_count[0].item() == (n + B - 1) // B
    assert (step_fast._graph_rec is not None) == (B % 4 == 0 and n >= B)
    assert log_fast == log_gen and log_fast.count("Train Epoch") >= 1
    for k in sd_gen:
        assert np.array_equal(sd_fast[k], sd_gen[k]), k


def test_validate_device_resident_pass_equals_the_generic_loop(hip_lib, tmp_path, monkeypatch):
    """validate() over the vectorised loader gathers from the resident table on the device; scores, metrics, thresholds
    written back and the printed report must equal the generic loop's (same batches -> same forward launches).  (The DENSE
    pass: the default embed-once pass of a list that names each utterance several times scores through another kernel —
    tests/test_validate_gpu.py.)"""
    monkeypatch.setenv("NPLDA_VALIDATE_DENSE", "1")
    import contextlib
    import io
    from neuralplda_amd import train
    rng = np.random.default_rng(6)
    mega, num_to_id, loader = _tiny_trial_set(tmp_path, rng, 128)
    p = rand_params(rng, 512, 150, 150)
    nc = NC(D1=150, D2=150, loss="SoftCdet")

    def run(fast):
        m = model_from(p, nc, thetas=[-0.5, -0.3])
        torch.manual_seed(3)
        out = io.StringIO()
        with contextlib.redirect_stdout(out):
            if fast:
                mc, th = train.validate(nc, m, torch.device("cuda"), mega, num_to_id, loader, update_thresholds=True)
            else:
                mc, th = train.validate(nc, m, torch.device("cuda"), mega, num_to_id, list(loader), update_thresholds=True)
        return float(mc), {k: float(v) for k, v in th.items()}, out.getvalue(), [float(m.threshold[b].detach()) for b in m.beta]

    fast, gen = run(True), run(False)
    assert fast == gen
    assert fast[3] == [fast[1][b] for b in nc.beta]  # update_thresholds wrote the arg-min scores into Th{beta}


@pytest.mark.parametrize("B", [1, 3, 255, 4096, 4097, 9000])
@pytest.mark.parametrize("kind", ["SoftCdet", "crossentropy"])
def test_loss_fwd_bwd_equals_the_two_passes(hip_lib, B, kind):
    """nplda_loss_fwd_bwd_f32 (one launch up to 4096 scores) gives the bits of loss_sums + loss_finish; a misaligned
    view takes the two-pass route inside the library."""
    from neuralplda_amd import ops
    rng = np.random.default_rng(B)
    s = torch.from_numpy((rng.standard_normal(B + 1) * 2).astype(np.float32)).cuda()
    t = torch.from_numpy((rng.random(B + 1) < 0.3).astype(np.float32)).cuda()
    t[0], t[-1] = 1.0, 0.0
    k = ops.LOSS_SOFTCDET if kind == "SoftCdet" else ops.LOSS_BCE
    ths = [torch.tensor([-0.4], device="cuda"), torch.tensor([0.2], device="cuda")] if k == ops.LOSS_SOFTCDET \
        else [torch.tensor([0.1], device="cuda")]
    betas = [99.0, 199.0] if k == ops.LOSS_SOFTCDET else []
    for sv, tv in ((s[:B], t[:B]), (s[1:], t[1:])):  # aligned, then offset by 4 bytes
        if tv.sum() == 0 or tv.sum() == B:
            continue
        sums = ops.loss_sums(sv, tv, ths, 15.0, k)
        L, g, dth = ops.loss_finish(sv, tv, ths, betas, 15.0, k, sums)
        L2, g2, dth2, sums2 = ops.loss_fwd_bwd(sv, tv, ths, betas, 15.0, k)
        if sv.data_ptr() % 16 == 0 and B <= 4096:  # single-block forms on both sides: deterministic, same bits
            assert torch.equal(sums, sums2) and torch.equal(L, L2) and torch.equal(g, g2) and torch.equal(dth, dth2)
        else:  # multi-block sums meet through fp64 atomics: equal up to the order of those additions
            torch.testing.assert_close(sums, sums2, rtol=1e-12, atol=1e-12)
            torch.testing.assert_close(L, L2, rtol=1e-6, atol=1e-7)
            torch.testing.assert_close(g, g2, rtol=1e-5, atol=1e-9)
            torch.testing.assert_close(dth, dth2, rtol=1e-5, atol=1e-9)


def test_reference_driver_train_and_validate_g12(hip_lib, tmp_path):
    """G12: the reference's OWN train() / validate() (xvector_NeuralPlda_pytorch.py:30-83) were run on a tiny seeded
    set (tests/golden/make_golden_r2.py); this build's train() / validate() on the same files under the same seeds must
    reproduce the threshold initialisation, every batch loss of the epoch, the final state dict and the validation
    metrics — with the autograd step and with the fused graph-replayed step."""
    from neuralplda_amd import models, train
    from neuralplda_amd.sv_trials_loaders import combine_trials_and_get_loader, get_trials_loaders_dict
    g = np.load(os.path.join(G, "g12_reference_driver.npz"))
    utt_ids = [str(u) for u in g["utt_ids"]]
    mega = {u: g["xvec"][i] for i, u in enumerate(utt_ids)}
    num_to_id = {i: u for i, u in enumerate(utt_ids)}
    id_to_num = {u: i for i, u in enumerate(utt_ids)}
    trf, vaf = str(tmp_path / "train_trials.tsv"), str(tmp_path / "val_trials.tsv")
    open(trf, "w").write(str(g["train_trials_text"]))
    open(vaf, "w").write(str(g["val_trials_text"]))
    seeds = [int(v) for v in g["seeds"]]

    class Conf:
        log_interval, loss, beta = 1, "SoftCdet", [99.0, 199.0]

    for fused in (False, True):
        np.random.seed(seeds[0])
        torch.manual_seed(seeds[0])
        m = models.NeuralPlda(NC(64, 24, 20))
        sd = m.state_dict()
        for k in g["keys"]:  # the constructor draws the same RNG values as the reference's; pin them anyway
            np.testing.assert_allclose(sd[str(k)].numpy(), g["p0_" + str(k)], atol=0, rtol=0)
        train_loader = combine_trials_and_get_loader([trf], id_to_num, subsample_factors=[1.01],
                                                     batch_size=int(g["batch_size"]))
        valid = get_trials_loaders_dict([vaf], id_to_num, subsample_factors=[1.01], batch_size=int(g["val_batch_size"]))
        assert list(valid.keys()) == [str(g["val_key"])]
        m = m.cuda()
        torch.manual_seed(seeds[1])
        minc0, th0 = train.validate(Conf, m, "cuda", mega, num_to_id, valid[str(g["val_key"])], update_thresholds=True)
        assert abs(float(minc0) - float(g["minc0"])) <= 1e-6
        np.testing.assert_allclose([float(th0[99.0]), float(th0[199.0])], g["th0"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose([m.Th99.item(), m.Th199.item()], g["th_init"], rtol=1e-5, atol=1e-6)
        losses = []
        if fused:
            step = train.FusedTrainStep(m, float(g["lr"]), weight_decay=float(g["weight_decay"]),
                                        batch_size=int(g["batch_size"]), graph=True)
            inner = step.step_rows

            def rec_rows(*a, **kw):
                L = inner(*a, **kw)
                losses.append(float(L))
                return L
            step.step_rows = rec_rows
            inner_rec = step.step_record

            def rec_record():  # (the device-resident epoch of train(): records walked by the captured step)
                L = inner_rec()
                losses.append(float(L))
                return L
            step.step_record = rec_record
            opt = None
        else:
            step = None
            opt = torch.optim.Adam(m.parameters(), lr=float(g["lr"]), weight_decay=float(g["weight_decay"]))
            orig = m.loss

            def rec(o, t_):
                L = orig(o, t_)
                losses.append(float(L))
                return L
            m.loss = rec
        torch.manual_seed(seeds[2])
        train.train(Conf, m, "cuda", train_loader, mega, num_to_id, opt, 1, step_fn=step)
        if not fused:
            m.loss = orig
        np.testing.assert_allclose(losses, g["losses"], rtol=5e-5)
        for k in g["keys"]:
            k = str(k)
            got = m.state_dict()[k].cpu().numpy()
            disp_ref = g["p1_" + k] - g["p0_" + k] if not k.startswith("Th") else g["p1_" + k] - g["th_init"][0 if k == "Th99" else 1]
            ref = g["p1_" + k]
            scale = max(np.abs(disp_ref).max(), 1e-12)
            # ten Adam steps at lr 1e-3; the reference's fp32 autograd noise bounds the agreement (2 % of the displacement)
            assert np.abs(got - ref).max() <= 0.02 * scale + 1e-6, (k, np.abs(got - ref).max() / scale)
        torch.manual_seed(seeds[3])
        minc1, th1 = train.validate(Conf, m, "cuda", mega, num_to_id, valid[str(g["val_key"])])
        assert abs(float(minc1) - float(g["minc1"])) <= 1e-3  # the north-star minDCF tolerance
        np.testing.assert_allclose([float(th1[99.0]), float(th1[199.0])], g["th1"], atol=2e-3)


def test_table_from_a_100k_vector_archive_trains_on_the_device(hip_lib, tmp_path):
    """SURVEY f2 end to end on the GPU box: 100 000 x-vectors in a binary Kaldi archive (the bytes assembled here with
    struct: key, blank, \\0B 'FV ' \\x04 int32 dim, floats — not by kaldi_format's writer) -> XvectorTable.from_ark (one pinned
    host matrix) -> .on('cuda') -> one FusedTrainStep.step_rows on pairs named by table rows.  The device copy must hold
    the archive's bits, and the step must equal the same step fed gathered (B, 512) tensors."""
    import struct
    from neuralplda_amd import ops, train
    from neuralplda_amd import sv_trials_loaders as svl
    rng = np.random.default_rng(17)
    N, D0, D, B = 100000, 512, 150, 2048
    M = rng.standard_normal((N, D0), dtype=np.float32)
    path = str(tmp_path / "xvector.1.ark")
    hdr = b"\0BFV \x04" + struct.pack("<i", D0)
    with open(path, "wb") as fh:
        for lo in range(0, N, 10000):
            fh.write(b"".join(b"utt%07d " % i + hdr + M[i].tobytes() for i in range(lo, min(N, lo + 10000))))
    tab = svl.XvectorTable.from_ark(path)
    assert len(tab) == N and tab.dim == D0 and tab.ids[12345] == "utt0012345"
    dev = tab.on("cuda")
    assert dev.is_cuda and dev.shape == (N, D0)
    torch.cuda.synchronize()
    assert torch.equal(dev[::997].cpu(), torch.from_numpy(M[::997])) and torch.equal(dev[-1].cpu(), torch.from_numpy(M[-1]))
    p = rand_params(rng, D0, D, D)
    nc = NC(D0, D, D)
    m_a, m_b = model_from(p, nc, thetas=[-0.5, -0.3]), model_from(p, nc, thetas=[-0.5, -0.3])
    sa = train.FusedTrainStep(m_a, 1e-3, weight_decay=1e-5, batch_size=B, graph=True)
    sb = train.FusedTrainStep(m_b, 1e-3, weight_decay=1e-5, batch_size=B, graph=False)
    r1 = torch.from_numpy(rng.integers(0, N, B)).cuda()
    r2 = torch.from_numpy(rng.integers(0, N, B)).cuda()
    t = torch.from_numpy((rng.random(B) < 0.2).astype(np.float32)).cuda()
    la = sa.step_rows(dev, r1, r2, t)
    lb = sb(ops.gather_rows(dev, r1), ops.gather_rows(dev, r2), t)
    assert la.item() == lb.item()
    for (k, a), (_, b) in zip(m_a.state_dict().items(), m_b.state_dict().items()):
        assert torch.equal(a, b), k


@pytest.mark.parametrize("lossname,B,D", [("SoftCdet", 4096, 150), ("crossentropy", 1000, 170), ("SoftCdet", 250, 40)])
def test_one_collective_dp_step_on_one_rank_equals_the_one_call_step(hip_lib, lossname, B, D):
    """nplda_train_step_grad_f32 -> (all-reduce) -> nplda_train_step_apply_f32 with identity reductions = one rank: the SAME
    parameter bits as nplda_train_step_f32 (same slabs, same update arithmetic); loss and thresholds go through the 16-bit
    limbs of the fp64 loss sums (2^-41 absolute)."""
    from neuralplda_amd import ops, train
    rng = np.random.default_rng(99)
    p = rand_params(rng, 512, D, D)
    xs = [(torch.from_numpy(rng.standard_normal((B, 512)).astype(np.float32)).cuda(),
           torch.from_numpy(rng.standard_normal((B, 512)).astype(np.float32)).cuda(),
           torch.from_numpy((rng.random(B) < 0.2).astype(np.float32)).cuda()) for _ in range(3)]
    nc = NC(512, D, D, loss=lossname)
    m_a = model_from(p, nc, thetas=[-0.5, -0.3], theta_xent=0.1)
    m_b = model_from(p, nc, thetas=[-0.5, -0.3], theta_xent=0.1)
    calls = []
    m_b._reduce_sums = lambda t: t
    m_b._reduce_flat = lambda t: (calls.append(t.numel()), t)[1]
    for graph in (False, True):
        sa = train.FusedTrainStep(m_a, 1e-3, weight_decay=1e-5, batch_size=B, graph=graph)
        sb = train.FusedTrainStep(m_b, 1e-3, weight_decay=1e-5, batch_size=B, graph=graph)
        assert sa._one_call and sb._dp_call and not sb._one_call
        for i, (x1, x2, t) in enumerate(xs):
            nt = float(t.sum().item())
            la = sa(x1, x2, t)
            lb = sb(x1, x2, t, global_counts=(nt, B - nt)) if i != 1 else sb(x1, x2, t)  # (i == 1: counts by the fallback)
            assert abs(la.item() - lb.item()) <= 1e-6 * abs(la.item())
            for (k, a), (_, b) in zip(m_a.state_dict().items(), m_b.state_dict().items()):
                if k.startswith("Th") or k.startswith("threshold"):
                    assert torch.allclose(a, b, rtol=1e-6, atol=1e-9), (graph, i, k)
                else:
                    assert torch.equal(a, b), (graph, i, k)
            fresh = ops.pack_params(*[q.detach() for q in sb.params])
            assert torch.equal(sb._packed.buf, fresh.buf), (graph, i)
        assert abs(sa.pop_loss_mean() - sb.pop_loss_mean()) < 1e-6
    assert calls and all(n == ops.train_step_flat_floats(sb._packed) for n in calls)  # ONE reduction per step, of this size


def test_one_collective_dp_step_two_shards_sum_to_the_global_step(hip_lib):
    """Two ranks' shards of a 4096-pair minibatch, emulated on one GPU: each shard's gradient phase with the GLOBAL counts,
    the two flat buffers added (what the all-reduce does), one update phase — against nplda_train_step_f32 on the whole
    minibatch.  (dL/ds needs only the global counts: no collective in front of the backward.)"""
    from neuralplda_amd import ops, train
    rng = np.random.default_rng(5)
    B, D = 4096, 150
    p = rand_params(rng, 512, D, D)
    x1 = torch.from_numpy(rng.standard_normal((B, 512)).astype(np.float32)).cuda()
    x2 = torch.from_numpy(rng.standard_normal((B, 512)).astype(np.float32)).cuda()
    t = torch.from_numpy((rng.random(B) < 0.15).astype(np.float32)).cuda()
    nc = NC(512, D, D)
    m_a = model_from(p, nc, thetas=[-0.5, -0.3])
    m_b = model_from(p, nc, thetas=[-0.5, -0.3])
    sa = train.FusedTrainStep(m_a, 1e-3, weight_decay=1e-5, batch_size=B, graph=False)
    la = sa(x1, x2, t)
    # the two "ranks": same parameters, shards [0, 1500) and [1500, 4096) (unequal on purpose)
    sb = train.FusedTrainStep(m_b, 1e-3, weight_decay=1e-5, batch_size=B, graph=False)
    sb._sync_packed()
    prm, ths = [q.detach() for q in sb.params], [th.detach() for th in sb.thetas]
    nt = float(t.sum().item())
    gc = torch.tensor([nt, B - nt], dtype=torch.float64, device="cuda")
    nflat = ops.train_step_flat_floats(sb._packed)
    flats = []
    for lo, hi in ((0, 1500), (1500, B)):
        step_r = sb.step_count.clone()  # every rank counts its own step
        ws = ops.train_step_workspace(hi - lo, sb._packed)
        f = torch.zeros(nflat, device="cuda")
        ops.train_step_grad(x1[lo:hi], x2[lo:hi], t[lo:hi].contiguous(), prm, ths, sb.betas_loss, sb.alpha, sb.kind, step_r,
                            sb._packed, ws, f, gc)
        flats.append(f)
    sb.step_count.copy_(step_r)
    flat = flats[0] + flats[1]
    loss = torch.zeros((), device="cuda")
    ops.train_step_apply(flat, prm, ths, sb.betas_loss, sb.alpha, sb.kind, sb.m, sb.v, sb.step_count, sb.lr, sb.betas[0],
                         sb.betas[1], sb.eps, sb.wd, sb._packed, loss)
    assert abs(la.item() - loss.item()) <= 1e-6 * abs(la.item())
    for (k, a), (_, b) in zip(m_a.state_dict().items(), m_b.state_dict().items()):
        # Adam's first step is lr * sign-like: compare the applied update, not just the parameter
        assert torch.allclose(a, b, rtol=1e-5, atol=2e-6), k


def test_train_loop_data_parallel_form_on_one_rank(hip_lib, tmp_path):
    """train() with a data-parallel model: the loader hands each rank its slice of every global batch and the global
    label counts (TrialLoader.device_batches(shard=...)), the step runs its one-collective form.  With one rank (identity
    reductions) the epoch must leave the parameters of the plain run, bit for bit, and print the same progress lines."""
    import contextlib
    import io
    from neuralplda_amd import train
    rng = np.random.default_rng(6)
    B = 128
    mega, num_to_id, loader = _tiny_trial_set(tmp_path, rng, B)
    p = rand_params(rng, 512, 150, 150)
    nc = NC(D1=150, D2=150, loss="SoftCdet")
    nc.log_interval = 3

    def run(dp):
        m = model_from(p, nc, thetas=[-0.5, -0.3])
        if dp:
            m._reduce_sums = lambda t: t
            m._reduce_flat = lambda t: t
        step = train.FusedTrainStep(m, 1e-3, weight_decay=1e-5, batch_size=B, graph=True)
        assert step._dp_call == dp
        torch.manual_seed(11)
        out = io.StringIO()
        with contextlib.redirect_stdout(out):
            train.train(nc, m, torch.device("cuda"), loader, mega, num_to_id, None, 1, step_fn=step)
        return {k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}, out.getvalue(), step

    sd_plain, log_plain, _ = run(False)
    sd_dp, log_dp, step_dp = run(True)
    assert step_dp._graph_rows is not None and step_dp.step_count[0].item() == 8
    assert log_dp == log_plain
    for k in sd_plain:
        if k.startswith("Th"):
            np.testing.assert_allclose(sd_dp[k], sd_plain[k], rtol=1e-6, atol=1e-9)
        else:
            assert np.array_equal(sd_dp[k], sd_plain[k]), k


def test_step_records_several_steps_per_graph_launch(hip_lib):
    """step_records(n): records_per_replay steps in one captured graph (each step's last kernel stages the next record, so
    they chain on the device) + single steps for the rest — the same parameter bits and loss mean as n x step_record()."""
    from neuralplda_amd import train
    rng = np.random.default_rng(17)
    B, nb, N = 256, 21, 3000
    p = rand_params(rng, 512, 150, 150)
    nc = NC(D1=150, D2=150, loss="SoftCdet")
    table = torch.from_numpy(rng.standard_normal((N, 512)).astype(np.float32)).cuda()
    records = torch.empty((nb, 20 * B), dtype=torch.uint8, device="cuda")
    records[:, :8 * B].view(torch.int64).copy_(torch.from_numpy(rng.integers(0, N, (nb, B))).cuda())
    records[:, 8 * B:16 * B].view(torch.int64).copy_(torch.from_numpy(rng.integers(0, N, (nb, B))).cuda())
    records[:, 16 * B:].view(torch.float32).copy_(torch.from_numpy((rng.random((nb, B)) < 0.2).astype(np.float32)).cuda())
    m_a, m_b = model_from(p, nc, thetas=[-0.5, -0.3]), model_from(p, nc, thetas=[-0.5, -0.3])
    sa = train.FusedTrainStep(m_a, 1e-3, weight_decay=1e-5, batch_size=B, graph=True)
    sb = train.FusedTrainStep(m_b, 1e-3, weight_decay=1e-5, batch_size=B, graph=True)
    assert sa.records_per_replay == 8
    sa.begin_epoch(table, records)
    sb.begin_epoch(table, records)
    assert sa._graph_rec_multi is not None
    la = sa.step_records(19)  # 8 + 8 + 3 singles
    for _ in range(19):
        lb = sb.step_record()
    assert la.item() == lb.item() and sa._records_left == sb._records_left == 2
    assert sa._cursor[1].item() == sb._cursor[1].item() and sa.step_count[0].item() == 19
    for (key, a), (_, b) in zip(m_a.state_dict().items(), m_b.state_dict().items()):
        assert torch.equal(a, b), key
    assert abs(sa.pop_loss_mean() - sb.pop_loss_mean()) < 1e-12
    with pytest.raises(RuntimeError):
        sa.step_records(3)
    sa.step_records(2)
    sb.step_record(); sb.step_record()
    assert torch.equal(next(iter(m_a.state_dict().values())), next(iter(m_b.state_dict().values())))


def test_validate_scores_in_kernel_sized_chunks(hip_lib, monkeypatch):
    """validate()'s dense device-resident pass (NPLDA_VALIDATE_DENSE=1, or a list with few repeats per utterance) scores the trial list in chunks sized for the kernels (c x 4096 pairs, gather folded
    into the balanced-tile kernel), not in the loader's batches.  Over a list longer than one chunk the metrics must agree
    with the loader-batched scoring of the same pairs (another kernel regime: same scores to the forward tolerance)."""
    import contextlib
    import io
    from neuralplda_amd import metrics, sv_trials_loaders as svl, train
    monkeypatch.setenv("NPLDA_VALIDATE_DENSE", "1")
    rng = np.random.default_rng(12)
    n_utt, n = 5000, 250000
    ids = [f"u{i:05d}" for i in range(n_utt)]
    spk = rng.integers(0, 200, n_utt)
    cent = rng.standard_normal((200, 512)).astype(np.float32)
    mat = (cent[spk] + 0.7 * rng.standard_normal((n_utt, 512))).astype(np.float32)
    mega = svl.XvectorTable.from_matrix(ids, mat)
    num_to_id = dict(enumerate(ids))
    a, b = rng.integers(0, n_utt, n), rng.integers(0, n_utt, n)
    lab = (spk[a] == spk[b]).astype(np.float32)
    ds = svl.TrialIndexDataset(torch.from_numpy(a), torch.from_numpy(b), torch.from_numpy(lab))
    loader = svl._loader(ds, 5 * 2048)
    nc = NC(D1=150, D2=150, loss="SoftCdet")
    nc.batch_size = 2048
    m = model_from(rand_params(rng, 512, 150, 150), nc, thetas=[-0.5, -0.3])
    assert train._validate_chunk(m) > 5 * 2048 and train._validate_chunk(m) < n  # more than one chunk, none loader-sized
    with contextlib.redirect_stdout(io.StringIO()):
        mc, th = train.validate(nc, m, torch.device("cuda"), mega, num_to_id, loader)
    X = torch.from_numpy(mat).cuda()
    ia, ib = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    with torch.no_grad():
        s = torch.cat([m(X[ia[lo:lo + 10240]], X[ib[lo:lo + 10240]]) for lo in range(0, n, 10240)])
    mc_ref, th_ref = metrics.minc(s, torch.from_numpy(lab).cuda(), nc.beta)
    assert abs(float(mc) - float(mc_ref)) <= 1e-4  # (the arg-min thresholds may sit on different scores where the cost is flat)


@pytest.mark.parametrize("lossname", ["SoftCdet", "crossentropy"])
@pytest.mark.parametrize("D,B", [(150, 2048), (150, 8), (150, 3), (150, 1003), (170, 2048), (170, 333), (170, 2), (150, 4096)])
def test_one_call_step_gradient_and_loss_against_the_fp64_oracle(hip_lib, lossname, D, B):
    """nplda_train_step_f32 on its own against the fp64 ORACLE (not against another kernel of this build): the loss, dL/dtheta
    and the applied flat gradient of one step.  B <= 2048 runs the 8-pair half-tile kernel (nplda_train_fb_half.h: the rows of
    a pair in one MFMA group, DPP cross terms), 4096 the 16-pair kernel; 8 = one half tile, 1003 / 333 = ragged last tiles.
    Tolerances: loss 1e-5 relative, gradients 1e-4 of the per-tensor max-abs."""
    from neuralplda_amd import ops
    rng = np.random.default_rng(1000 * D + B)
    p = rand_params(rng, 512, D, D)
    prm = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in p.tensors()]
    x1 = rng.standard_normal((B, 512)).astype(np.float32)
    x2 = rng.standard_normal((B, 512)).astype(np.float32)
    t = (rng.random(B) < 0.3).astype(np.float32)
    t[0], t[-1] = 1.0, 0.0  # both classes present
    kind = ops.LOSS_SOFTCDET if lossname == "SoftCdet" else ops.LOSS_BCE
    theta = [-0.4, -0.2] if kind == ops.LOSS_SOFTCDET else [0.1]
    betas, alpha = ([99.0, 199.0], 15.0) if kind == ops.LOSS_SOFTCDET else ([], 0.0)
    ths = [torch.tensor([v], device="cuda") for v in theta]
    packed = ops.pack_params(*prm)
    n = int(sum(q.numel() for q in prm))
    K = len(ths)
    m, v, step = torch.zeros(n + K, device="cuda"), torch.zeros(n + K, device="cuda"), torch.zeros(2, device="cuda")
    out, lbuf = torch.zeros(n + K, device="cuda"), torch.zeros((), device="cuda")
    ws = ops.train_step_workspace(B, packed)
    ops.train_step(torch.from_numpy(x1).cuda(), torch.from_numpy(x2).cuda(), torch.from_numpy(t).cuda(), prm, ths, betas, alpha,
                   kind, m, v, step, 1e-3, 0.9, 0.999, 1e-8, 1e-5, packed, ws, lbuf, grad_out=out)
    s_ref = orc.forward(x1, x2, p, np.float64)
    if kind == ops.LOSS_SOFTCDET:
        L_ref = orc.softcdet(s_ref, t, theta, betas, alpha, np.float64)
        g_ref, dth_ref = orc.softcdet_grad(s_ref, t, theta, betas, alpha)
    else:
        L_ref = orc.crossentropy(s_ref, t, theta[0], np.float64)
        g_ref, dth_ref = orc.crossentropy_grad(s_ref, t, theta[0])
        dth_ref = np.atleast_1d(dth_ref)
    assert abs(lbuf.item() - L_ref) <= 1e-5 * abs(L_ref), (lbuf.item(), L_ref)
    ref = orc.backward(x1, x2, g_ref, p)
    got = [a.cpu().numpy() for a in ops.split_flat_grad(out[:n], 512, D, D)]
    for name, a in zip(("W1", "b1", "W2", "b2", "P_sqrt", "Q"), got):
        assert relmax(a, ref[name]) <= 1e-4, (name, relmax(a, ref[name]))
    np.testing.assert_allclose(out[n:].cpu().numpy(), np.asarray(dth_ref, np.float64), rtol=2e-4, atol=1e-7)
    assert step[0].item() == 1.0
