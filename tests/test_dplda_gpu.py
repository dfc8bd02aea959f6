"""GPU parity of DPlda.forward (utils/models.py:463-495) evaluated as a quadratic form by the fused MODE_GB kernel,
against reference outputs (G10) and the oracle.  Tolerance: |ds| <= 2e-5 + 2e-5 |s| (fp32 MFMA vs fp32/fp64 BLAS)."""
import os
import pickle

import numpy as np
import pytest
import torch

from oracle import nplda_oracle as orc

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class NC:
    def __init__(self, D0, D1, beta=(99.0, 199.0)):
        self.xvector_dim, self.layer1_LDA_dim, self.layer2_PLDA_spkfactor_dim = D0, D1, D1
        self.beta, self.alpha, self.device, self.loss = list(beta), 15.0, "cuda", "SoftCdet"


def make(D0, D1, W1, b1, wlr, blr):
    from neuralplda_amd import models
    m = models.DPlda(NC(D0, D1))
    with torch.no_grad():
        m.centering_and_LDA.weight.copy_(torch.from_numpy(np.asarray(W1)))
        m.centering_and_LDA.bias.copy_(torch.from_numpy(np.asarray(b1)))
        m.logistic_regres.weight.copy_(torch.from_numpy(np.asarray(wlr)))
        m.logistic_regres.bias.copy_(torch.from_numpy(np.asarray(blr, dtype=np.float32).reshape(1)))
    return m.cuda()


@pytest.fixture(autouse=True)
def _no_grad_by_default():
    # scoring tests run like the reference's validate()/generate_scorefile (torch.no_grad); the training tests re-enable
    with torch.no_grad():
        yield


def test_dplda_golden_small(hip_lib):
    g = np.load(os.path.join(G, "g10_dplda_small.npz"), allow_pickle=True)
    m = make(64, 24, g["W1"], g["b1"], g["wlr"], g["blr"])
    assert set(m.state_dict().keys()) == set(str(k) for k in g["state_dict_keys"])
    x1, x2 = torch.from_numpy(g["x1"]).cuda(), torch.from_numpy(g["x2"]).cuda()
    s = m(x1, x2).cpu().numpy()
    np.testing.assert_allclose(s, g["s"], atol=2e-5, rtol=2e-5)
    y1 = m.extract_plda_embeddings(x1)
    np.testing.assert_allclose(y1.cpu().numpy(), g["y1"], atol=1e-6, rtol=1e-5)
    sfe = m.forward_from_plda_embeddings(y1, m.extract_plda_embeddings(x2)).cpu().numpy()
    np.testing.assert_allclose(sfe, g["s_from_emb"], atol=2e-5, rtol=2e-5)
    # embeddings are used as given (no re-normalisation): scale them and compare with the oracle
    y2 = m.extract_plda_embeddings(x2)
    ref = orc.dplda_from_embeddings(1.5 * y1.cpu().numpy(), 0.5 * y2.cpu().numpy(), g["wlr"], g["blr"], np.float64)
    got = m.forward_from_plda_embeddings(1.5 * y1, 0.5 * y2).cpu().numpy()
    np.testing.assert_allclose(got, ref, atol=2e-5, rtol=2e-5)


def test_dplda_kaldi170_golden(hip_lib):
    g1 = np.load(os.path.join(G, "g1_kaldi_params.npz"))
    f = np.load(os.path.join(G, "g2_forward_kaldi170.npz"))
    k = np.load(os.path.join(G, "g10_dplda_kaldi170.npz"))
    rg = np.random.default_rng(int(k["seed"]))
    wlr = (rg.standard_normal((1, 2 * 170 * 170 + 170)) * 0.05).astype(np.float32)
    m = make(512, 170, g1["W1"], g1["b1"], wlr, [0.125])
    s = m(torch.from_numpy(f["x1"]).cuda(), torch.from_numpy(f["x2"]).cuda()).cpu().numpy()
    np.testing.assert_allclose(s, k["s"], atol=2e-5, rtol=2e-5)
    y = m.extract_plda_embeddings(torch.from_numpy(f["x1"]).cuda())
    s2 = m.forward_from_plda_embeddings(y, m.extract_plda_embeddings(torch.from_numpy(f["x2"]).cuda())).cpu().numpy()
    np.testing.assert_allclose(s2, k["s"], atol=2e-5, rtol=2e-5)  # D1 = 170 exercises the zero-padded identity layer


@pytest.mark.parametrize("B", [0, 1, 17, 16385, 70001])
def test_dplda_vs_oracle_sizes(hip_lib, B):
    rg = np.random.default_rng(B + 5)
    D0, D1 = 128, 40
    W1 = (rg.standard_normal((D1, D0)) / np.sqrt(D0)).astype(np.float32)
    b1 = (0.1 * rg.standard_normal(D1)).astype(np.float32)
    wlr = (0.2 * rg.standard_normal((1, 2 * D1 * D1 + D1))).astype(np.float32)
    m = make(D0, D1, W1, b1, wlr, [-0.3])
    x1 = rg.standard_normal((B, D0)).astype(np.float32)
    x2 = rg.standard_normal((B, D0)).astype(np.float32)
    s = m(torch.from_numpy(x1).cuda(), torch.from_numpy(x2).cuda()).cpu().numpy()
    assert s.shape == (B,)
    n = min(B, 3000)
    if n:
        idx = rg.choice(B, n, replace=False)
        ref = orc.dplda_forward(x1[idx], x2[idx], W1, b1, wlr, [-0.3], np.float64)
        np.testing.assert_allclose(s[idx], ref, atol=2e-5, rtol=2e-5)


def test_dplda_losses_metrics_and_pickle(hip_lib, tmp_path):
    g = np.load(os.path.join(G, "g10_dplda_small.npz"), allow_pickle=True)
    m = make(64, 24, g["W1"], g["b1"], g["wlr"], g["blr"])
    s = m(torch.from_numpy(g["x1"]).cuda(), torch.from_numpy(g["x2"]).cuda())
    t = (torch.arange(50, device="cuda") % 3 == 0).float()
    sn, tn = s.cpu().numpy(), t.cpu().numpy()
    assert abs(float(m.softcdet(s, t)) - float(orc.softcdet(sn, tn, [0.0, 0.0], [99.0, 199.0], 15.0, np.float64))) < 1e-5
    assert abs(float(m.crossentropy(s, t)) - float(orc.crossentropy(sn, tn, 0.0, np.float64))) < 1e-6
    assert abs(float(m.cdet(s, t)) - float(orc.cdet(sn, tn, [0.0, 0.0], [99.0, 199.0]))) < 1e-6
    mc, th = m.minc(s, t, update_thresholds=True)
    ref_mc, ref_th = orc.minc_reference(sn, tn, [99.0, 199.0])
    assert abs(float(mc) - float(ref_mc)) < 1e-6
    assert abs(float(m.state_dict()["Th99"]) - float(ref_th[99.0])) < 1e-6
    p = tmp_path / "dplda.pt"
    m.SaveModel(str(p))
    m2 = pickle.load(open(p, "rb"))
    s2 = m2(torch.from_numpy(g["x1"]).cuda(), torch.from_numpy(g["x2"]).cuda())
    assert torch.equal(s, s2)


def _freeze_lda(m):
    m.centering_and_LDA.weight.requires_grad = False
    m.centering_and_LDA.bias.requires_grad = False


def test_dplda_gradients_golden(hip_lib):
    """Gradient of the linear unit + thresholds against the reference's autograd (G10 grad, fp64 re-evaluation)."""
    g = np.load(os.path.join(G, "g10_dplda_small.npz"), allow_pickle=True)
    gg = np.load(os.path.join(G, "g10_dplda_grad.npz"))
    m = make(64, 24, g["W1"], g["b1"], g["wlr"], g["blr"])
    _freeze_lda(m)
    with torch.no_grad():
        m.threshold[99.0].fill_(float(gg["theta"][0]))
        m.threshold[199.0].fill_(float(gg["theta"][1]))
    x1, x2 = torch.from_numpy(g["x1"]).cuda(), torch.from_numpy(g["x2"]).cuda()
    t = torch.from_numpy(gg["t"]).cuda()
    for lossname in ("SoftCdet", "crossentropy"):
        m.lossfn = lossname
        m.zero_grad()
        with torch.enable_grad():
            L = m.loss(m(x1, x2), t)
            L.backward()
        ref = {k: gg[f"{lossname}_f64_{k}"] for k in ("L", "dwlr", "dblr")}
        assert abs(float(L) - float(ref["L"])) < 2e-5 * max(1.0, abs(float(ref["L"])))
        mag = np.abs(ref["dwlr"]).max()
        np.testing.assert_allclose(m.logistic_regres.weight.grad.cpu().numpy(), ref["dwlr"], atol=2e-5 * mag, rtol=2e-4)
        np.testing.assert_allclose(m.logistic_regres.bias.grad.cpu().numpy(), ref["dblr"], rtol=2e-4)
        if lossname == "SoftCdet":
            for b in (99, 199):
                np.testing.assert_allclose(m.threshold[float(b)].grad.cpu().numpy(), gg[f"SoftCdet_f64_dTh{b}"], rtol=2e-4)
        assert m.centering_and_LDA.weight.grad is None


def test_dplda_unfrozen_lda_trains(hip_lib):
    """With the LDA left trainable (not the reference's recipe, but what its autograd allows) the backward reaches
    centering_and_LDA too; golden check in tests/test_input_grads_gpu.py::test_dplda_input_and_lda_grads_golden."""
    g = np.load(os.path.join(G, "g10_dplda_small.npz"), allow_pickle=True)
    m = make(64, 24, g["W1"], g["b1"], g["wlr"], g["blr"])
    m(torch.from_numpy(g["x1"]).cuda(), torch.from_numpy(g["x2"]).cuda())  # scoring needs no freeze
    with torch.enable_grad():
        s = m(torch.from_numpy(g["x1"]).cuda(), torch.from_numpy(g["x2"]).cuda())
        s.sum().backward()
    assert m.centering_and_LDA.weight.grad is not None and torch.isfinite(m.centering_and_LDA.weight.grad).all()
    assert m.logistic_regres.weight.grad is not None


def test_dplda_adam_steps_match_torch_recipe(hip_lib):
    """Three steps of the DPlda recipe (xvector_DPlda_pytorch.py:140-147: Adam on logistic_regres + thresholds, LDA frozen)
    against the same recipe built from plain torch ops on the GPU (fp32 autograd reference of the same arithmetic)."""
    rg = np.random.default_rng(3)
    D0, D1, B = 128, 40, 2048
    W1 = (rg.standard_normal((D1, D0)) / np.sqrt(D0)).astype(np.float32)
    b1 = (0.1 * rg.standard_normal(D1)).astype(np.float32)
    wlr = (0.05 * rg.standard_normal((1, 2 * D1 * D1 + D1))).astype(np.float32)
    m = make(D0, D1, W1, b1, wlr, [0.0])
    _freeze_lda(m)
    ref_w = torch.from_numpy(wlr).cuda().requires_grad_()
    ref_b = torch.zeros(1, device="cuda", requires_grad=True)
    ref_th = [torch.zeros(1, device="cuda", requires_grad=True) for _ in range(2)]
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, weight_decay=1e-5)
    ropt = torch.optim.Adam([ref_w, ref_b] + ref_th, lr=1e-3, weight_decay=1e-5)
    W1t, b1t = torch.from_numpy(W1).cuda(), torch.from_numpy(b1).cuda()
    for step in range(3):
        x1 = torch.from_numpy(rg.standard_normal((B, D0)).astype(np.float32)).cuda()
        x2 = torch.from_numpy(rg.standard_normal((B, D0)).astype(np.float32)).cuda()
        t = torch.from_numpy((rg.random(B) < 0.2).astype(np.float32)).cuda()
        opt.zero_grad()
        with torch.enable_grad():
            L = m.loss(m(x1, x2), t)
            L.backward()
        opt.step()
        ropt.zero_grad()
        torch.set_grad_enabled(True)
        y1 = torch.nn.functional.normalize(x1 @ W1t.T + b1t)
        y2 = torch.nn.functional.normalize(x2 @ W1t.T + b1t)
        n = D1 * D1
        Wb, Ww, ws = ref_w[0, :n].reshape(D1, D1), ref_w[0, n:2 * n].reshape(D1, D1), ref_w[0, 2 * n:]
        s = ((y1 @ Wb) * y2).sum(1) + ((y2 @ Wb) * y1).sum(1) \
            + ((y1 @ Ww) * y1).sum(1) + ((y2 @ Ww) * y2).sum(1) + (y1 + y2) @ ws + ref_b
        sig = torch.sigmoid
        Lr = sum((sig(15.0 * (th - s)) * t).sum() / t.sum() + b * (sig(15.0 * (s - th)) * (1 - t)).sum() / (1 - t).sum()
                 for th, b in zip(ref_th, (99.0, 199.0))) / 2
        Lr.backward()
        torch.set_grad_enabled(False)
        ropt.step()
        assert abs(float(L) - float(Lr)) < 1e-4 * max(1.0, abs(float(Lr)))
    # Adam's first steps move every weight by ~lr regardless of gradient scale: compare the trajectories loosely
    # where the gradient is tiny, tightly in aggregate
    dw = (m.logistic_regres.weight.detach() - ref_w.detach()).abs()
    assert float(dw.mean()) < 2e-5 and float(dw.max()) < 2.1e-3
    for b, th in zip((99.0, 199.0), ref_th):
        assert abs(float(m.threshold[b]) - float(th)) < 1e-5


@pytest.mark.parametrize("graph", [False, True])
def test_fused_dplda_step_follows_the_autograd_recipe(hip_lib, graph):
    """train.FusedDPldaStep (direct launches + one-launch Adam, optionally a HIP-graph replay) against the autograd +
    torch.optim.Adam recipe on models.DPlda: same losses, same parameters after five steps."""
    from neuralplda_amd import train
    rg = np.random.default_rng(8)
    D0, D1, B = 128, 40, 512
    W1 = (rg.standard_normal((D1, D0)) / np.sqrt(D0)).astype(np.float32)
    b1 = (0.1 * rg.standard_normal(D1)).astype(np.float32)
    wlr = (0.05 * rg.standard_normal((1, 2 * D1 * D1 + D1))).astype(np.float32)
    batches = [(torch.from_numpy(rg.standard_normal((B, D0)).astype(np.float32)).cuda(),
                torch.from_numpy(rg.standard_normal((B, D0)).astype(np.float32)).cuda(),
                torch.from_numpy((rg.random(B) < 0.2).astype(np.float32)).cuda()) for _ in range(5)]
    m_ref = make(D0, D1, W1, b1, wlr, [0.0])
    _freeze_lda(m_ref)
    opt = torch.optim.Adam([p for p in m_ref.parameters() if p.requires_grad], lr=1e-3, weight_decay=1e-5)
    ref_losses = []
    for x1, x2, t in batches:
        opt.zero_grad()
        with torch.enable_grad():
            L = m_ref.loss(m_ref(x1, x2), t)
            L.backward()
        opt.step()
        ref_losses.append(float(L))
    m = make(D0, D1, W1, b1, wlr, [0.0])
    _freeze_lda(m)
    step = train.FusedDPldaStep(m, 1e-3, weight_decay=1e-5, batch_size=B, graph=graph)
    losses = [float(step(x1, x2, t)) for x1, x2, t in batches]
    np.testing.assert_allclose(losses, ref_losses, rtol=2e-5)
    for (k, a), (_, b) in zip(m.state_dict().items(), m_ref.state_dict().items()):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-4, atol=2e-6, err_msg=k)
    assert step.step_count[0].item() == 5
    m2 = make(D0, D1, W1, b1, wlr, [0.0])  # LDA not frozen: refused up front
    with pytest.raises(ValueError):
        train.FusedDPldaStep(m2, 1e-3, batch_size=B)


def test_dplda_device_resident_epoch_equals_the_generic_loop(hip_lib, tmp_path):
    """train() + FusedDPldaStep over the vectorised loader (epoch resident on the device, gathers inside the captured
    step) leaves the same parameters as the generic host-batch loop, bit for bit (ragged last batch included)."""
    import contextlib
    import io
    from neuralplda_amd import sv_trials_loaders as svl, train
    rng = np.random.default_rng(12)
    n_utt, n_trials, B, D0, D1 = 200, 700, 128, 128, 24
    ids = [f"u{u:04d}" for u in range(n_utt)]
    xv = rng.standard_normal((n_utt, D0)).astype(np.float32)
    mega = {u: xv[i] for i, u in enumerate(ids)}
    num_to_id = dict(enumerate(ids))
    id_to_num = {u: i for i, u in num_to_id.items()}
    a, b = rng.integers(0, n_utt, n_trials), rng.integers(0, n_utt, n_trials)
    lab = (rng.random(n_trials) < 0.2).astype(int)
    tf = tmp_path / "train.tsv"
    tf.write_text("\n".join(f"{ids[i]}\t{ids[j]}\t{l}" for i, j, l in zip(a, b, lab)) + "\n")
    loader = svl.combine_trials_and_get_loader([str(tf)], id_to_num, subsample_factors=[1.01], batch_size=B)
    W1 = (rng.standard_normal((D1, D0)) / np.sqrt(D0)).astype(np.float32)
    b1 = (0.1 * rng.standard_normal(D1)).astype(np.float32)
    wlr = (0.05 * rng.standard_normal((1, 2 * D1 * D1 + D1))).astype(np.float32)
    nc = NC(D0, D1)
    nc.log_interval = 2

    def run(fast):
        m = make(D0, D1, W1, b1, wlr, [0.0])
        _freeze_lda(m)
        step = train.FusedDPldaStep(m, 1e-3, weight_decay=1e-5, batch_size=B, graph=True)
        torch.manual_seed(4)
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
    assert step_fast._graph_rows is not None and step_fast.step_count[0].item() == 6  # 5 replays + 1 ragged eager step
    assert log_fast == log_gen
    for k in sd_gen:
        assert np.array_equal(sd_fast[k], sd_gen[k]), k


@pytest.mark.parametrize("D1,B", [(24, 1), (24, 100), (150, 2048), (170, 777), (96, 4099)])
def test_dplda_grad_one_call_equals_moments_then_fold(hip_lib, D1, B):
    """nplda_dplda_grad_f32 (moments + fold in one call) gives the bits of weighted_moments -> dplda_fold_grad; against
    numpy fp64 as well (d wlr = [G12 + G21 | G11 + G22 | s1 + s2], d bias = sum g: utils/models.py:484-490)."""
    from neuralplda_amd import ops
    rng = np.random.default_rng(5 * D1 + B)
    x = rng.standard_normal((B, 2 * D1)).astype(np.float32)
    g = (rng.standard_normal(B) / B).astype(np.float32)
    X, G = torch.from_numpy(x).cuda(), torch.from_numpy(g).cuda()
    dw, db = ops.dplda_grad(X, G, D1)
    dw0, db0 = ops.dplda_fold_grad(*ops.weighted_moments(X, G), D1)
    assert torch.equal(dw, dw0) and torch.equal(db, db0)
    M = np.einsum("k,ki,kj->ij", g.astype(np.float64), x.astype(np.float64), x.astype(np.float64))
    ref = np.concatenate([(M[:D1, D1:] + M[D1:, :D1]).ravel(), (M[:D1, :D1] + M[D1:, D1:]).ravel(),
                          (g[:, None].astype(np.float64) * x).sum(0)[:D1] + (g[:, None].astype(np.float64) * x).sum(0)[D1:]])
    np.testing.assert_allclose(dw.cpu().numpy().ravel(), ref, atol=2e-6 * max(1.0, np.abs(ref).max()), rtol=2e-5)
    np.testing.assert_allclose(db.cpu().numpy(), [g.astype(np.float64).sum()], atol=1e-6)


@pytest.mark.parametrize("D1", [24, 150, 170])
def test_dplda_quadform_image_one_launch(hip_lib, D1):
    """nplda_dplda_quadform_f32 = dplda_quadform (three torch.cat) + pack_matrix(mode 2), bit for bit; the product with it
    is the gradient of x^T M x + x^T v w.r.t. x."""
    from neuralplda_amd import ops
    rng = np.random.default_rng(D1)
    w = torch.from_numpy(rng.standard_normal((1, 2 * D1 * D1 + D1)).astype(np.float32)).cuda()
    (frag, K, N), v = ops.dplda_quadform_image(w, D1)
    M, v0, _ = ops.dplda_quadform(w, None, D1)
    frag0, K0, N0 = ops.pack_matrix(M, mode=2)
    assert (K, N) == (K0, N0) == (2 * D1, 2 * D1)
    assert torch.equal(frag, frag0) and torch.equal(v, v0)
    x = torch.from_numpy(rng.standard_normal((37, 2 * D1)).astype(np.float32)).cuda()
    out = ops.rows_matmul(x, (frag, K, N), bias=v)
    ref = x.double() @ (M + M.T).double() + v0.double()
    np.testing.assert_allclose(out.cpu().numpy(), ref.cpu().numpy(), atol=2e-4, rtol=1e-5)


@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("lossname,D0,D1,B", [("crossentropy", 512, 170, 256), ("SoftCdet", 512, 170, 2048), ("SoftCdet", 128, 40, 100),
                                               ("crossentropy", 128, 40, 4100), ("SoftCdet", 128, 40, 4096)])
def test_dplda_recipe_step_in_four_launches_equals_the_separate_calls(hip_lib, monkeypatch, graph, lossname, D0, D1, B):
    """The recipe step of xvector_DPlda_pytorch.py:35-43 as four launches (score | loss | moments | nplda_dplda_update_f32:
    gradient fold + Adam + parameter and quadratic-form-image stores) against the seven separate launches it replaces
    (image x 2, score, loss, moments, fold, Adam): the same device functions in the same order -> the same parameter bits
    after every step, thresholds included; the image the step carries equals a fresh pack of the updated parameters, also
    after somebody else rewrites a parameter between two steps."""
    from neuralplda_amd import ops, train
    rg = np.random.default_rng(21)
    W1 = (rg.standard_normal((D1, D0)) / np.sqrt(D0)).astype(np.float32)
    b1 = (0.1 * rg.standard_normal(D1)).astype(np.float32)
    wlr = (0.05 * rg.standard_normal((1, 2 * D1 * D1 + D1))).astype(np.float32)
    batches = [(torch.from_numpy(rg.standard_normal((B, D0)).astype(np.float32)).cuda(),
                torch.from_numpy(rg.standard_normal((B, D0)).astype(np.float32)).cuda(),
                torch.from_numpy((rg.random(B) < 0.2).astype(np.float32)).cuda()) for _ in range(4)]

    def run(separate):
        if separate:
            monkeypatch.setenv("NPLDA_DPLDA_SEPARATE", "1")
        else:
            monkeypatch.delenv("NPLDA_DPLDA_SEPARATE", raising=False)
        m = make(D0, D1, W1, b1, wlr, [0.1])
        m.lossfn = lossname
        _freeze_lda(m)
        step = train.FusedDPldaStep(m, 1e-3, weight_decay=1e-5, batch_size=B, graph=graph)
        out = []
        for i, (x1, x2, t) in enumerate(batches):
            loss = float(step(x1, x2, t))
            out.append((loss, {k: v.detach().clone() for k, v in m.state_dict().items()}))
            if i == 2:  # the progress line's mean of the losses since the previous line: the update launch keeps the sum (ABI 4)
                mean3 = step.pop_loss_mean()
                assert abs(mean3 - float(np.mean([o[0] for o in out]))) <= 1e-6 * abs(mean3), (separate, mean3)
            if not separate:
                fresh = ops.dplda_pack(m.centering_and_LDA.weight.detach(), m.centering_and_LDA.bias.detach(),
                                       m.logistic_regres.weight.detach(), m.logistic_regres.bias.detach())
                nb = {40: 4, 170: 11}[D1]
                used = fresh[0].numel() - 2 * nb * 256  # (the image ends in a chunk of never-read slack: torch.empty)
                assert torch.equal(step._img[0][:used], fresh[0][:used]), i
            if i == 1:  # an outside write between two steps: the step notices (version counters) and re-packs in place
                with torch.no_grad():
                    m.logistic_regres.weight.mul_(0.5)
        return out, step

    new, step_new = run(False)
    old, _ = run(True)
    # (round 6: the loss rides in the moments launch up to 4096 pairs: three launches; a launch of its own above that)
    assert step_new.launches_per_step.startswith("3 " if B <= 4096 else "4 ")
    for i, ((la, sa), (lb, sb)) in enumerate(zip(new, old)):
        assert la == lb, (i, la, lb)
        for k in sa:
            assert torch.equal(sa[k], sb[k]), (i, k, (sa[k] - sb[k]).abs().max().item())
    assert step_new.step_count[0].item() == 4
    assert abs(step_new.pop_loss_mean() - new[3][0]) <= 1e-6 * abs(new[3][0])  # (one step since the pop after the third)
