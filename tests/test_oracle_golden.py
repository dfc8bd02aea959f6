"""Pin the CPU oracle (oracle/nplda_oracle.py) against golden vectors produced by the REFERENCE
itself (tests/golden/make_golden.py, run in the build container).  CPU only.
"""
import os

import numpy as np
import pytest

from oracle import nplda_oracle as orc

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def params_from(d):
    return orc.Params(d["W1"], d["b1"], d["W2"], d["b2"], d["P_sqrt"], d["Q"])


@pytest.fixture(scope="module")
def g1():
    return load("g1_kaldi_params.npz")


def test_g1_kaldi_init_formulas(g1):
    """Psi -> (P, Q) and bias folding reproduce the reference's LoadPldaParamsFromKaldi."""
    psi = g1["psi"]
    np.testing.assert_allclose(psi[:5], [27.6976, 16.8390, 12.4640, 9.8139, 8.7320], atol=1e-4)
    diagP, diagQ = orc.kaldi_psi_to_pq(psi)
    np.testing.assert_allclose(np.sqrt(diagP).astype(np.float32), g1["P_sqrt"], rtol=1e-6)
    np.testing.assert_allclose(diagQ.astype(np.float32), g1["Q"], rtol=1e-6)
    assert abs(diagP[0] - 0.491134) < 1e-6 and abs(diagQ[0] + 0.474020) < 1e-6
    assert list(g1["state_dict_keys"]) == ["P_sqrt", "Q", "Th99", "Th199", "threshold_Xent",
                                           "centering_and_LDA.weight", "centering_and_LDA.bias",
                                           "centering_and_wccn_plda.weight", "centering_and_wccn_plda.bias"]


def test_g2_forward_kaldi170(g1):
    g = load("g2_forward_kaldi170.npz")
    p = params_from(g1)
    z1, (u1, y1, _) = orc.extract_plda_embeddings(g["x1"], p, np.float32, True)
    np.testing.assert_allclose(u1, g["u1"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(y1, g["y1"], atol=1e-6, rtol=1e-5)
    np.testing.assert_allclose(z1, g["z1"], atol=2e-6, rtol=1e-5)
    s32 = orc.forward(g["x1"], g["x2"], p, np.float32)
    s64 = orc.forward(g["x1"], g["x2"], p, np.float64)
    np.testing.assert_allclose(s32, g["s"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(s64, g["s64"], atol=1e-9, rtol=1e-9)
    np.testing.assert_allclose(orc.forward_from_plda_embeddings(g["z1"], g["z2"], p), g["s_from_z"], atol=2e-6)
    np.testing.assert_allclose(orc.forward(g["x1"][:1], g["x2"][:1], p), g["s_b1"], atol=2e-5)
    # indexed formulation == pair formulation
    z = np.concatenate([g["z1"], g["z2"]])
    n = g["z1"].shape[0]
    np.testing.assert_allclose(orc.score_indexed(z, np.arange(n), n + np.arange(n), p), g["s_from_z"], atol=2e-6)


def test_g2_forward_rand150():
    g = load("g2_forward_rand150.npz")
    p = params_from(g)
    np.testing.assert_allclose(orc.forward(g["x1"], g["x2"], p), g["s"], atol=2e-5, rtol=1e-5)
    p0 = params_from(g)
    p0.b1 = np.zeros_like(p.b1)
    s0 = orc.forward(g["x1"], g["x2"], p0)
    assert np.all(np.isfinite(s0))
    np.testing.assert_allclose(s0, g["s_zero_b1"], atol=2e-5, rtol=1e-5)  # row 7 hits the eps branch


def test_g3_loss_and_gradients_small():
    g = load("g3_loss_grad_small.npz")
    p = params_from(g)
    # thresholds are fp32 parameters in the reference: -0.3 is float32(-0.3) there
    theta, beta, alpha = g["theta"].astype(np.float32).astype(np.float64), g["beta"], float(g["alpha"])
    s = orc.forward(g["x1"], g["x2"], p, np.float64)
    np.testing.assert_allclose(s, g["SoftCdet_s"], atol=2e-5)
    # losses
    np.testing.assert_allclose(orc.softcdet(g["SoftCdet_s"], g["t"], theta, beta, alpha), g["SoftCdet_L"], rtol=2e-6)
    np.testing.assert_allclose(orc.crossentropy(g["crossentropy_s"], g["t"], float(g["theta_xent"])),
                               g["crossentropy_L"], rtol=2e-6)
    np.testing.assert_allclose(orc.softcdet(s, g["t"], theta, beta, alpha, np.float64), g["SoftCdet64_L"], rtol=1e-10)
    # dL/ds and dL/dtheta
    gs, dth = orc.softcdet_grad(s, g["t"], theta, beta, alpha)
    np.testing.assert_allclose(gs, g["SoftCdet64_g"], atol=1e-12, rtol=1e-9)
    assert np.abs(gs - g["SoftCdet_g"]).max() <= 1e-3 * np.abs(g["SoftCdet_g"]).max()  # fp32 autograd: 1-sigmoid quantised at 6e-8
    np.testing.assert_allclose(dth, [g["SoftCdet64_grad_Th99"][0], g["SoftCdet64_grad_Th199"][0]], rtol=1e-9)
    gx, dthx = orc.crossentropy_grad(s, g["t"], float(g["theta_xent"]))
    assert np.abs(gx - g["crossentropy_g"]).max() <= 1e-4 * np.abs(g["crossentropy_g"]).max()
    np.testing.assert_allclose(dthx, g["crossentropy_grad_threshold_Xent"], rtol=1e-4)
    # parameter gradients: fp64 oracle vs the reference's fp64 autograd (tight), and vs fp32 autograd (loose)
    grads = orc.backward(g["x1"], g["x2"], gs, p)
    names = {"W1": "centering_and_LDA.weight", "b1": "centering_and_LDA.bias", "W2": "centering_and_wccn_plda.weight",
             "b2": "centering_and_wccn_plda.bias", "P_sqrt": "P_sqrt", "Q": "Q"}
    for k, rn in names.items():
        ref64 = g["SoftCdet64_grad_" + rn]
        np.testing.assert_allclose(grads[k], ref64, atol=1e-10 * max(1.0, np.abs(ref64).max()), rtol=1e-8, err_msg=k)
        ref32 = g["SoftCdet_grad_" + rn]
        assert np.abs(grads[k] - ref32).max() <= 2e-2 * np.abs(ref32).max(), k  # bounded by the noise of the reference fp32 autograd itself
    gradsx = orc.backward(g["x1"], g["x2"], gx, p)
    for k, rn in names.items():
        ref32 = g["crossentropy_grad_" + rn]
        assert np.abs(gradsx[k] - ref32).max() <= 1e-4 * np.abs(ref32).max(), k


def test_g3_loss_kaldi170(g1):
    g = load("g3_loss_kaldi170.npz")
    f = load("g2_forward_kaldi170.npz")
    p = params_from(g1)
    s = orc.forward(f["x1"], f["x2"], p, np.float64)
    theta, beta, alpha = g["theta"], g["beta"], float(g["alpha"])
    np.testing.assert_allclose(orc.softcdet(s, g["t"], theta, beta, alpha, np.float64), g["L"], rtol=1e-4)
    gs, dth = orc.softcdet_grad(s, g["t"], theta, beta, alpha)
    np.testing.assert_allclose(gs, g["g"], atol=2e-4 * np.abs(g["g"]).max())
    grads = orc.backward(f["x1"], f["x2"], gs, p)
    for k, ref in (("Q", g["grad_Q"]), ("P_sqrt", g["grad_P_sqrt"]), ("b2", g["grad_b2"]), ("b1", g["grad_b1"])):
        assert np.abs(grads[k] - ref).max() <= 5e-4 * np.abs(ref).max(), k
    assert np.abs(grads["W2"][0] - g["grad_W2_row0"]).max() <= 5e-4 * np.abs(g["grad_W2_row0"]).max()
    assert np.abs(grads["W1"][0] - g["grad_W1_row0"]).max() <= 5e-4 * np.abs(grads["W1"]).max()
    np.testing.assert_allclose(np.linalg.norm(grads["W1"]), g["grad_W1_fro"], rtol=5e-4)
    np.testing.assert_allclose(np.linalg.norm(grads["W2"]), g["grad_W2_fro"], rtol=5e-4)
    np.testing.assert_allclose(dth, [g["grad_Th99"][0], g["grad_Th199"][0]], rtol=5e-4)


def test_g5_metrics():
    g = load("g5_metrics.npz")
    s, t, theta, beta, alpha = g["s"], g["t"], g["theta"], g["beta"], float(g["alpha"])
    np.testing.assert_allclose(orc.cdet(s, t, theta, beta), g["cdet"], rtol=1e-6)
    np.testing.assert_allclose(orc.softcdet(s, t, theta, beta, alpha), g["softcdet"], rtol=2e-6)
    np.testing.assert_allclose(orc.crossentropy(s, t, 0.0), g["xent"], rtol=2e-6)
    mc, th = orc.minc_reference(s, t, list(beta))
    np.testing.assert_allclose(mc, g["minc"], rtol=1e-6)
    np.testing.assert_array_equal(np.asarray([th[99.0], th[199.0]], np.float32), g["minc_th"].astype(np.float32))
    np.testing.assert_array_equal(g["th_after_update"].astype(np.float32), g["minc_th"].astype(np.float32))
    # separable set: reference quirk floor (count-1 / "1.0 when empty"), exact sweep gives 0
    mcs, ths = orc.minc_reference(g["s_sep"], t, list(beta))
    np.testing.assert_allclose(mcs, g["minc_sep"], rtol=1e-6)
    assert orc.minc_exact(g["s_sep"], t, list(beta))[0] == 0.0
    assert mcs > 0


def test_g6_asnorm_matches_reference_script():
    g = load("g6_asnorm.npz")
    ids = list(g["ids"])
    stats = orc.cohort_stats(g["cohort"], int(g["topn"]), "lowest")
    row = {k: i for i, k in enumerate(ids)}
    ie = [row[e] for e in g["enroll"]]
    it = [row[t.replace(".sph", "")] for t in g["test"]]
    out = orc.asnorm_apply(g["raw"], ie, it, stats)
    for c, k in enumerate(("znorm", "tnorm", "snorm", "asnorm1")):
        np.testing.assert_allclose(out[:, c], g[k], rtol=1e-12, atol=1e-12, err_msg=k)
    # the reference's "top-N" is the N SMALLEST (ascending sort then [:N]); highest-N differs
    hi = orc.cohort_stats(g["cohort"], int(g["topn"]), "highest")
    assert np.abs(hi[:, 2] - stats[:, 2]).min() > 0.1


def test_g7_gaussian_backend():
    g = load("g7_gb.npz")
    s = orc.gb_forward(g["x1"], g["x2"], g["W1"], g["b1"], g["mu_t"], g["Lam_t"], g["mu_n"], g["Lam_n"])
    np.testing.assert_allclose(s, g["s"], atol=2e-4, rtol=2e-5)
    s64 = orc.gb_forward(g["x1"], g["x2"], g["W1"], g["b1"], g["mu_t"], g["Lam_t"], g["mu_n"], g["Lam_n"], np.float64)
    np.testing.assert_allclose(s64, g["s"], atol=2e-4, rtol=2e-5)


def test_g9_e2e_scores_and_metrics(g1):
    from neuralplda_amd import kaldi_format  # noqa: F401  (host-only module, no GPU needed)
    from tests import synth
    g = load("g9_e2e_kaldi170.npz")
    p = params_from(g1)
    # regenerate x from the seed; Dt/plda_mean are not in g1, so invert W2 = Dt, b2 = -Dt m
    Dt = g1["W2"].astype(np.float64)
    pm = g1["plda_mean"]
    x, spk = synth.speaker_structured_xvectors(g1["W1"], g1["b1"], Dt, pm, g1["psi"], int(g["S"]), int(g["U"]),
                                               float(g["c"]), int(g["seed"]))
    # the north-star gate must not vanish: a drifted RNG stream is a hard failure (regenerate g9 with make_golden.py)
    assert np.allclose(x[:4], g["x_head"], atol=1e-4) and np.allclose(x.sum(axis=0, dtype=np.float64), g["x_colsum"], atol=1e-2), \
        "numpy RNG stream differs from the fixture generator; G9 inputs cannot be regenerated"
    s = orc.forward(x[g["i1"]], x[g["i2"]], p, np.float32)
    np.testing.assert_allclose(s, g["s"], atol=5e-5, rtol=1e-5)
    mc, th = orc.minc_reference(g["s"], g["t"], [99.0, 199.0])
    np.testing.assert_allclose(mc, g["minc_ref"], rtol=1e-6)
    np.testing.assert_allclose([th[99.0], th[199.0]], g["minc_ref_th"], rtol=1e-6)
    e = orc.eer(g["s"], g["t"])
    assert 0.002 < e < 0.08


def test_g10_dplda_forward(g1):
    """DPlda.forward / forward_from_plda_embeddings (utils/models.py:479-495) against reference outputs."""
    g = np.load(os.path.join(G, "g10_dplda_small.npz"), allow_pickle=True)
    s32 = orc.dplda_forward(g["x1"], g["x2"], g["W1"], g["b1"], g["wlr"], g["blr"], np.float32)
    s64 = orc.dplda_forward(g["x1"], g["x2"], g["W1"], g["b1"], g["wlr"], g["blr"], np.float64)
    np.testing.assert_allclose(s32, g["s"], atol=2e-6, rtol=2e-5)
    np.testing.assert_allclose(s64, g["s"], atol=2e-6, rtol=2e-5)
    np.testing.assert_allclose(g["s_from_emb"], g["s"], atol=1e-6)
    assert list(g["state_dict_keys"]) == ["Th99", "centering_and_LDA.weight", "centering_and_LDA.bias",
                                          "logistic_regres.weight", "logistic_regres.bias"] or \
        set(g["state_dict_keys"]) >= {"centering_and_LDA.weight", "logistic_regres.weight"}
    # Kaldi-initialised LDA at the production width with a seeded linear unit
    f = np.load(os.path.join(G, "g2_forward_kaldi170.npz"))
    k = np.load(os.path.join(G, "g10_dplda_kaldi170.npz"))
    rg = np.random.default_rng(int(k["seed"]))
    wlr = (rg.standard_normal((1, 2 * 170 * 170 + 170)) * 0.05).astype(np.float32)
    s = orc.dplda_forward(f["x1"], f["x2"], g1["W1"], g1["b1"], wlr, np.asarray([0.125]), np.float64)
    np.testing.assert_allclose(s, k["s"], atol=2e-5, rtol=2e-5)


def test_g10_dplda_gradients():
    """Oracle chain (forward -> loss grad -> dplda_backward) against the reference's autograd (G10 grad)."""
    g = np.load(os.path.join(G, "g10_dplda_small.npz"), allow_pickle=True)
    gg = np.load(os.path.join(G, "g10_dplda_grad.npz"))
    f64 = np.float64
    y1, _ = orc.normalize(g["x1"].astype(f64) @ g["W1"].astype(f64).T + g["b1"], f64)
    y2, _ = orc.normalize(g["x2"].astype(f64) @ g["W1"].astype(f64).T + g["b1"], f64)
    s = orc.dplda_from_embeddings(y1, y2, g["wlr"], g["blr"], f64)
    theta = [float(v) for v in gg["theta"]]  # the fp64 re-evaluation filled its thresholds in fp64
    L = orc.softcdet(s, gg["t"], theta, [99.0, 199.0], 15.0, f64)
    gs, dth = orc.softcdet_grad(s, gg["t"], theta, [99.0, 199.0], 15.0, f64)
    dw, db = orc.dplda_backward(y1, y2, gs)
    np.testing.assert_allclose(L, gg["SoftCdet_f64_L"], rtol=1e-9)
    np.testing.assert_allclose(dw, gg["SoftCdet_f64_dwlr"], atol=1e-9, rtol=1e-8)
    np.testing.assert_allclose(db, gg["SoftCdet_f64_dblr"], rtol=1e-9)
    np.testing.assert_allclose(dth, [gg["SoftCdet_f64_dTh99"][0], gg["SoftCdet_f64_dTh199"][0]], rtol=1e-9)
    np.testing.assert_allclose(dw, gg["SoftCdet_f32_dwlr"], atol=2e-5, rtol=1e-4)
    gx, _ = orc.crossentropy_grad(s, gg["t"], 0.0, f64)
    dwx, dbx = orc.dplda_backward(y1, y2, gx)
    np.testing.assert_allclose(dwx, gg["crossentropy_f64_dwlr"], atol=1e-11, rtol=1e-8)
    np.testing.assert_allclose(dbx, gg["crossentropy_f64_dblr"], rtol=1e-9)


# ---- G11: input gradients (reference autograd through utils/models.py:366-382 / :484-495) ----------------------

def _rel(a, b):
    return np.abs(np.asarray(a, np.float64) - b).max() / max(np.abs(b).max(), 1e-30)


def test_g11_input_grads_small():
    g = load("g11_input_grads_small.npz")
    p = params_from(g)
    # thresholds are float32 parameters in the reference (also in its .double() copy): -0.3 is really float32(-0.3)
    th, beta, alpha = [float(np.float32(v)) for v in g["theta"]], [float(b) for b in g["beta"]], float(g["alpha"])
    s = orc.forward(g["x1"], g["x2"], p, np.float64)
    for lossname, (gs, _) in (("SoftCdet", orc.softcdet_grad(s, g["t"], th, beta, alpha)),
                              ("crossentropy", orc.crossentropy_grad(s, g["t"], float(g["theta_xent"])))):
        dx1, dx2 = orc.input_grads(g["x1"], g["x2"], gs, p)
        assert _rel(dx1, g[f"{lossname}_f64_dx1"]) <= 1e-9 and _rel(dx2, g[f"{lossname}_f64_dx2"]) <= 1e-9
        # the reference's own fp32 autograd is much noisier than that (sigma' from a saturated sigmoid, fp32 GEMMs)
        assert _rel(dx1, g[f"{lossname}_f32_dx1"]) <= 2e-2
        assert _rel(orc.backward(g["x1"], g["x2"], gs, p)["W1"], g[f"{lossname}_f64_dW1"]) <= 1e-9
    e = orc.embed_backward(g["x1"], g["Gz"], p)
    for k, name in (("x", "dx"), ("W1", "dW1"), ("b1", "db1"), ("W2", "dW2"), ("b2", "db2")):
        assert _rel(e[k], g[f"embed_f64_{name}"]) <= 1e-9, k
        assert _rel(e[k], g[f"embed_f32_{name}"]) <= 1e-4, k
    np.testing.assert_allclose(orc.extract_plda_embeddings(g["x1"], p, np.float64), g["embed_f64_z"], atol=1e-12)
    d = orc.embscore_backward(g["z1"], g["z2"], g["gs"], p)
    for k in ("z1", "z2", "P_sqrt", "Q"):
        assert _rel(d[k], g[f"embscore_f64_d{k}"]) <= 1e-9, k
        assert _rel(d[k], g[f"embscore_f32_d{k}"]) <= 1e-5, k


def test_g11_input_grads_kaldi170(g1):
    g = load("g11_input_grads_kaldi170.npz")
    g2, g3 = load("g2_forward_kaldi170.npz"), load("g3_loss_kaldi170.npz")
    p = params_from(g1)
    s = orc.forward(g2["x1"], g2["x2"], p, np.float64)
    gs, _ = orc.softcdet_grad(s, g3["t"], [float(np.float32(v)) for v in g3["theta"]], [99.0, 199.0], 15.0)
    dx1, dx2 = orc.input_grads(g2["x1"], g2["x2"], gs, p)
    assert _rel(dx1, g["dx1_64"]) <= 1e-8 and _rel(dx2, g["dx2_64"]) <= 1e-8
    assert _rel(dx1, g["dx1"]) <= 2e-2 and _rel(dx2, g["dx2"]) <= 2e-2


def test_g11_dplda_input_grads():
    g = load("g11_dplda_input_grads.npz")
    th, beta, alpha = [float(np.float32(v)) for v in g["theta"]], [float(b) for b in g["beta"]], float(g["alpha"])
    s = orc.dplda_forward(g["x1"], g["x2"], g["W1"], g["b1"], g["wlr"], g["blr"], np.float64)
    for lossname, (gs, _) in (("SoftCdet", orc.softcdet_grad(s, g["t"], th, beta, alpha)),
                              ("crossentropy", orc.crossentropy_grad(s, g["t"], 0.0))):
        d = orc.dplda_lda_backward(g["x1"], g["x2"], gs, g["W1"], g["b1"], g["wlr"])
        for k, name in (("x1", "dx1"), ("x2", "dx2"), ("W1", "dW1"), ("b1", "db1")):
            assert _rel(d[k], g[f"{lossname}_f64_{name}"]) <= 1e-9, (lossname, k)
            assert _rel(d[k], g[f"{lossname}_f32_{name}"]) <= 2e-2, (lossname, k)


def test_g7_gb_kaldi170_matrices_are_stored():
    """The 340-d GaussianBackend golden carries its statistics (round 1 stored only a seed) and the oracle meets the
    reference's fp64 evaluation of them unconditionally."""
    g = load("g7_gb_kaldi170.npz")
    g2 = load("g2_forward_kaldi170.npz")
    g1_ = load("g1_kaldi_params.npz")
    s64 = orc.gb_forward(g2["x1"], g2["x2"], g1_["W1"], g1_["b1"], g["mt"], g["Lt"], g["mn"], g["Ln"], np.float64)
    np.testing.assert_allclose(s64, g["s64"], atol=1e-8, rtol=1e-9)
    np.testing.assert_allclose(s64, g["s"], atol=5e-4, rtol=5e-5)  # the reference's fp32 difference of two O(100) forms


def test_torch_cpu_port_matches_the_reference_outputs(g1):
    """oracle/nplda_oracle_torch.py (bench.py's cpu_baseline: the reference's forward as torch CPU ops) against G2."""
    import torch
    from oracle import nplda_oracle_torch as ot
    g = load("g2_forward_kaldi170.npz")
    p = ot.TorchParams(g1["W1"], g1["b1"], g1["W2"], g1["b2"], g1["P_sqrt"], g1["Q"])
    with torch.no_grad():
        s = ot.forward(torch.from_numpy(g["x1"]), torch.from_numpy(g["x2"]), p).numpy()
        z1 = ot.extract_plda_embeddings(torch.from_numpy(g["x1"]), p).numpy()
    np.testing.assert_allclose(s, g["s"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(z1, g["z1"], atol=2e-6, rtol=1e-5)
    mega = {f"u{i}": g["x1"][i] for i in range(8)}
    n2i = {i: f"u{i}" for i in range(8)}
    a, b = ot.gather_numbatch(mega, n2i, [3, 1], [0, 7])
    assert np.array_equal(a.numpy(), g["x1"][[3, 1]]) and np.array_equal(b.numpy(), g["x1"][[0, 7]])
