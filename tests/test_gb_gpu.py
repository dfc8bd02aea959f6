"""GPU parity of the fused GaussianBackend.forward kernel vs the reference's outputs (G7) and the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import nplda_oracle as orc

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class NC:
    def __init__(self, D0, D1):
        self.xvector_dim, self.layer1_LDA_dim, self.layer2_PLDA_spkfactor_dim = D0, D1, D1
        self.beta, self.alpha, self.device, self.loss = [99.0], 15.0, "cuda", "SoftCdet"


def make_gb(D0, D1, W1, b1, mu_t, Lam_t, mu_n, Lam_n):
    from neuralplda_amd import models
    gb = models.GaussianBackend(NC(D0, D1))
    with torch.no_grad():
        gb.centering_and_LDA.weight.copy_(torch.from_numpy(W1))
        gb.centering_and_LDA.bias.copy_(torch.from_numpy(b1))
    gb.paired_mean_target, gb.paired_cov_inv_target = torch.from_numpy(mu_t), torch.from_numpy(Lam_t)
    gb.paired_mean_nontarget, gb.paired_cov_inv_nontarget = torch.from_numpy(mu_n), torch.from_numpy(Lam_n)
    return gb.cuda()


def test_gb_golden_small(hip_lib):
    g = np.load(os.path.join(G, "g7_gb.npz"))
    gb = make_gb(32, 16, g["W1"], g["b1"], g["mu_t"], g["Lam_t"], g["mu_n"], g["Lam_n"])
    s = gb(torch.from_numpy(g["x1"]).cuda(), torch.from_numpy(g["x2"]).cuda()).cpu().numpy()
    mag = np.abs(g["s"]).max()
    np.testing.assert_allclose(s, g["s"], atol=2e-5 * max(mag, 1.0), rtol=2e-5)
    xp = gb.forward_getpaired(torch.from_numpy(g["x1"]).cuda(), torch.from_numpy(g["x2"]).cuda()).cpu().numpy()
    np.testing.assert_allclose(xp, g["paired"], atol=1e-6, rtol=1e-5)


def test_gb_kaldi170_golden(hip_lib):
    """340-d GaussianBackend on the G2 inputs with the statistics stored in the fixture, against the reference's own
    fp32 and fp64 evaluations (unconditional)."""
    g1 = np.load(os.path.join(G, "g1_kaldi_params.npz"))
    f = np.load(os.path.join(G, "g2_forward_kaldi170.npz"))
    g = np.load(os.path.join(G, "g7_gb_kaldi170.npz"))
    Lt, Ln, mt, mn = g["Lt"], g["Ln"], g["mt"], g["mn"]
    gb = make_gb(512, 170, g1["W1"], g1["b1"], mt, Lt, mn, Ln)
    s = gb(torch.from_numpy(f["x1"]).cuda(), torch.from_numpy(f["x2"]).cuda()).cpu().numpy()
    # the score is a difference of two O(100) quadratic forms: tolerance relative to their magnitude
    np.testing.assert_allclose(s, g["s64"], atol=2e-5, rtol=2e-5)
    np.testing.assert_allclose(s, g["s"], atol=5e-4, rtol=5e-5)  # the reference's fp32 evaluation is the noisier one


@pytest.mark.parametrize("D0,D1,B", [(512, 150, 1000), (64, 40, 33), (512, 170, 20000)])
def test_gb_matches_oracle(hip_lib, D0, D1, B):
    rng = np.random.default_rng(D1 + B)
    k = 1 / np.sqrt(D0)
    W1 = rng.uniform(-k, k, (D1, D0)).astype(np.float32)
    b1 = rng.uniform(-k, k, D1).astype(np.float32)
    n2 = 2 * D1
    Lt = rng.standard_normal((n2, n2)).astype(np.float32)       # deliberately NOT symmetric
    Ln = rng.standard_normal((n2, n2)).astype(np.float32)
    mt = (0.1 * rng.standard_normal(n2)).astype(np.float32)
    mn = (0.1 * rng.standard_normal(n2)).astype(np.float32)
    x1 = rng.standard_normal((B, D0)).astype(np.float32)
    x2 = rng.standard_normal((B, D0)).astype(np.float32)
    gb = make_gb(D0, D1, W1, b1, mt, Lt, mn, Ln)
    s = gb(torch.from_numpy(x1).cuda(), torch.from_numpy(x2).cuda()).cpu().numpy()
    ref = orc.gb_forward(x1, x2, W1, b1, mt, Lt, mn, Ln, np.float64)
    scale = max(1.0, np.abs(ref).max())
    assert np.abs(s - ref).max() <= 3e-5 * scale, np.abs(s - ref).max()
    assert gb(torch.empty(0, D0).cuda(), torch.empty(0, D0).cuda()).shape == (0,)


@pytest.mark.parametrize("D0,D1", [(512, 170), (512, 150), (64, 40), (32, 16)])
@pytest.mark.parametrize("B", [1, 7, 8, 9, 255, 2048, 2049])
def test_half_tile_kernel_general_and_symmetric_images(hip_lib, D0, D1, B):
    """Batches of up to 8 pairs per CU run on 8-pair HALF tiles (csrc/nplda_gb_half.h: both rows of a pair in one MFMA row
    group; 2 049 pairs are back on 16-pair tiles): a GENERAL GaussianBackend image (two passes of the quadratic form) and
    DPlda's block-symmetric one (one pass with own / partner rows) against the fp64 oracle — scores, the paired rows
    [y1 | y2] and 1 / norm — with a ragged last tile (pairs past the batch must not be written)."""
    from neuralplda_amd import ops
    from tests.test_dplda_gpu import make as make_dplda
    rng = np.random.default_rng(D1 * 7 + B)
    k = 1 / np.sqrt(D0)
    W1 = rng.uniform(-k, k, (D1, D0)).astype(np.float32)
    b1 = rng.uniform(-k, k, D1).astype(np.float32)
    n2 = 2 * D1
    Lt = rng.standard_normal((n2, n2)).astype(np.float32)
    Ln = rng.standard_normal((n2, n2)).astype(np.float32)
    mt = (0.1 * rng.standard_normal(n2)).astype(np.float32)
    mn = (0.1 * rng.standard_normal(n2)).astype(np.float32)
    x1 = rng.standard_normal((B, D0)).astype(np.float32)
    x2 = rng.standard_normal((B, D0)).astype(np.float32)
    X1, X2 = torch.from_numpy(x1).cuda(), torch.from_numpy(x2).cuda()
    with torch.no_grad():
        gb = make_gb(D0, D1, W1, b1, mt, Lt, mn, Ln)
        s = gb(X1, X2).cpu().numpy()
        ref = orc.gb_forward(x1, x2, W1, b1, mt, Lt, mn, Ln, np.float64)
        assert np.abs(s - ref).max() <= 3e-5 * max(1.0, np.abs(ref).max())
        # paired rows and 1 / norm through the same launch (guard values behind the batch stay untouched)
        pk = ops.gb_pack(*[torch.from_numpy(a).cuda() for a in (W1, b1, mt, Lt, mn, Ln)])
        sc, paired, rn = ops._gb_call(X1, X2, pk, True, True, want_rn=True)
        u1 = x1.astype(np.float64) @ W1.T.astype(np.float64) + b1
        u2 = x2.astype(np.float64) @ W1.T.astype(np.float64) + b1
        n1, n2_ = np.linalg.norm(u1, axis=1), np.linalg.norm(u2, axis=1)
        np.testing.assert_allclose(paired.cpu().numpy(), np.concatenate([u1 / n1[:, None], u2 / n2_[:, None]], axis=1),
                                   atol=2e-6, rtol=1e-5)
        np.testing.assert_allclose(rn.cpu().numpy(), np.concatenate([1 / n1, 1 / n2_]), rtol=2e-6)
        np.testing.assert_allclose(sc.cpu().numpy(), s, rtol=0, atol=0)
        # DPlda: the block-symmetric image
        wlr = (0.2 * rng.standard_normal((1, 2 * D1 * D1 + D1))).astype(np.float32)
        m = make_dplda(D0, D1, W1, b1, wlr, [-0.3])
        sd = m(X1, X2).cpu().numpy()
        refd = orc.dplda_forward(x1, x2, W1, b1, wlr, [-0.3], np.float64)
        np.testing.assert_allclose(sd, refd, atol=2e-5 * max(1.0, np.abs(refd).max()), rtol=2e-5)


def test_half_tile_kernel_equals_the_16_pair_kernel_to_rounding(hip_lib):
    """The same batches through the 16-pair tiles (NPLDA_GB_NO_HALF=1, read once per process: a child): the scores of the two
    kernels agree to rounding; the paired rows and 1 / norm bit for bit at D1 = 150, where both accumulate every layer-1
    output element in the same k order (at D1 = 170 the 16-pair kernel splits layer 1 over its waves by K: rounding)."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys, numpy as np, torch
        from neuralplda_amd import ops
        rng = np.random.default_rng(3)
        out = {}
        for D1, B in ((170, 256), (150, 2048), (170, 37)):
            W1 = rng.uniform(-.04, .04, (D1, 512)).astype(np.float32); b1 = rng.uniform(-.04, .04, D1).astype(np.float32)
            wlr = (0.2 * rng.standard_normal((1, 2 * D1 * D1 + D1))).astype(np.float32); blr = np.asarray([0.1], np.float32)
            x1 = torch.from_numpy(rng.standard_normal((B, 512)).astype(np.float32)).cuda()
            x2 = torch.from_numpy(rng.standard_normal((B, 512)).astype(np.float32)).cuda()
            pk = ops.dplda_pack(*[torch.from_numpy(a).cuda() for a in (W1, b1, wlr, blr)])
            s, paired, rn = ops._gb_call(x1, x2, pk, True, True, want_rn=True)
            out[f"s{D1}_{B}"], out[f"p{D1}_{B}"], out[f"r{D1}_{B}"] = s.cpu().numpy(), paired.cpu().numpy(), rn.cpu().numpy()
        np.savez(sys.argv[1], **out)
    """)
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        res = {}
        for tag, env in (("half", {}), ("full", {"NPLDA_GB_NO_HALF": "1"})):
            f = os.path.join(td, tag + ".npz")
            r = subprocess.run([sys.executable, "-c", code, f], cwd=root, env=dict(os.environ, **env), capture_output=True,
                               text=True, timeout=600)
            assert r.returncode == 0, r.stderr[-2000:]
            res[tag] = dict(np.load(f))
    for k in res["half"]:
        a, b = res["half"][k], res["full"][k]
        if k[0] == "s":
            np.testing.assert_allclose(a, b, atol=2e-5 * max(1.0, np.abs(b).max()), rtol=2e-5, err_msg=k)
        elif k[1:4] == "150":
            assert np.array_equal(a, b), k
        else:
            np.testing.assert_allclose(a, b, atol=2e-6, rtol=1e-5, err_msg=k)
