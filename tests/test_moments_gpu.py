"""GPU parity of nplda_weighted_moments_f32 (per-class count / sum x / sum x x^T of paired rows) and of the closed-form
Gaussian-backend statistics built on it (xvector_GaussianBackend_pytorch.py:30-56) against the fp64 oracle.
Tolerance: fp32 products summed in fp32 within a row group, fp64 across groups -> |d| <= 1e-5 * scale of the sum."""
import os

import numpy as np
import pytest
import torch

from oracle import nplda_oracle as orc

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("B,n", [(1, 4), (7, 32), (50, 48), (2048, 340), (2049, 300), (16, 384), (100003, 64)])
@pytest.mark.parametrize("nc", [1, 2])
def test_moments_vs_oracle(hip_lib, B, n, nc):
    from neuralplda_amd import ops
    rg = np.random.default_rng(B * 7 + n + nc)
    x = rg.standard_normal((B, n)).astype(np.float32)
    w0 = rg.standard_normal(B).astype(np.float32)
    w1 = (rg.random(B) < 0.3).astype(np.float32)
    xd = torch.from_numpy(x).cuda()
    out = ops.weighted_moments(xd, torch.from_numpy(w0).cuda(), torch.from_numpy(w1).cuda() if nc == 2 else None)
    cnt, sm, sq = [o.cpu().numpy() for o in out]
    for c, w in enumerate((w0, w1)[:nc]):
        rc, rs, rq = orc.weighted_moments(x, w)
        scale = np.sqrt(max(B, 1)) * 4
        assert abs(cnt[c] - rc) < 1e-5 * max(1.0, np.abs(w).sum())
        np.testing.assert_allclose(sm[c], rs, atol=1e-5 * scale)
        np.testing.assert_allclose(sq[c], rq, atol=1e-5 * scale)
        assert np.array_equal(sq[c], sq[c].T)  # lower tiles are mirrored, diagonal tiles are symmetric bit for bit


def test_moments_accumulate_empty_and_strided(hip_lib):
    from neuralplda_amd import ops
    rg = np.random.default_rng(0)
    x = torch.from_numpy(rg.standard_normal((300, 40)).astype(np.float32)).cuda()
    w = torch.from_numpy(rg.random(300).astype(np.float32)).cuda()
    whole = ops.weighted_moments(x, w)
    part = ops.weighted_moments(x[:100], w[:100])
    part = ops.weighted_moments(x[100:], w[100:], out=part)
    part = ops.weighted_moments(x[:0], w[:0], out=part)  # empty batch leaves the sums alone
    for a, b in zip(whole, part):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), atol=1e-4)
    z = ops.weighted_moments(x[:0], w[:0])
    assert all(float(t.abs().sum()) == 0.0 for t in z)
    wide = torch.zeros((300, 64), dtype=torch.float32, device="cuda")
    wide[:, :40] = x
    sub = ops.weighted_moments(wide[:, :40], w)  # row stride 64, 40 valid columns
    for a, b in zip(whole, sub):
        assert torch.equal(a, b)
    again = ops.weighted_moments(x, w)
    for a, b in zip(whole, again):
        assert torch.equal(a, b)  # deterministic


def test_moments_rejects(hip_lib):
    from neuralplda_amd import _lib, ops
    x = torch.zeros((8, 6), dtype=torch.float32, device="cuda")
    with pytest.raises(ValueError):
        ops.weighted_moments(x, torch.zeros(8, device="cuda"))
    with pytest.raises(_lib.NpldaHipError):
        ops.weighted_moments(torch.zeros((8, 388), device="cuda"), torch.zeros(8, device="cuda"))
    with pytest.raises(ValueError):
        ops.weighted_moments(torch.zeros((8, 8), device="cuda"), torch.zeros(7, device="cuda"))


class NC:
    def __init__(self, D0, D1):
        self.xvector_dim, self.layer1_LDA_dim, self.layer2_PLDA_spkfactor_dim = D0, D1, D1
        self.beta, self.alpha, self.device, self.loss = [99.0, 199.0], 15.0, "cuda", "SoftCdet"


def test_gaussian_backend_closed_form_training(hip_lib):
    """train_gaussian_backend == the oracle's restatement of xvector_GaussianBackend_pytorch.py:30-56 on a
    speaker-structured synthetic set (the reference script itself cannot be imported)."""
    from neuralplda_amd import models, train
    from neuralplda_amd.sv_trials_loaders import combine_trials_and_get_loader
    rg = np.random.default_rng(11)
    D0, D1, nspk, per = 64, 16, 40, 6
    spk = rg.standard_normal((nspk, D0)).astype(np.float32)
    xv = (np.repeat(spk, per, 0) + 0.7 * rg.standard_normal((nspk * per, D0))).astype(np.float32)
    ids = [f"s{i // per:02d}-u{i:03d}" for i in range(nspk * per)]
    mega = {u: xv[i] for i, u in enumerate(ids)}
    num_to_id = dict(enumerate(ids))
    id_to_num = {u: i for i, u in num_to_id.items()}
    a, b = rg.integers(0, len(ids), 6000), rg.integers(0, len(ids), 6000)
    lab = (a // per == b // per).astype(int)
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        f = os.path.join(td, "tr.tsv")
        open(f, "w").write("\n".join(f"{ids[i]}\t{ids[j]}\t{l}" for i, j, l in zip(a, b, lab)) + "\n")
        np.random.seed(0)
        torch.manual_seed(0)
        loader = combine_trials_and_get_loader([f], id_to_num, subsample_factors=[1.0], batch_size=512)
    gb = models.GaussianBackend(NC(D0, D1))
    W1 = (rg.standard_normal((D1, D0)) / np.sqrt(D0)).astype(np.float32)
    b1 = (0.1 * rg.standard_normal(D1)).astype(np.float32)
    with torch.no_grad():
        gb.centering_and_LDA.weight.copy_(torch.from_numpy(W1))
        gb.centering_and_LDA.bias.copy_(torch.from_numpy(b1))
    gb = gb.cuda()
    train.train_gaussian_backend(None, gb, loader, mega, num_to_id)
    # oracle: same pairs (order is irrelevant for the sums)
    pairs = [(int(p), int(q), float(t)) for d1, d2, tt in loader for p, q, t in zip(d1, d2, tt)]
    p1 = np.asarray([p for p, _, _ in pairs]); p2 = np.asarray([q for _, q, _ in pairs]); t = np.asarray([v for _, _, v in pairs])
    y1, _ = orc.normalize(xv[p1].astype(np.float64) @ W1.T.astype(np.float64) + b1, np.float64)
    y2, _ = orc.normalize(xv[p2].astype(np.float64) @ W1.T.astype(np.float64) + b1, np.float64)
    paired = np.concatenate([y1, y2], 1)
    mu_t, Lam_t, mu_n, Lam_n = orc.gb_fit(paired, t)
    np.testing.assert_allclose(gb.paired_mean_target.cpu().numpy(), mu_t, atol=1e-6)
    np.testing.assert_allclose(gb.paired_mean_nontarget.cpu().numpy(), mu_n, atol=1e-6)
    for got, ref in ((gb.paired_cov_inv_target, Lam_t), (gb.paired_cov_inv_nontarget, Lam_n)):
        np.testing.assert_allclose(got.cpu().numpy(), ref, atol=2e-4 * np.abs(ref).max(), rtol=2e-3)
    # the fitted backend separates the classes, and its losses / metrics run (they cannot in the reference)
    x1, x2 = torch.from_numpy(xv[p1]).cuda(), torch.from_numpy(xv[p2]).cuda()
    s = gb(x1, x2)
    tt = torch.from_numpy(t.astype(np.float32)).cuda()
    assert float(s[tt > 0.5].mean()) > float(s[tt < 0.5].mean())
    mc, th = gb.minc(s, tt, update_thresholds=True)
    assert 0.0 <= float(mc) < 1.0 and float(gb.cdet(s, tt)) >= float(mc) - 1e-6
    assert np.isfinite(float(gb.softcdet(s, tt)))
    assert list(gb.state_dict().keys()) == ["centering_and_LDA.weight", "centering_and_LDA.bias"]
