"""validate() (xvector_NeuralPlda_pytorch.py:56-83) on the device: the embed-once pass of a trial list that names each
utterance many times — every distinct utterance through extract_plda_embeddings ONCE (nplda_embed_rows_f32, the gather
folded in), the trials scored by index (nplda_score_indexed_f32) — against the dense forward and the fp64 oracle, and the
loader-batched pass DPlda keeps.  Tolerance: |ds| <= 2e-5 + 1e-5 |s|."""
import contextlib
import io

import numpy as np
import pytest
import torch

from oracle import nplda_oracle as orc
from tests.test_train_gpu import NC, model_from, rand_params

pytestmark = pytest.mark.gpu


def _speaker_set(rng, n_utt, n, n_spk=200):
    from neuralplda_amd import sv_trials_loaders as svl
    ids = [f"u{i:06d}" for i in range(n_utt)]
    spk = rng.integers(0, n_spk, n_utt)
    cent = rng.standard_normal((n_spk, 512)).astype(np.float32)
    mat = (cent[spk] + 0.7 * rng.standard_normal((n_utt, 512))).astype(np.float32)
    mega = svl.XvectorTable.from_matrix(ids, mat)
    a, b = rng.integers(0, n_utt, n), rng.integers(0, n_utt, n)
    lab = (spk[a] == spk[b]).astype(np.float32)
    ds = svl.TrialIndexDataset(torch.from_numpy(a), torch.from_numpy(b), torch.from_numpy(lab))
    return mega, dict(enumerate(ids)), svl._loader(ds, 5 * 2048), mat, a, b, lab


@pytest.mark.parametrize("D,n_utt,n", [(150, 20000, 300000), (170, 20000, 100000), (150, 700, 5000), (170, 64, 129)])
def test_embed_rows_and_forward_distinct_match_the_dense_forward_and_the_oracle(hip_lib, D, n_utt, n):
    from neuralplda_amd import ops
    rng = np.random.default_rng(D + n)
    mega, num_to_id, loader, mat, a, b, lab = _speaker_set(rng, n_utt, n)
    p = rand_params(rng, 512, D, D)
    nc = NC(D1=D, D2=D)
    m = model_from(p, nc, thetas=[-0.5, -0.3])
    dev = torch.device("cuda")
    X = torch.from_numpy(mat).to(dev)
    packed = ops.pack_params(*[q.detach() for q in m._params()])
    # embed_rows == gather + embed (the same kernel reads the same rows: bit for bit where the fused form runs)
    rows = torch.from_numpy(rng.integers(0, n_utt, max(n_utt // 2, 40))).to(dev)
    z1, q1 = ops.embed_rows(X, rows, packed)
    z0, q0 = ops.embed(ops.gather_rows(X, rows), packed)
    assert torch.allclose(z1, z0, rtol=0, atol=2e-6) and torch.allclose(q1, q0, rtol=1e-5, atol=1e-6)
    zo = orc.extract_plda_embeddings(mat[rows.cpu().numpy()][:500], p, np.float64)
    assert np.abs(z1[:500, :D].cpu().numpy() - zo).max() <= 2e-5
    assert float(z1[:, D:].abs().max()) == 0.0  # padding columns written as zero
    # trials through the distinct set
    n_, e1, e2, el, urows, j1, j2 = loader.device_columns_distinct(dev, None)
    assert n_ == n and torch.equal(urows[j1], e1.long()) and torch.equal(urows[j2], e2.long())
    assert torch.equal(urows, torch.unique(torch.cat([e1, e2]).long()))
    s_idx = m.forward_distinct(X, urows, j1, j2)
    with torch.no_grad():
        s_dense = torch.cat([m(X[e1[lo:lo + 10240].long()], X[e2[lo:lo + 10240].long()]) for lo in range(0, n, 10240)])
    d = (s_idx - s_dense).abs()
    assert bool((d <= 2e-5 + 1e-5 * s_dense.abs()).all()), float(d.max())
    k = min(n, 2000)
    ref = orc.forward(mat[a[:k]], mat[b[:k]], p, np.float64)
    assert np.all(np.abs(s_idx[:k].cpu().numpy() - ref) <= 2e-5 + 1e-5 * np.abs(ref))


def test_validate_embed_once_pass_agrees_with_the_dense_pass(hip_lib, monkeypatch):
    """1 M-trial shape in small: 250 k trials over 5 k utterances.  The default pass embeds once and scores by index; its
    metrics agree with the dense pass (NPLDA_VALIDATE_DENSE=1) to the G12 tolerances, and a second call reuses the
    loader's resident columns and distinct set."""
    from neuralplda_amd import train
    rng = np.random.default_rng(12)
    mega, num_to_id, loader, mat, a, b, lab = _speaker_set(rng, 5000, 250000)
    nc = NC(D1=150, D2=150, loss="SoftCdet")
    nc.batch_size = 2048
    m = model_from(rand_params(rng, 512, 150, 150), nc, thetas=[-0.5, -0.3])
    dev = torch.device("cuda")

    def run():
        out = io.StringIO()
        with contextlib.redirect_stdout(out):
            mc, th = train.validate(nc, m, dev, mega, num_to_id, loader)
        return float(mc), {k: float(v) for k, v in th.items()}, out.getvalue()

    calls = {"n": 0}
    real = type(m).forward_distinct

    def spy(self, *args):
        calls["n"] += 1
        return real(self, *args)
    monkeypatch.setattr(type(m), "forward_distinct", spy)
    mc_i, th_i, rep_i = run()
    assert calls["n"] == 1  # the embed-once pass ran
    cache = loader._dev_distinct
    mc_i2, th_i2, rep_i2 = run()
    assert loader._dev_distinct is cache and (mc_i2, th_i2, rep_i2) == (mc_i, th_i, rep_i)  # deterministic, cache reused
    monkeypatch.setenv("NPLDA_VALIDATE_DENSE", "1")
    mc_d, th_d, rep_d = run()
    assert calls["n"] == 2
    assert abs(mc_i - mc_d) <= 1e-4
    lines = lambda rep: [ln for ln in rep.splitlines() if ln.strip()][:3]  # noqa: E731  (C_det, soft C_det, C_min)
    for ln_i, ln_d in zip(lines(rep_i), lines(rep_d)):
        assert abs(float(ln_i.split(":")[-1]) - float(ln_d.split(":")[-1])) <= 2e-4, (ln_i, ln_d)


def test_validate_list_with_few_repeats_keeps_the_dense_pass(hip_lib, monkeypatch):
    from neuralplda_amd import models, train
    rng = np.random.default_rng(3)
    mega, num_to_id, loader, *_ = _speaker_set(rng, 4000, 3000)  # 3 000 trials over up to 4 000 utterances
    nc = NC(D1=150, D2=150)
    m = model_from(rand_params(rng, 512, 150, 150), nc, thetas=[-0.5, -0.3])
    monkeypatch.setattr(models.NeuralPlda, "forward_distinct", lambda *a: (_ for _ in ()).throw(AssertionError("not here")))
    with contextlib.redirect_stdout(io.StringIO()):
        mc, _ = train.validate(nc, m, torch.device("cuda"), mega, num_to_id, loader)
    assert np.isfinite(float(mc))


def test_validate_with_a_dplda_model(hip_lib):
    """DPlda inherits forward_rows but has no PLDA layer — validate() must take the loader-batched pass for it
    (main_dplda's first validate() initialises the thresholds: xvector_DPlda_pytorch.py:56-83)."""
    from neuralplda_amd import metrics, models, train
    rng = np.random.default_rng(8)
    mega, num_to_id, loader, mat, a, b, lab = _speaker_set(rng, 600, 30000, n_spk=40)
    nc = NC(D1=170, D2=170, loss="crossentropy")
    nc.batch_size = 2048
    torch.manual_seed(2)
    dp = models.DPlda(nc).cuda()
    with torch.no_grad():
        dp.logistic_regres.weight.mul_(0.05)
    with contextlib.redirect_stdout(io.StringIO()):
        mc, th = train.validate(nc, dp, torch.device("cuda"), mega, num_to_id, loader, update_thresholds=True)
    X = torch.from_numpy(mat).cuda()
    with torch.no_grad():
        s = torch.cat([dp(X[torch.from_numpy(a[lo:lo + 10240]).cuda()], X[torch.from_numpy(b[lo:lo + 10240]).cuda()])
                       for lo in range(0, len(a), 10240)])
    mc_ref, th_ref = metrics.minc(s, torch.from_numpy(lab).cuda(), nc.beta)
    assert float(mc) == float(mc_ref)  # the same batches through the same kernels
    assert [float(dp.threshold[bb].detach()) for bb in dp.beta] == [float(th_ref[bb]) for bb in nc.beta]


def test_validate_first_run_under_inference_mode_in_a_fresh_process(hip_lib):
    """the loader's mapped-column / distinct-row caches keyed on `num_to_row._version`, which an INFERENCE tensor
    does not have — a validate() whose first call (the one that builds the row map) runs under torch.inference_mode() raised
    RuntimeError.  Fresh process: nothing is cached yet.  The second call (outside inference mode, a NEW map object for the
    same dict) must not hit columns cached for the first one's map by address."""
    import os, subprocess, sys, textwrap
    code = textwrap.dedent("""
        import contextlib, io
        import numpy as np, torch
        from neuralplda_amd import train
        from tests.test_validate_gpu import _speaker_set
        from tests.test_train_gpu import NC, model_from, rand_params
        rng = np.random.default_rng(5)
        mega, num_to_id, loader, mat, a, b, lab = _speaker_set(rng, 3000, 20000)
        nc = NC(D1=150, D2=150)
        m = model_from(rand_params(rng, 512, 150, 150), nc, thetas=[-0.5, -0.3])
        dev = torch.device("cuda")
        with torch.inference_mode(), contextlib.redirect_stdout(io.StringIO()):
            r1 = train.validate(nc, m, dev, mega, num_to_id, loader)
            n, e1, e2, el = loader.device_columns(dev, train._device_table(mega, num_to_id, dev)[1])
            assert train._device_table(mega, num_to_id, dev)[1].is_inference()
        with contextlib.redirect_stdout(io.StringIO()):
            r2 = train.validate(nc, m, dev, mega, num_to_id, loader)
        assert r1 == r2, (r1, r2)
        # a different map of the same size must never be served the first map's cached columns
        other = torch.arange(3000, device=dev).flip(0).contiguous()
        n2, f1, f2, _ = loader.device_columns(dev, other)
        assert torch.equal(f1, 2999 - e1.long()) and torch.equal(f2, 2999 - e2.long())
        print("validate ok", r1)
    """)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "validate ok" in r.stdout
