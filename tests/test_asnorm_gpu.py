"""GPU parity of the AS-norm kernels (cohort MFMA GEMM, per-row radix-select statistics, per-trial
normalisation) vs the oracle and the golden output of the reference script (G6)."""
import io
import os

import numpy as np
import pytest
import torch

from oracle import nplda_oracle as orc

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rand_params(rng, D0, D1, D2):
    k1, k2 = 1 / np.sqrt(D0), 1 / np.sqrt(D1)
    return orc.Params(rng.uniform(-k1, k1, (D1, D0)).astype(np.float32), rng.uniform(-k1, k1, D1).astype(np.float32),
                      rng.uniform(-k2, k2, (D2, D1)).astype(np.float32), rng.uniform(-k2, k2, D2).astype(np.float32),
                      rng.uniform(0, 1, D2).astype(np.float32), rng.uniform(0, 1, D2).astype(np.float32))


@pytest.mark.parametrize("R,M,topn", [(5, 600, 500), (33, 10000, 500), (3, 7, 500), (4, 50000, 500), (9, 1000, 1),
                                      (2, 40000, 39999)])
@pytest.mark.parametrize("select", ["lowest", "highest"])
def test_row_stats_matches_sort_then_slice(hip_lib, R, M, topn, select):
    from neuralplda_amd import ops
    rng = np.random.default_rng(R * 1000 + M)
    S = (rng.standard_normal((R, M)) * (1 + np.arange(R))[:, None] - 0.7).astype(np.float32)
    S[0, : M // 2] = np.round(S[0, : M // 2], 1)  # heavy ties, also across the selection boundary
    if R > 1:
        S[1] = 0.25                                  # a constant row: std exactly 0
    got = ops.row_stats(torch.from_numpy(S).cuda(), topn=topn, select=select).cpu().numpy()
    ref = orc.cohort_stats(S, topn, select)
    np.testing.assert_allclose(got[:, 0], ref[:, 0], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(got[:, 2], ref[:, 2], rtol=1e-12, atol=1e-12)
    # std via E[x^2] - mean^2 in fp64: absolute accuracy ~1e-8 of the scale
    scale = np.abs(S).max(axis=1) + 1
    assert np.all(np.abs(got[:, 1] - ref[:, 1]) <= 1e-7 * scale)
    assert np.all(np.abs(got[:, 3] - ref[:, 3]) <= 1e-7 * scale)
    if R > 1:
        assert got[1, 1] < 1e-7 and got[1, 3] < 1e-7


@pytest.mark.parametrize("D", [150, 170, 16, 24])  # 16 / 24: one / two k16 stages per tile (the single-stage pipeline)
@pytest.mark.parametrize("R,M", [(200, 1000), (130, 257), (140, 24700)])  # the last spans two super-bands of column tiles
def test_cohort_stats_full_pipeline(hip_lib, D, R, M):
    from neuralplda_amd import ops
    rng = np.random.default_rng(D + R)
    p = rand_params(rng, 512, D, D)
    xr = rng.standard_normal((R, 512)).astype(np.float32)
    xc = rng.standard_normal((M, 512)).astype(np.float32)
    packed = ops.pack_params(*[torch.from_numpy(a).cuda() for a in p.tensors()])
    zr, qr = ops.embed(torch.from_numpy(xr).cuda(), packed)
    zc, qc = ops.embed(torch.from_numpy(xc).cuda(), packed)
    topn = 100
    got = ops.cohort_stats(zr, qr, zc, qc, packed, topn=topn).cpu().numpy()
    # oracle: NeuralPlda.forward on the expanded pair list (fp64), then sort-then-slice statistics
    z_r = orc.extract_plda_embeddings(xr, p, np.float64)
    z_c = orc.extract_plda_embeddings(xc, p, np.float64)
    C = orc.cohort_scores(z_r, z_c, p, np.float64)
    np.testing.assert_allclose(C[:3, :5], [[orc.forward(xr[i:i + 1], xc[j:j + 1], p, np.float64)[0] for j in range(5)]
                                           for i in range(3)], rtol=1e-9)
    ref = orc.cohort_stats(C, topn)
    np.testing.assert_allclose(got, ref, atol=2e-5, rtol=2e-5)
    # chunked workspace (forces several GEMM + select rounds) gives the same bits
    got2 = ops.cohort_stats(zr, qr, zc, qc, packed, topn=topn, max_ws_bytes=40 * ((M + 3) // 4 * 4) * 4).cpu().numpy()
    np.testing.assert_array_equal(got, got2)
    hi = ops.cohort_stats(zr, qr, zc, qc, packed, topn=topn, select="highest").cpu().numpy()
    np.testing.assert_allclose(hi, orc.cohort_stats(C, topn, "highest"), atol=2e-5, rtol=2e-5)


def test_asnorm_apply_and_scorefile_golden(hip_lib, tmp_path):
    from neuralplda_amd import adaptive_score_normalization as asn, ops
    g = np.load(os.path.join(G, "g6_asnorm.npz"))
    ids = [str(i) for i in g["ids"]]
    # kernel-level: statistics of the golden cohort matrix + per-trial normalisation vs the oracle (fp64)
    C = g["cohort"]
    stats = ops.row_stats(torch.from_numpy(C.astype(np.float32)).cuda(), topn=int(g["topn"]))
    row = {k: i for i, k in enumerate(ids)}
    ie = np.asarray([row[str(e)] for e in g["enroll"]])
    it = np.asarray([row[str(t).replace(".sph", "")] for t in g["test"]])
    out = ops.asnorm_apply(torch.from_numpy(g["raw"]), ie, it, stats).cpu().numpy()
    ref = orc.asnorm_apply(g["raw"], ie, it, orc.cohort_stats(C.astype(np.float32), int(g["topn"])))
    np.testing.assert_allclose(out, ref, rtol=1e-9, atol=1e-9)
    bad = ops.asnorm_apply(torch.tensor([0.5]), [len(ids)], [0], stats).cpu().numpy()
    assert np.all(np.isnan(bad))
    # file-level: same TSVs in, the reference script's four files out
    rawf, cohf = tmp_path / "raw.tsv", tmp_path / "cohort.tsv"
    with open(rawf, "w") as f:
        f.write("modelid\tsegmentid\tside\tLLR\n")
        for e, t, sd, v in zip(g["enroll"], g["test"], g["side"], g["raw"]):
            f.write(f"{e}\t{t}\t{sd}\t{float(v)!r}\n")
    with open(cohf, "w") as f:
        f.write("id\tcohort\tLLR\n")
        for i, a in enumerate(ids):
            for j in range(C.shape[1]):
                f.write(f"{a}\tcoh{j:04d}\t{float(C[i, j])!r}\n")
    res = asn.normalize_scorefile(str(rawf), str(cohf))
    for k in ("znorm", "tnorm", "snorm", "asnorm1"):
        # cohort scores go through fp32 on the device: 1e-6 relative on the normalised score
        np.testing.assert_allclose(res[k], g[k], rtol=2e-6, atol=2e-6, err_msg=k)
        txt = open(str(rawf) + f"_{k}.tsv").read()
        ref_txt = str(g[k + "_text"])
        assert txt.splitlines()[0] == ref_txt.splitlines()[0]           # "# modelid\tsegmentid\tside\tLLR"
        tab = np.genfromtxt(io.StringIO(txt), dtype=str, skip_header=1)
        rtab = np.genfromtxt(io.StringIO(ref_txt), dtype=str, skip_header=1)
        np.testing.assert_array_equal(tab[:, :-1], rtab[:, :-1])
        np.testing.assert_allclose(tab[:, -1].astype(float), rtab[:, -1].astype(float), rtol=2e-6, atol=2e-6)


def test_asnorm_scores_pipeline(hip_lib):
    from neuralplda_amd import adaptive_score_normalization as asn, models
    rng = np.random.default_rng(4)

    class NC:
        xvector_dim, layer1_LDA_dim, layer2_PLDA_spkfactor_dim = 512, 150, 150
        beta, alpha, device, loss = [99.0], 15.0, "cuda", "SoftCdet"

    torch.manual_seed(4)
    m = models.NeuralPlda(NC()).cuda()
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    p = orc.Params(sd["centering_and_LDA.weight"], sd["centering_and_LDA.bias"], sd["centering_and_wccn_plda.weight"],
                   sd["centering_and_wccn_plda.bias"], sd["P_sqrt"], sd["Q"])
    R, M, T = 60, 700, 500
    xr = rng.standard_normal((R, 512)).astype(np.float32)
    xc = rng.standard_normal((M, 512)).astype(np.float32)
    ie, it = rng.integers(0, 20, T), rng.integers(20, R, T)
    raw = orc.forward(xr[ie], xr[it], p, np.float64)
    out = asn.asnorm_scores(m, torch.from_numpy(xr).cuda(), torch.from_numpy(xc).cuda(), raw, ie, it, topN=200)
    zr, zc = orc.extract_plda_embeddings(xr, p, np.float64), orc.extract_plda_embeddings(xc, p, np.float64)
    ref = orc.asnorm_apply(raw, ie, it, orc.cohort_stats(orc.cohort_scores(zr, zc, p, np.float64), 200))
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=5e-4, atol=5e-4)


def test_cfg3_full_size_properties(hip_lib):
    """BASELINE cfg3 at full size on one GPU — 22 000 enroll/test rows x 10 000 cohort utterances, top-500, 2 M trials,
    D = 170 — through size-independent properties: sampled rows and trials against the fp64 oracle, row-order
    independence (bit-exact), and ordering invariants of the statistics."""
    from neuralplda_amd import ops
    rng = np.random.default_rng(33)
    D, R, M, T, topn = 170, 22000, 10000, 2_000_000, 500
    p = rand_params(rng, 512, D, D)
    packed = ops.pack_params(*[torch.from_numpy(a).cuda() for a in p.tensors()])
    gen = torch.Generator(device="cuda").manual_seed(5)
    xr = torch.randn(R, 512, device="cuda", generator=gen)
    xc = torch.randn(M, 512, device="cuda", generator=gen)
    zr, qr = ops.embed(xr, packed)
    zc, qc = ops.embed(xc, packed)
    stats = ops.cohort_stats(zr, qr, zc, qc, packed, topn=topn)
    assert stats.shape == (R, 4) and stats.dtype == torch.float64 and bool(torch.isfinite(stats).all())
    st = stats.cpu().numpy()
    # the 500 smallest of 10 000 scores: mean below the row mean, spread below the row spread
    assert (st[:, 2] < st[:, 0]).all() and (st[:, 3] < st[:, 1]).all() and (st[:, 1] > 0).all()
    # sampled rows vs the oracle (fp64 scores of the expanded pair list, sort-then-slice)
    rows = rng.choice(R, 48, replace=False)
    z_r = orc.extract_plda_embeddings(xr[torch.from_numpy(rows).cuda()].cpu().numpy(), p, np.float64)
    z_c = orc.extract_plda_embeddings(xc.cpu().numpy(), p, np.float64)
    ref = orc.cohort_stats(orc.cohort_scores(z_r, z_c, p, np.float64), topn)
    np.testing.assert_allclose(st[rows], ref, atol=2e-5, rtol=2e-5)
    # a row's statistics do not depend on where the row sits
    perm = torch.randperm(R, device="cuda", generator=gen)
    assert torch.equal(ops.cohort_stats(zr[perm], qr[perm], zc, qc, packed, topn=topn), stats[perm])
    # 2 M trials normalised on the device; a sample vs the oracle
    ie = torch.randint(0, 2000, (T,), device="cuda", generator=gen)
    it = torch.randint(2000, R, (T,), device="cuda", generator=gen)
    raw = ops.score_indexed(zr, qr, ie, it, packed).double()
    out = ops.asnorm_apply(raw, ie, it, stats)
    assert out.shape == (T, 4) and bool(torch.isfinite(out).all())
    k = torch.from_numpy(rng.choice(T, 5000, replace=False)).cuda()
    refn = orc.asnorm_apply(raw[k].cpu().numpy(), ie[k].cpu().numpy(), it[k].cpu().numpy(), st)
    np.testing.assert_allclose(out[k].cpu().numpy(), refn, rtol=1e-12, atol=1e-12)
    # snorm is the mean of znorm and tnorm for every trial
    assert float((out[:, 2] - (out[:, 0] + out[:, 1]) / 2).abs().max()) < 1e-12


@pytest.mark.parametrize("select", ["lowest", "highest"])
def test_row_stats_shapes_that_defeat_the_quantile_shortcut(hip_lib, select):
    """The quantile-bracket shortcut of row_stats_kernel only fires when its bracket provably holds the rank; rows
    built to defeat it (bimodal, heavy-tailed, coarse quantisation with thousands of ties, a spike exactly at the
    selection boundary, mostly-constant rows) must take the general search and still equal sort-then-slice."""
    from neuralplda_amd import ops
    rng = np.random.default_rng(17)
    M, topn = 10000, 500
    rows = [
        np.concatenate([rng.normal(-50, 0.1, M // 2), rng.normal(80, 0.1, M - M // 2)]),          # bimodal
        rng.standard_cauchy(M) * 3,                                                                # heavy tails
        np.round(rng.standard_normal(M) * 2) / 2,                                                  # ~20 distinct values
        np.where(rng.random(M) < 0.9, 1.5, rng.standard_normal(M)),                                # 90 % constant
        np.concatenate([np.full(400, -3.0), np.full(300, -2.999), rng.standard_normal(M - 700)]),  # spike at the boundary
        np.concatenate([rng.standard_normal(M - 700), np.full(400, 3.0), np.full(300, 2.999)]),    # ... for "highest"
        rng.exponential(1.0, M),                                                                   # skewed
        rng.standard_normal(M) * 1e-30,                                                            # tiny magnitudes
        np.sort(rng.standard_normal(M)),                                                           # plain normal (shortcut)
    ]
    # normal rows with a run of equal values straddling the selection boundary on either end: the shortcut fires and
    # its rank slots must resolve the ties (unwritten-slot rule of the kernel)
    z = np.sort(rng.standard_normal(M))
    z[495:504] = z[499]
    z[M - 504:M - 495] = z[M - 500]
    z[470:474] = z[471]
    rows.append(rng.permutation(z))
    S = np.stack(rows).astype(np.float32)
    S[:, ::997] += 0.0  # keep dtype
    got = ops.row_stats(torch.from_numpy(S).cuda(), topn=topn, select=select).cpu().numpy()
    ref = orc.cohort_stats(S, topn, select)
    scale = np.abs(S).max(axis=1) + 1e-30
    np.testing.assert_allclose(got[:, 0], ref[:, 0], atol=1e-9 * scale.max(), rtol=1e-10)
    assert np.all(np.abs(got[:, 2] - ref[:, 2]) <= 1e-9 * scale)
    assert np.all(np.abs(got[:, 1] - ref[:, 1]) <= 2e-7 * scale)
    assert np.all(np.abs(got[:, 3] - ref[:, 3]) <= 2e-7 * scale)
    again = ops.row_stats(torch.from_numpy(S).cuda(), topn=topn, select=select).cpu().numpy()
    assert np.array_equal(got, again)  # bit-reproducible (rank-slot summation, no atomics on values)


def test_row_stats_random_sweep(hip_lib):
    """Random (M, topn, distribution, select) sweep against sort-then-slice: odd row lengths (16-byte tail groups), rows at
    the LDS / no-LDS boundary, top-N from 1 to beyond M, mixtures with exact duplicates."""
    from neuralplda_amd import ops
    rng = np.random.default_rng(99)
    Ms = [64, 65, 66, 67, 101, 513, 2047, 4099, 10239, 10241, 20003, 37999, 38001, 38005]
    for M in Ms:
        for _ in range(2):
            topn = int(rng.choice([1, 2, 7, max(1, M // 20), max(1, M // 2), M - 1, M, M + 5]))
            kind = int(rng.integers(0, 4))
            if kind == 0:
                row = rng.standard_normal((3, M))
            elif kind == 1:
                row = rng.uniform(-5, 5, (3, M))
            elif kind == 2:
                row = np.round(rng.standard_normal((3, M)) * 4) / 4 + rng.standard_normal((3, 1))
            else:
                row = np.where(rng.random((3, M)) < 0.3, rng.standard_normal((3, 1)), rng.standard_normal((3, M)) * 3)
            S = row.astype(np.float32)
            for select in ("lowest", "highest"):
                got = ops.row_stats(torch.from_numpy(S).cuda(), topn=topn, select=select).cpu().numpy()
                ref = orc.cohort_stats(S, topn, select)
                scale = np.abs(S).max(axis=1) + 1
                msg = f"M={M} topn={topn} kind={kind} {select}"
                np.testing.assert_allclose(got[:, 0], ref[:, 0], rtol=1e-11, atol=1e-11, err_msg=msg)
                np.testing.assert_allclose(got[:, 2], ref[:, 2], rtol=1e-11, atol=1e-11, err_msg=msg)
                assert np.all(np.abs(got[:, 1] - ref[:, 1]) <= 2e-7 * scale), msg
                assert np.all(np.abs(got[:, 3] - ref[:, 3]) <= 2e-7 * scale), msg


def test_cohort_stats_shape_sweep(hip_lib):
    """Row / cohort counts around the 128-wide tile edges, single rows / columns, cohort sizes that are not a multiple of
    4 (padded score rows) — the persistent, counter-driven GEMM must cover every tile exactly once."""
    from neuralplda_amd import ops
    rng = np.random.default_rng(41)
    D = 24
    p = rand_params(rng, 512, D, D)
    packed = ops.pack_params(*[torch.from_numpy(a).cuda() for a in p.tensors()])
    for R in (1, 7, 127, 128, 129, 300):
        for M in (1, 3, 127, 128, 129, 1025):
            xr = rng.standard_normal((R, 512)).astype(np.float32)
            xc = rng.standard_normal((M, 512)).astype(np.float32)
            zr, qr = ops.embed(torch.from_numpy(xr).cuda(), packed)
            zc, qc = ops.embed(torch.from_numpy(xc).cuda(), packed)
            topn = min(5, M)
            got = ops.cohort_stats(zr, qr, zc, qc, packed, topn=topn).cpu().numpy()
            C = orc.cohort_scores(orc.extract_plda_embeddings(xr, p, np.float64),
                                  orc.extract_plda_embeddings(xc, p, np.float64), p, np.float64)
            np.testing.assert_allclose(got, orc.cohort_stats(C, topn), atol=2e-5, rtol=2e-5, err_msg=f"R={R} M={M}")


def test_prepared_cohort_gives_the_same_statistics(hip_lib):
    """nplda_cohort_prepare_f32 + nplda_cohort_stats_prepared_f32 (a CohortState: the cohort's pre-pass once per (model,
    cohort, top-N)) against nplda_cohort_stats_f32: the same kernels on the same inputs — bit-identical statistics — for
    whole tables, row shards and both selections; asnorm_scores(cohort=state) equals asnorm_scores(x_cohort)."""
    from neuralplda_amd import adaptive_score_normalization as asn
    from neuralplda_amd import models, ops
    from tests.test_train_gpu import NC, model_from, rand_params
    rng = np.random.default_rng(41)
    D, R, M = 150, 1500, 6000
    m = model_from(rand_params(rng, 512, D, D), NC(D1=D, D2=D))
    xr = torch.from_numpy(rng.standard_normal((R, 512)).astype(np.float32)).cuda()
    xc = torch.from_numpy(rng.standard_normal((M, 512)).astype(np.float32)).cuda()
    state = asn.CohortState.build(m, xc, topN=500)
    assert state.prepared.state is not None  # the fused path's shape: there IS something to prepare
    packed = state.packed
    zr, qr = ops.embed(xr, packed)
    for sel in ("lowest", "highest"):
        ref = ops.cohort_stats(zr, qr, state.z_coh, state.q_coh, packed, topn=500, select=sel)
        got = ops.cohort_stats(zr, qr, state.z_coh, state.q_coh, packed, topn=500, select=sel, prepared=state.prepared)
        assert torch.equal(ref, got)
        part = ops.cohort_stats(zr[300:900], qr[300:900], state.z_coh, state.q_coh, packed, topn=500, select=sel,
                                prepared=state.prepared)
        assert torch.equal(part, ref[300:900])  # a rank's row shard gives exactly the single-GPU rows
    T = 20000
    raw = torch.from_numpy(rng.standard_normal(T)).cuda()
    ie = torch.from_numpy(rng.integers(0, 200, T)).cuda()
    it = torch.from_numpy(rng.integers(200, R, T)).cuda()
    a = asn.asnorm_scores(m, xr, xc, raw, ie, it, topN=500)
    b = asn.asnorm_scores(m, xr, None, raw, ie, it, topN=500, cohort=state)
    assert torch.equal(a, b)
    with pytest.raises(ValueError):
        asn.asnorm_scores(m, xr, None, raw, ie, it, topN=300, cohort=state)
    # a small cohort takes the spilling path: nothing to prepare, same results
    small = asn.CohortState.build(m, xc[:1000], topN=100)
    assert small.prepared.state is None
    s1 = ops.cohort_stats(zr, qr, small.z_coh, small.q_coh, packed, topn=100)
    s2 = ops.cohort_stats(zr, qr, small.z_coh, small.q_coh, packed, topn=100, prepared=small.prepared)
    assert torch.equal(s1, s2)
    # a state outlives no parameter change: an in-place update (what an optimiser step is) is noticed
    with torch.no_grad():
        m.Q.mul_(1.01)
    with pytest.raises(ValueError, match="changed since"):
        asn.asnorm_scores(m, xr, None, raw, ie, it, topN=500, cohort=state)
    fresh = asn.CohortState.build(m, xc, topN=500)
    c = asn.asnorm_scores(m, xr, None, raw, ie, it, topN=500, cohort=fresh)
    assert torch.equal(c, asn.asnorm_scores(m, xr, xc, raw, ie, it, topN=500)) and not torch.equal(a, c)
