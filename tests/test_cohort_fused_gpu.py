"""The fused cohort-statistics path (csrc/nplda_cohort_fused.hip: statistics in the score GEMM's epilogue, no score
matrix) against the fp64 oracle (NeuralPlda.forward on the expanded pair list, then sort-then-slice:
adaptive_score_normalization.py:32-36) and against the spilling path it replaces.  Tolerance 2e-5 like
tests/test_asnorm_gpu.py; run-to-run and row-order bit-reproducibility."""
import numpy as np
import pytest
import torch

from oracle import nplda_oracle as orc
from tests.test_train_gpu import rand_params

pytestmark = pytest.mark.gpu


def setup(D, R, M, seed, coh_fn=None):
    from neuralplda_amd import ops
    rng = np.random.default_rng(seed)
    p = rand_params(rng, 512, D, D)
    xr = rng.standard_normal((R, 512)).astype(np.float32)
    xc = rng.standard_normal((M, 512)).astype(np.float32)
    if coh_fn is not None:
        xc = coh_fn(rng, xc)
    packed = ops.pack_params(*[torch.from_numpy(a).cuda() for a in p.tensors()])
    zr, qr = ops.embed(torch.from_numpy(xr).cuda(), packed)
    zc, qc = ops.embed(torch.from_numpy(xc).cuda(), packed)
    C = orc.cohort_scores(orc.extract_plda_embeddings(xr, p, np.float64), orc.extract_plda_embeddings(xc, p, np.float64),
                          p, np.float64)
    return ops, packed, zr, qr, zc, qc, C


@pytest.mark.parametrize("D,R,M,topn", [(150, 300, 10000, 500), (170, 129, 4096, 100), (24, 260, 24700, 100),
                                        (16, 128, 5000, 37), (170, 1, 10000, 500), (150, 517, 9999, 1),
                                        (64, 200, 4500, 50), (128, 150, 5000, 200), (192, 140, 4200, 64)])
def test_fused_matches_oracle_and_the_spilling_path(hip_lib, D, R, M, topn):
    ops, packed, zr, qr, zc, qc, C = setup(D, R, M, D + R + M)
    lib = hip_lib
    assert lib.nplda_cohort_fused_min_workspace_bytes(M, topn, D, D) > 0  # these shapes take the fused path
    for select in ("lowest", "highest"):
        got, nfb = ops.cohort_stats(zr, qr, zc, qc, packed, topn=topn, select=select, return_fallback_rows=True)
        assert nfb is not None and nfb <= R // 20, (select, nfb)  # Gaussian-ish rows: the proposal brackets nearly all
        ref = orc.cohort_stats(C, topn, select)
        np.testing.assert_allclose(got.cpu().numpy(), ref, atol=2e-5, rtol=2e-5, err_msg=select)
        spill = ops.cohort_stats(zr, qr, zc, qc, packed, topn=topn, select=select, force_spill=True)
        # same fp32 scores underneath: the two paths differ only by the summation order of their fp64 / centred sums
        np.testing.assert_allclose(got.cpu().numpy(), spill.cpu().numpy(), rtol=2e-6, atol=2e-7, err_msg=select)
        # top-N mean: both select the same N scores to rounding.  The fused epilogue forms the score CENTRED on the row's mean,
        # d = acc + (q_m + (q_r - c_r)), and the select kernel adds c_r back in fp64 (round 6); the spilling path forms
        # acc + (q_m + q_r): the same value rounded at a different point (and at D = 150 / 170 the fused GEMM runs its last
        # k-block in 2 / 3 steps over re-ordered columns, round 5) — the fp32 tolerance of SURVEY 8(c), not bit equality
        np.testing.assert_allclose(got[:, 2].cpu().numpy(), spill[:, 2].cpu().numpy(), rtol=2e-6, atol=2e-7,
                                   err_msg=f"{select} (rows on the general path: {nfb})")
        assert torch.equal(got, ops.cohort_stats(zr, qr, zc, qc, packed, topn=topn, select=select))  # bit-reproducible
    # a row's statistics do not depend on its position, on its neighbours, or on the workspace chunking
    got = ops.cohort_stats(zr, qr, zc, qc, packed, topn=topn)
    perm = torch.randperm(R, device="cuda")
    assert torch.equal(ops.cohort_stats(zr[perm].contiguous(), qr[perm].contiguous(), zc, qc, packed, topn=topn), got[perm])
    if R > 128:
        fmin = lib.nplda_cohort_fused_min_workspace_bytes(M, topn, D, D)
        assert torch.equal(ops.cohort_stats(zr, qr, zc, qc, packed, topn=topn, max_ws_bytes=fmin), got)  # 128-row chunks


def _spiky(rng, xc):
    xc[: int(0.6 * len(xc))] = xc[0]                      # 60 % identical cohort utterances: a spike of tied scores
    return xc


def _bimodal(rng, xc):
    xc[::2] += 3.0 * rng.standard_normal(512).astype(np.float32)  # two clusters: rows are far from normal
    return xc


def _outliers(rng, xc):
    xc[:40] *= 25.0                                        # a few huge-norm utterances (same direction after the
    return xc                                              # length normalisation, but they move the moments)


@pytest.mark.parametrize("coh_fn", [_spiky, _bimodal, _outliers])
def test_rows_the_proposal_misses_take_the_general_path(hip_lib, coh_fn):
    """Cohorts built so that the normal-model threshold cannot bracket the N-th smallest score for (some) rows — ties
    far beyond the candidate capacity, bimodal rows: those rows are recomputed by the exact general path, results as
    the oracle's."""
    D, R, M, topn = 150, 200, 6000, 400
    ops, packed, zr, qr, zc, qc, C = setup(D, R, M, 5, coh_fn)
    for select in ("lowest", "highest"):
        got, nfb = ops.cohort_stats(zr, qr, zc, qc, packed, topn=topn, select=select, return_fallback_rows=True)
        got = got.cpu().numpy()
        print(coh_fn.__name__, select, "rows on the general path:", nfb)
        ref = orc.cohort_stats(C, topn, select)
        scale = np.maximum(np.abs(ref).max(axis=1, keepdims=True), 1.0)
        assert np.all(np.abs(got - ref) <= 2e-5 * scale), (coh_fn.__name__, select, np.abs(got - ref).max())
        assert np.array_equal(got, ops.cohort_stats(zr, qr, zc, qc, packed, topn=topn, select=select).cpu().numpy())


def test_ineligible_shapes_keep_the_spilling_path(hip_lib):
    assert hip_lib.nplda_cohort_fused_min_workspace_bytes(1000, 100, 150, 150) == 0        # small cohort
    assert hip_lib.nplda_cohort_fused_min_workspace_bytes(10000, 4000, 150, 150) == 0      # top-N too large for the lists
    assert hip_lib.nplda_cohort_fused_min_workspace_bytes(10000, 500, 150, 150) > 0
    ops, packed, zr, qr, zc, qc, C = setup(150, 64, 8000, 9)
    got = ops.cohort_stats(zr, qr, zc, qc, packed, topn=3000).cpu().numpy()                # 3000 of 8000: spilling path
    np.testing.assert_allclose(got, orc.cohort_stats(C, 3000), atol=2e-5, rtol=2e-5)


def test_rows_with_more_candidates_than_the_key_run_take_the_exact_path(hip_lib):
    """The select kernel's LDS run holds NPLDA_COHORT_CAP x the proposal (1.4 by default); a row that brings more goes to the
    fail list and through the general path.  With the factor forced to 0.9 (read once per process, hence the child) a good
    part of the rows overflow — and every row still equals the spilling path."""
    import os, subprocess, sys, textwrap
    code = textwrap.dedent("""
        import numpy as np, torch
        from tests.test_cohort_fused_gpu import setup
        ops, packed, zr, qr, zc, qc, C = setup(150, 600, 10000, 77)
        got, nfb = ops.cohort_stats(zr, qr, zc, qc, packed, topn=500, return_fallback_rows=True)
        spill = ops.cohort_stats(zr, qr, zc, qc, packed, topn=500, force_spill=True)
        assert 50 <= nfb <= 600, nfb
        np.testing.assert_allclose(got.cpu().numpy(), spill.cpu().numpy(), rtol=2e-6, atol=2e-7)
        np.testing.assert_allclose(got[:, 2].cpu().numpy(), spill[:, 2].cpu().numpy(), rtol=2e-6, atol=2e-7)  # (D = 150: short last k-block)
        print("overflow rows", nfb)
    """)
    env = dict(os.environ, NPLDA_COHORT_CAP="0.9")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "overflow rows" in r.stdout


def test_row_tile_size_does_not_change_a_row(hip_lib):
    """A call with few rows walks 128-row tiles (twice as many, half-size work items: an 8-way row shard of cfg3 is 11 tiles
    of 256 rows for 256 CUs), a large call 256-row tiles.  The same row must get the same four statistics either way, bit
    for bit: what a rank computes for its row shard IS the single-GPU result."""
    from neuralplda_amd import ops
    rng = np.random.default_rng(77)
    D, M, Rbig = 150, 10000, 6400
    p = rand_params(rng, 512, D, D)
    packed = ops.pack_params(*[torch.from_numpy(a).cuda() for a in p.tensors()])
    zr, qr = ops.embed(torch.from_numpy(rng.standard_normal((Rbig, 512)).astype(np.float32)).cuda(), packed)
    zc, qc = ops.embed(torch.from_numpy(rng.standard_normal((M, 512)).astype(np.float32)).cuda(), packed)
    big = ops.cohort_stats(zr, qr, zc, qc, packed, topn=500)                      # 25 tiles x 32 list bands: 256-row tiles
    for lo, hi in ((0, 300), (1000, 3750), (6399, 6400), (5000, 5129)):
        part = ops.cohort_stats(zr[lo:hi].contiguous(), qr[lo:hi].contiguous(), zc, qc, packed, topn=500)  # 128-row tiles
        assert torch.equal(part, big[lo:hi]), (lo, hi)


def test_prepared_cohort_is_bound_to_its_cohort_and_model(hip_lib):
    """a PreparedCohort carries the first moments and the covariance image of ONE cohort under ONE model image;
    cohort_stats(prepared=...) must refuse a state prepared for another cohort / model of the same size (its statistics
    would be silently wrong) and a cohort table modified in place since."""
    ops, packed, zr, qr, zc, qc, C = setup(150, 300, 10000, 11)
    prep = ops.cohort_prepare(zc, qc, packed, topn=500)
    a = ops.cohort_stats(zr, qr, zc, qc, packed, topn=500, prepared=prep)
    b = ops.cohort_stats(zr, qr, zc, qc, packed, topn=500)
    assert torch.allclose(a, b, rtol=1e-12, atol=0)
    zc2, qc2 = zc.clone(), qc.clone()  # same size, another table
    with pytest.raises(ValueError):
        ops.cohort_stats(zr, qr, zc2, qc2, packed, topn=500, prepared=prep)
    zc.mul_(1.0)  # in-place write: the version moves
    with pytest.raises(ValueError):
        ops.cohort_stats(zr, qr, zc, qc, packed, topn=500, prepared=prep)



def test_split_bf16_form_and_fp32_form_of_the_fused_gemm_agree(hip_lib, monkeypatch):
    """Round 6: at NB = 10 / 11 the fused GEMM takes its fp32 operands as THREE bf16 pieces and six
    v_mfma_f32_16x16x32_bf16 passes (hh + hm + mh + hl + lh + mm; what is dropped is below 2^-26 of a product), the form every
    other test of this file now runs.  The fp32-input form stays in the library (NPLDA_COHORT_SPLIT=0, read at every call):
    both against the fp64 oracle at the file's tolerance, against each other at the tolerance the fused path has against the
    spilling path, the split form no further from the oracle than 1.5 x the fp32 form (+ 1e-7) — and a prepared cohort serves
    both (its split image is always built)."""
    for D, R, M, topn in ((150, 600, 10000, 500), (170, 300, 6000, 200)):
        ops, packed, zr, qr, zc, qc, C = setup(D, R, M, 11 + D)
        prep = ops.cohort_prepare(zc, qc, packed, topn=topn)
        for select in ("lowest", "highest"):
            ref = orc.cohort_stats(C, topn, select)
            got = {}
            for split in ("0", "1"):
                monkeypatch.setenv("NPLDA_COHORT_SPLIT", split)
                g, nfb = ops.cohort_stats(zr, qr, zc, qc, packed, topn=topn, select=select, return_fallback_rows=True)
                assert nfb is not None and nfb <= R // 20
                got[split] = g.cpu().numpy()
                np.testing.assert_allclose(got[split], ref, atol=2e-5, rtol=2e-5, err_msg=f"{D} {select} split={split}")
                if select == "lowest":
                    gp = ops.cohort_stats(zr, qr, zc, qc, packed, topn=topn, select=select, prepared=prep)
                    assert torch.allclose(gp, g, rtol=1e-12, atol=0), (D, split)
            monkeypatch.delenv("NPLDA_COHORT_SPLIT")
            a, b = got["0"], got["1"]
            np.testing.assert_allclose(b, a, rtol=2e-6, atol=2e-7, err_msg=f"{D} {select}")
            ea, eb = np.abs(a - ref).max(axis=0), np.abs(b - ref).max(axis=0)
            assert np.all(eb <= 1.5 * ea + 1e-7), (D, select, ea, eb)
