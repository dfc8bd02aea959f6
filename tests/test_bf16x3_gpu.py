"""GPU: the opt-in split-bf16 scoring kernels (six bf16 MFMA passes per fp32 product) against the fp64 oracle and
the reference's own scores — same tolerance as the exact-fp32 kernels (|ds| <= 2e-5 + 1e-5 |s|)."""
import os

import numpy as np
import pytest
import torch

from oracle import nplda_oracle as orc

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ATOL, RTOL = 2e-5, 1e-5


def rand_params(rng, D0, D1, D2):
    k1, k2 = 1 / np.sqrt(D0), 1 / np.sqrt(D1)
    return orc.Params(rng.uniform(-k1, k1, (D1, D0)).astype(np.float32), rng.uniform(-k1, k1, D1).astype(np.float32),
                      rng.uniform(-k2, k2, (D2, D1)).astype(np.float32), rng.uniform(-k2, k2, D2).astype(np.float32),
                      rng.uniform(0, 1, D2).astype(np.float32), rng.uniform(0, 1, D2).astype(np.float32))


@pytest.mark.parametrize("D0,D1,D2", [(512, 150, 150), (512, 170, 170), (512, 170, 150), (64, 40, 24), (72, 150, 150),
                                       (32, 16, 16), (512, 192, 180), (516, 100, 60)])
@pytest.mark.parametrize("B", [1, 15, 1000, 20037])
def test_bf16x3_scores_match_oracle(hip_lib, D0, D1, D2, B):
    from neuralplda_amd import ops
    rng = np.random.default_rng(31 * D1 + B)
    p = rand_params(rng, D0, D1, D2)
    x1 = rng.standard_normal((B, D0)).astype(np.float32)
    x2 = rng.standard_normal((B, D0)).astype(np.float32)
    dev = [torch.from_numpy(a).cuda() for a in p.tensors()]
    pk3 = ops.pack_params(*dev, precision="bf16x3")
    X1, X2 = torch.from_numpy(x1).cuda(), torch.from_numpy(x2).cuda()
    s3 = ops.score_pairs(X1, X2, pk3).cpu().numpy()
    ref = orc.forward(x1, x2, p, np.float64)
    assert np.all(np.abs(s3 - ref) <= ATOL + RTOL * np.abs(ref)), np.abs(s3 - ref).max()
    # about as close to fp64 as the exact-fp32 kernel is
    s32 = ops.score_pairs(X1, X2, ops.pack_params(*dev)).cpu().numpy()
    assert np.abs(s3 - ref).max() <= 4 * max(np.abs(s32 - ref).max(), 1e-6)
    z3, q3 = ops.embed(X1, pk3)
    zr = orc.extract_plda_embeddings(x1, p, np.float64)
    np.testing.assert_allclose(z3.cpu().numpy()[:, :D2], zr, atol=3e-6, rtol=1e-5)
    assert np.all(z3.cpu().numpy()[:, D2:] == 0)
    np.testing.assert_allclose(q3.cpu().numpy(), orc.self_term(zr, p, np.float64), atol=3e-6, rtol=1e-5)


def test_bf16x3_reference_scores_and_metrics(hip_lib):
    """Kaldi-initialised model (G1/G2) and the end-to-end set (G9): scores vs the reference, minDCF / EER unchanged."""
    from neuralplda_amd import metrics, ops
    from tests import synth
    g1 = np.load(os.path.join(G, "g1_kaldi_params.npz"))
    f = np.load(os.path.join(G, "g2_forward_kaldi170.npz"))
    dev = [torch.from_numpy(g1[k]).cuda() for k in ("W1", "b1", "W2", "b2", "P_sqrt", "Q")]
    pk3 = ops.pack_params(*dev, precision="bf16x3")
    s = ops.score_pairs(torch.from_numpy(f["x1"]).cuda(), torch.from_numpy(f["x2"]).cuda(), pk3).cpu().numpy()
    assert np.all(np.abs(s - f["s64"]) <= ATOL + RTOL * np.abs(f["s64"]))
    assert np.abs(s - f["s64"]).max() < 8e-6
    g = np.load(os.path.join(G, "g9_e2e_kaldi170.npz"))
    x, spk = synth.speaker_structured_xvectors(g1["W1"], g1["b1"], g1["W2"].astype(np.float64), g1["plda_mean"],
                                               g1["psi"], int(g["S"]), int(g["U"]), float(g["c"]), int(g["seed"]))
    assert np.allclose(x[:4], g["x_head"], atol=1e-4) and np.allclose(x.sum(axis=0, dtype=np.float64), g["x_colsum"], atol=1e-2), \
        "numpy RNG stream differs from the fixture generator: regenerate g9 (tests/golden/make_golden.py)"
    X = torch.from_numpy(x).cuda()
    s9 = ops.score_pairs(X[torch.from_numpy(g["i1"]).cuda()], X[torch.from_numpy(g["i2"]).cuda()], pk3).cpu().numpy()
    assert np.all(np.abs(s9 - g["s"]) <= ATOL + RTOL * np.abs(g["s"]))
    mc, _ = metrics.minc(torch.from_numpy(s9), torch.from_numpy(g["t"]), [99.0, 199.0])
    assert abs(mc.item() - float(g["minc_ref"])) <= 1e-3
    assert abs(metrics.eer(torch.from_numpy(s9), torch.from_numpy(g["t"])) - orc.eer(g["s"], g["t"])) <= 1e-3


def test_bf16x3_image_is_scoring_only(hip_lib):
    from neuralplda_amd import ops
    p = rand_params(np.random.default_rng(0), 512, 150, 150)
    pk3 = ops.pack_params(*[torch.from_numpy(a).cuda() for a in p.tensors()], precision="bf16x3")
    x = torch.randn(8, 512, device="cuda")
    with pytest.raises(ValueError):
        ops.forward_train(x, x, pk3)


def test_bf16_rows_through_a_bf16x3_model_are_widened(hip_lib):
    """NeuralPlda.forward under no_grad takes bfloat16 rows straight to ops.score_pairs; with scoring_precision = 'bf16x3'
    the streaming bf16-rows kernel does not apply, so the rows must be widened there (they used to reach the fp32 check
    and raise TypeError): same scores as the explicit .float() call."""
    from neuralplda_amd import models

    class NC:
        xvector_dim, layer1_LDA_dim, layer2_PLDA_spkfactor_dim = 512, 150, 150
        beta, alpha, device, loss = [99.0, 199.0], 15.0, "cuda", "SoftCdet"

    torch.manual_seed(0)
    m = models.NeuralPlda(NC()).cuda()
    m.scoring_precision = "bf16x3"
    x1 = torch.randn(300, 512, device="cuda").bfloat16()
    x2 = torch.randn(300, 512, device="cuda").bfloat16()
    with torch.no_grad():
        a = m(x1, x2)
        b = m(x1.float(), x2.float())
    assert a.dtype == torch.float32 and torch.equal(a, b)
