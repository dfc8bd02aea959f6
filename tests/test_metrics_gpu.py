"""GPU parity of the detection-cost sweep (nplda_detcost_sweep_f32): NeuralPlda.minc with the reference's quirks bit for
bit (golden G5 = the reference's own outputs), the exact minimum detection cost and the EER against the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import nplda_oracle as orc

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_minc_reference_semantics_golden(hip_lib):
    from neuralplda_amd import metrics
    g = np.load(os.path.join(G, "g5_metrics.npz"))
    for dev in ("cpu", "cuda"):  # CPU tensors are moved to the device; results come back on the caller's device
        S, T = torch.from_numpy(g["s"]).to(dev), torch.from_numpy(g["t"]).to(dev)
        mc, th = metrics.minc(S, T, [99.0, 199.0])
        assert mc.device.type == dev and mc.dim() == 0
        assert abs(mc.item() - float(g["minc"])) <= 1e-7
        assert th[99.0].item() == np.float32(g["minc_th"][0]) and th[199.0].item() == np.float32(g["minc_th"][1])
        mcs, _ = metrics.minc(torch.from_numpy(g["s_sep"]).to(dev), T, [99.0, 199.0])
        assert abs(mcs.item() - float(g["minc_sep"])) <= 1e-7 and mcs.item() > 0
        assert metrics.minc(torch.from_numpy(g["s_sep"]).to(dev), T, [99.0, 199.0], reference_semantics=False)[0].item() == 0.0
        assert abs(metrics.eer(S, T) - orc.eer(g["s"], g["t"])) < 1e-6
        assert abs(metrics.minc_exact(S, T, [99.0])[0].item() - orc.minc_exact(g["s"], g["t"], [99.0])[0]) < 1e-6


@pytest.mark.parametrize("n,ptgt,quant", [(2, 0.5, None), (17, 0.3, None), (1000, 0.1, 0.05), (4097, 0.02, None),
                                          (300000, 0.01, 0.001), (1 << 20, 0.05, None)])
def test_sweep_vs_oracle(hip_lib, n, ptgt, quant):
    from neuralplda_amd import metrics
    rg = np.random.default_rng(n)
    t = (rg.random(n) < ptgt).astype(np.float32)
    t[0], t[1] = 1.0, 0.0
    s = (rg.standard_normal(n) + 2.0 * t).astype(np.float32)
    if quant:  # heavy ties
        s = (np.round(s / quant) * quant).astype(np.float32)
    betas = [99.0, 199.0, 9.9]
    S, T = torch.from_numpy(s).cuda(), torch.from_numpy(t).cuda()
    mc, th = metrics.minc(S, T, betas)
    ref_mc, ref_th = orc.minc_reference(s, t, betas)
    assert abs(mc.item() - float(ref_mc)) <= 1e-6 * max(1.0, abs(float(ref_mc)))
    for b in betas:  # with ties in the cost the oracle (numpy argmin) and the kernel both take the first occurrence
        assert th[b].item() == np.float32(ref_th[b])
    mce, the = metrics.minc(S, T, betas, reference_semantics=False)
    ref_e, ref_the = orc.minc_exact(s, t, betas)
    assert abs(mce.item() - float(ref_e)) <= 1e-6
    for b in betas:
        assert the[b].item() == np.float32(ref_the[b])
    assert abs(metrics.eer(S, T) - orc.eer(s, t)) <= 1e-6
    # determinism
    mc2, th2 = metrics.minc(S, T, betas)
    assert mc2.item() == mc.item() and all(th2[b].item() == th[b].item() for b in betas)


def test_sweep_edge_cases(hip_lib):
    from neuralplda_amd import _lib, metrics, ops
    s = torch.tensor([0.5, 0.5, 0.5, 0.5], device="cuda")
    t = torch.tensor([1.0, 0.0, 1.0, 0.0], device="cuda")
    mc, th = metrics.minc(s, t, [99.0])
    ref_mc, ref_th = orc.minc_reference(s.cpu().numpy(), t.cpu().numpy(), [99.0])
    assert abs(mc.item() - float(ref_mc)) < 1e-6 and th[99.0].item() == 0.5
    # all non-targets: the reference raises (torch.min of an empty tensor); here the result is NaN
    mc, _ = metrics.minc(s, torch.zeros(4, device="cuda"), [99.0])
    assert np.isnan(mc.item())
    with pytest.raises(_lib.NpldaHipError):
        ops.detcost_sweep(s, t, [1.0] * 9)
    with pytest.raises(ValueError):
        ops.detcost_sweep(s, t[:3], [1.0])
