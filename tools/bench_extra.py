#!/usr/bin/env python3
"""Secondary measurements quoted in DESIGN.md (not the driver's bench line): Regime B indexed scoring,
embedding, a full training step at B = 4096, the AS-norm pipeline of BASELINE cfg3 on one GPU, and the
GaussianBackend scorer.  All inputs resident in HBM, HIP-event timing on torch's current stream."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from neuralplda_amd import models, ops  # noqa: E402


def timeit(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


class NC:
    def __init__(self, D):
        self.xvector_dim, self.layer1_LDA_dim, self.layer2_PLDA_spkfactor_dim = 512, D, D
        self.beta, self.alpha, self.device, self.loss = [99.0, 199.0], 15.0, "cuda", "SoftCdet"


def main():
    out = {}
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(7)
    for D in (150, 170):
        torch.manual_seed(D)
        m = models.NeuralPlda(NC(D)).to(dev)
        packed = ops.pack_params(*[p.detach() for p in m._params()])
        # Regime B: 100 k-utterance table, 1 M index pairs
        N, B = 100_000, 1 << 20
        X = torch.randn(N, 512, device=dev, generator=gen)
        i1 = torch.randint(0, N, (B,), device=dev, generator=gen)
        i2 = torch.randint(0, N, (B,), device=dev, generator=gen)
        ms_e = timeit(lambda: ops.embed(X, packed))
        z, q = ops.embed(X, packed)
        ms_i = timeit(lambda: ops.score_indexed(z, q, i1, i2, packed))
        bytes_pair = 2 * 4 * D + 2 * 4 + 2 * 8 + 4
        out[f"D{D}"] = {
            "embed_utts_per_s": N / ms_e * 1e3, "embed_ms_100k": ms_e,
            "indexed_pairs_per_s": B / ms_i * 1e3, "indexed_ms_1M": ms_i,
            "indexed_algorithmic_TBps": B * bytes_pair / ms_i / 1e9,
            "gather_rows_per_s": B / timeit(lambda: ops.gather_rows(X, i1)) * 1e3,
        }
        # the same call on a table larger than the 256 MB memory-side cache: 1.2 M utterances (VoxCeleb scale), rows come
        # from HBM in ldz * 4-byte pieces at random
        Nb = 1_200_000
        zb = torch.randn(Nb, z.shape[1], device=dev, generator=gen)
        qb = torch.randn(Nb, device=dev, generator=gen)
        j1 = torch.randint(0, Nb, (B,), device=dev, generator=gen)
        j2 = torch.randint(0, Nb, (B,), device=dev, generator=gen)
        ms_b = timeit(lambda: ops.score_indexed(zb, qb, j1, j2, packed))
        out[f"D{D}"].update({"indexed_big_table_MB": zb.numel() * 4 / 1e6, "indexed_big_pairs_per_s": B / ms_b * 1e3,
                             "indexed_big_algorithmic_TBps": B * bytes_pair / ms_b / 1e9,
                             "indexed_big_row_bytes_TBps": B * 2 * z.shape[1] * 4 / ms_b / 1e9})
        del zb, qb
        # training step, B = 4096 (BASELINE cfg2): forward + SoftCdet + backward + Adam
        Bt = 4096
        x1 = torch.randn(Bt, 512, device=dev, generator=gen)
        x2 = torch.randn(Bt, 512, device=dev, generator=gen)
        t = (torch.rand(Bt, device=dev, generator=gen) < 0.1).float()
        opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=1e-5)

        def step():
            opt.zero_grad()
            o = m(x1, x2)
            L = m.loss(o, t)
            L.backward()
            opt.step()

        ms_t = timeit(step, reps=20)
        out[f"D{D}"].update({"train_step_eager_autograd_torchAdam_ms_B4096": ms_t,
                             "train_eager_pairs_per_s": Bt / ms_t * 1e3})
        from neuralplda_amd import train as ntrain
        torch.manual_seed(D)
        mf = models.NeuralPlda(NC(D)).to(dev)
        fs = ntrain.FusedTrainStep(mf, 1e-4, batch_size=Bt, graph=False)
        ms_f = timeit(lambda: fs(x1, x2, t), reps=50)
        out[f"D{D}"].update({"train_step_fused_ms_B4096": ms_f, "train_fused_pairs_per_s": Bt / ms_f * 1e3})
        fwd_only = timeit(lambda: ops.forward_train(x1, x2, packed), reps=20)
        out[f"D{D}"]["forward_train_ms_B4096"] = fwd_only
        s, saved = ops.forward_train(x1, x2, packed)
        g = torch.randn(Bt, device=dev, generator=gen) / Bt
        out[f"D{D}"]["backward_ms_B4096"] = timeit(lambda: ops.backward(saved, g, packed, m.P_sqrt.detach()), reps=20)
        del X, z, q
    # AS-norm, BASELINE cfg3 on ONE GPU: R = 22 000 rows x M = 10 000 cohort, 2 M trials, top-500
    D = 170
    torch.manual_seed(3)
    m = models.NeuralPlda(NC(D)).to(dev)
    packed = ops.pack_params(*[p.detach() for p in m._params()])
    R, M, T = 22_000, 10_000, 2_000_000
    zr, qr = ops.embed(torch.randn(R, 512, device=dev, generator=gen), packed)
    zc, qc = ops.embed(torch.randn(M, 512, device=dev, generator=gen), packed)
    ms_c = timeit(lambda: ops.cohort_stats(zr, qr, zc, qc, packed, topn=500), reps=5, warm=2)
    stats = ops.cohort_stats(zr, qr, zc, qc, packed, topn=500)
    raw = torch.randn(T, device=dev, generator=gen).double()
    ie = torch.randint(0, 2000, (T,), device=dev, generator=gen)
    it = torch.randint(2000, R, (T,), device=dev, generator=gen)
    ms_a = timeit(lambda: ops.asnorm_apply(raw, ie, it, stats))
    out["asnorm_cfg3_1gpu"] = {"cohort_stats_ms": ms_c, "cohort_scores_per_s": R * M / ms_c * 1e3,
                               "cohort_gemm_TFLOPs": 2.0 * R * M * 176 / ms_c / 1e9,
                               "apply_ms_2M": ms_a, "apply_trials_per_s": T / ms_a * 1e3,
                               "apply_algorithmic_GBps": T * 100 / ms_a / 1e6}
    # GaussianBackend scorer, 2 D1 = 340
    gb = models.GaussianBackend(NC(170)).to(dev)
    A = torch.randn(340, 340, generator=torch.Generator().manual_seed(1))
    gb.paired_cov_inv_target = A @ A.T / 340 + torch.eye(340)
    gb.paired_cov_inv_nontarget = A.T @ A / 340 + 0.5 * torch.eye(340)
    gpk = ops.gb_pack(gb.centering_and_LDA.weight.detach(), gb.centering_and_LDA.bias.detach(),
                      *[t.to(dev) for t in (gb.paired_mean_target, gb.paired_cov_inv_target,
                                            gb.paired_mean_nontarget, gb.paired_cov_inv_nontarget)])
    Bg = 1 << 19
    x1 = torch.randn(Bg, 512, device=dev, generator=gen)
    x2 = torch.randn(Bg, 512, device=dev, generator=gen)
    ms_g = timeit(lambda: ops._gb_call(x1, x2, gpk, True, False))
    xs1, xs2 = x1[:2048].contiguous(), x2[:2048].contiguous()
    ms_gs = timeit(lambda: ops._gb_call(xs1, xs2, gpk, True, True), reps=50)
    xm1, xm2 = x1[:10240].contiguous(), x2[:10240].contiguous()
    ms_gm = timeit(lambda: ops._gb_call(xm1, xm2, gpk, True, False), reps=50)
    out["gaussian_backend_D170"] = {"pairs_per_s": Bg / ms_g * 1e3, "ms_512k": ms_g, "ms_2048_with_paired": ms_gs,
                                    "ms_10240": ms_gm,
                                    "TFLOPs_padded": Bg * (2 * 2 * 512 * 176 + 2 * 4 * 176 * 176) / ms_g / 1e9}
    # weighted moments of paired rows (n = 340): GB statistics pass (two classes) and DPlda gradient (one weight vector)
    paired = ops._gb_call(x1, x2, gpk, False, True)[1]
    t = (torch.rand(Bg, device=dev, generator=gen) < 0.1).float()
    ms_m2 = timeit(lambda: ops.weighted_moments(paired, t, 1 - t), reps=5, warm=2)
    small = paired[:2048].contiguous()
    gsm = torch.randn(2048, device=dev, generator=gen)
    ms_m1 = timeit(lambda: ops.weighted_moments(small, gsm), reps=20)
    out["weighted_moments_n340"] = {"two_class_ms_512k_rows": ms_m2, "rows_per_s": Bg / ms_m2 * 1e3,
                                    "TFLOPs_upper_triangle": 2 * 2.0 * Bg * 340 * 341 / 2 / ms_m2 / 1e9,
                                    "one_class_ms_2048_rows": ms_m1}
    # DPlda: scoring at D = 170 and one recipe step (LDA frozen, Adam on the linear unit) at B = 2048
    dp = models.DPlda(NC(170)).to(dev)
    for prm in dp.centering_and_LDA.parameters():
        prm.requires_grad = False
    with torch.no_grad():
        ms_d = timeit(lambda: dp(x1, x2))
    opt = torch.optim.Adam([p for p in dp.parameters() if p.requires_grad], lr=1e-4, weight_decay=1e-5)
    xa, xb, tt = x1[:2048].contiguous(), x2[:2048].contiguous(), t[:2048].contiguous()

    def dstep():
        opt.zero_grad()
        L = dp.loss(dp(xa, xb), tt)
        L.backward()
        opt.step()
    ms_ds = timeit(dstep, reps=20)
    from neuralplda_amd import train as _train
    fstep = _train.FusedDPldaStep(dp, 1e-4, weight_decay=1e-5, batch_size=2048, graph=True)
    ms_df = timeit(lambda: fstep(xa, xb, tt), reps=50, warm=5)
    out["dplda_D170"] = {"score_pairs_per_s": Bg / ms_d * 1e3, "score_ms_512k": ms_d,
                         "train_step_eager_autograd_torchAdam_ms_B2048": ms_ds, "train_step_fused_ms_B2048": ms_df}
    # validation metrics: minc (reference semantics) / exact min-DCF + EER over N scores
    for N in (1 << 20, 10_000_000):
        sc = torch.randn(N, device=dev, generator=gen)
        tg = (torch.rand(N, device=dev, generator=gen) < 0.05).float()
        sc = sc + 2 * tg
        ms_r = timeit(lambda: ops.detcost_sweep(sc, tg, [99.0, 199.0]), reps=5, warm=2)
        ms_e = timeit(lambda: ops.detcost_sweep(sc, tg, [99.0, 199.0], exact=True, want_eer=True), reps=5, warm=2)
        out[f"detcost_N{N}"] = {"minc_reference_semantics_ms": ms_r, "exact_mindcf_eer_ms": ms_e,
                                "scores_per_s": N / ms_r * 1e3}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
