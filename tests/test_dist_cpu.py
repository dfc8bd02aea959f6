"""world_size-2 gloo tests (CPU) of the multi-GPU layer: sharding arithmetic + collectives.  The compute
injected here is the oracle (the HIP kernels need a GPU); what is verified is that the partitioning of
SURVEY.md §8e reproduces the single-process result: trial-list sharding + score all-gather, AS-norm row
sharding + all-gather of row statistics, and data-parallel SoftCdet (global counts all-reduced before
the gradient, summed flat gradients)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import nplda_oracle as orc


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _params(rng, D0=64, D1=24, D2=20):
    k1, k2 = 1 / np.sqrt(D0), 1 / np.sqrt(D1)
    return orc.Params(rng.uniform(-k1, k1, (D1, D0)).astype(np.float32), rng.uniform(-k1, k1, D1).astype(np.float32),
                      rng.uniform(-k2, k2, (D2, D1)).astype(np.float32), rng.uniform(-k2, k2, D2).astype(np.float32),
                      rng.uniform(0, 1, D2).astype(np.float32), rng.uniform(0, 1, D2).astype(np.float32))


def _softcdet_sums(s, t, theta, alpha):
    """The additive fp64 sums of csrc/nplda_loss.hip, restated with the oracle's sigmoid."""
    out = [t.sum(), (1 - t).sum()]
    for th in theta:
        sg = orc._sigmoid(alpha * (th - s))
        d = sg * (1 - sg)
        out += [(sg * t).sum(), ((1 - sg) * (1 - t)).sum(), (d * t).sum(), (d * (1 - t)).sum()]
    return np.asarray(out, dtype=np.float64)


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from neuralplda_amd import dist as nd
    nd.init("gloo")
    try:
        assert nd.world() == (rank, world)
        rng = np.random.default_rng(0)  # same data on every rank
        p = _params(rng)
        # ---- 1. scoring: contiguous trial-list shards, no exchange, all-gather of scores --------------
        n = 1001  # not divisible by the world size
        x = rng.standard_normal((300, 64)).astype(np.float32)
        i1, i2 = rng.integers(0, 300, n), rng.integers(0, 300, n)
        full = orc.forward(x[i1], x[i2], p)
        lo, hi = nd.shard_bounds(n, world, rank)
        calls = []

        def score(lo_, hi_):
            calls.append((lo_, hi_))
            return torch.from_numpy(orc.forward(x[i1[lo_:hi_]], x[i2[lo_:hi_]], p))

        got = nd.sharded_apply(score, n)
        assert calls == [(lo, hi)] and got.shape == (n,)
        np.testing.assert_array_equal(got.numpy(), full)
        local = nd.sharded_apply(score, n, gather=False)
        np.testing.assert_array_equal(local.numpy(), full[lo:hi])
        # ---- 2. AS-norm: rows sharded, ONE all-gather of (mean, std, mean_top, std_top) -------------
        R, M = 37, 80
        zr = orc.extract_plda_embeddings(rng.standard_normal((R, 64)).astype(np.float32), p)
        zc = orc.extract_plda_embeddings(rng.standard_normal((M, 64)).astype(np.float32), p)
        stats_full = orc.cohort_stats(orc.cohort_scores(zr, zc, p), topn=10)
        rlo, rhi = nd.shard_bounds(R, world, rank)
        stats_local = torch.from_numpy(orc.cohort_stats(orc.cohort_scores(zr[rlo:rhi], zc, p), topn=10))
        stats = nd.all_gather_rows(stats_local, R)
        np.testing.assert_allclose(stats.numpy(), stats_full, rtol=1e-12)
        # ---- 3. data-parallel SoftCdet: global counts first, then SUMMED gradients ------------------
        B = 64
        x1 = rng.standard_normal((B, 64)).astype(np.float32)
        x2 = rng.standard_normal((B, 64)).astype(np.float32)
        t = (rng.random(B) < 0.2).astype(np.float64)
        t[:3] = 1
        t[B // 2:B // 2 + 2] = 0 if rank >= 0 else 1
        theta, beta, alpha = [-0.5, -0.3], [99.0, 199.0], 15.0
        s_full = orc.forward(x1, x2, p, np.float64)
        g_full, dth_full = orc.softcdet_grad(s_full, t, theta, beta, alpha)
        grads_full = orc.backward(x1, x2, g_full, p)
        L_full = orc.softcdet(s_full, t, theta, beta, alpha, np.float64)
        (sx1, sx2, st) = nd.shard_batch((torch.from_numpy(x1), torch.from_numpy(x2), torch.from_numpy(t)))
        sx1, sx2, st = sx1.numpy(), sx2.numpy(), st.numpy()
        s_loc = orc.forward(sx1, sx2, p, np.float64)
        sums = torch.from_numpy(_softcdet_sums(s_loc, st, theta, alpha))
        nd.allreduce_sum_(sums)
        sums = sums.numpy()
        np.testing.assert_allclose(sums, _softcdet_sums(s_full, t, theta, alpha), rtol=1e-12)
        L = np.mean([sums[2 + 4 * k] / sums[0] + beta[k] * sums[3 + 4 * k] / sums[1] for k in range(2)])
        assert abs(L - L_full) <= 1e-12 * abs(L_full)
        g_loc, _ = orc.softcdet_grad(s_loc, st, theta, beta, alpha, nt=sums[0], nn=sums[1])
        blo, bhi = nd.shard_bounds(B, world, rank)
        np.testing.assert_allclose(g_loc, g_full[blo:bhi], rtol=1e-12)
        gl = orc.backward(sx1, sx2, g_loc, p)
        flat = torch.from_numpy(np.concatenate([gl[k].ravel() for k in ("W1", "b1", "W2", "b2", "P_sqrt", "Q")]))
        nd.allreduce_sum_(flat)  # SUM, not mean
        ref = np.concatenate([grads_full[k].ravel() for k in ("W1", "b1", "W2", "b2", "P_sqrt", "Q")])
        np.testing.assert_allclose(flat.numpy(), ref, rtol=1e-9, atol=1e-14)
        dth = [(alpha * sums[4 + 4 * k] / sums[0] - beta[k] * alpha * sums[5 + 4 * k] / sums[1]) / 2 for k in range(2)]
        np.testing.assert_allclose(dth, dth_full, rtol=1e-10)
        # ---- 3b. the ONE-collective form (nplda_train_step_grad_f32 -> all-reduce -> nplda_train_step_apply_f32): the
        #          counts come from the global label vector (no collective before the backward), and the fp64 loss sums
        #          ride in the gradient's fp32 all-reduce as four 16-bit limbs each -------------------------------------------
        nt_g, nn_g = float(t.sum()), float(B - t.sum())
        g1, _ = orc.softcdet_grad(s_loc, st, theta, beta, alpha, nt=nt_g, nn=nn_g)
        np.testing.assert_allclose(g1, g_full[blo:bhi], rtol=1e-12)
        gl1 = orc.backward(sx1, sx2, g1, p)
        own = _softcdet_sums(s_loc, st, theta, alpha)

        def limbs(v):  # csrc/nplda_loss_tail.h: loss_limbs_of — four 16-bit fixed-point limbs, exact under fp32 addition
            sg, r = np.where(v < 0, -1.0, 1.0), np.abs(v)
            a = np.floor(r * 2.0 ** -8); r = r - a * 2.0 ** 8
            b = np.floor(r * 2.0 ** 8); r = r - b * 2.0 ** -8
            c = np.floor(r * 2.0 ** 24); r = r - c * 2.0 ** -24
            d = np.rint(r * 2.0 ** 40)
            return np.concatenate([sg * a, sg * b, sg * c, sg * d]).astype(np.float32)

        one = torch.from_numpy(np.concatenate([np.concatenate([gl1[k].ravel() for k in ("W1", "b1", "W2", "b2", "P_sqrt", "Q")])
                                               .astype(np.float32), limbs(own)]))
        nd.allreduce_sum_(one)  # the step's only exchange
        one = one.numpy()
        ng, ns_ = ref.size, own.size
        np.testing.assert_allclose(one[:ng], ref, rtol=2e-5, atol=1e-7)
        L4 = one[ng:].astype(np.float64).reshape(4, ns_)
        sums1 = L4[0] * 2.0 ** 8 + L4[1] * 2.0 ** -8 + L4[2] * 2.0 ** -24 + L4[3] * 2.0 ** -40
        np.testing.assert_allclose(sums1, _softcdet_sums(s_full, t, theta, alpha), rtol=0, atol=2.0 ** -39)
        L1 = np.mean([sums1[2 + 4 * k] / sums1[0] + beta[k] * sums1[3 + 4 * k] / sums1[1] for k in range(2)])
        assert abs(L1 - L_full) <= 1e-11 * abs(L_full) and sums1[0] == nt_g and sums1[1] == nn_g
        # ---- 4. make_data_parallel wires the two reductions into the module ----------------------------
        from neuralplda_amd import models

        class NC:
            xvector_dim, layer1_LDA_dim, layer2_PLDA_spkfactor_dim = 64, 24, 20
            beta, alpha, device, loss = [99.0, 199.0], 15.0, "cpu", "SoftCdet"

        torch.manual_seed(rank)  # deliberately different init per rank
        m = models.NeuralPlda(NC())
        nd.broadcast_parameters(m, src=0)
        w = m.centering_and_LDA.weight.detach().clone()
        ws = [torch.empty_like(w) for _ in range(world)]
        torch.distributed.all_gather(ws, w)
        assert all(torch.equal(ws[0], wi) for wi in ws)
        nd.make_data_parallel(m)
        v = torch.tensor([1.0 + rank, 2.0], dtype=torch.float64)
        assert m._reduce_sums(v).tolist() == [sum(1.0 + r for r in range(world)), 2.0 * world]
        f = torch.ones(5)
        assert m._reduce_flat(f).tolist() == [float(world)] * 5
        # ---- 5. data-parallel DPlda step: global loss sums, then the folded fp64 gradient of the linear unit summed
        #         through the hook make_data_parallel installs (what train.FusedDPldaStep / _DPldaScoreFn call) ----------
        from neuralplda_amd import ops
        D1 = 8
        Wl = rng.uniform(-0.2, 0.2, (D1, 64)).astype(np.float32)
        bl = rng.uniform(-0.1, 0.1, D1).astype(np.float32)
        wlr = (0.1 * rng.standard_normal((1, 2 * D1 * D1 + D1))).astype(np.float32)
        blr = np.asarray([0.05], np.float32)
        sd_full = orc.dplda_forward(x1, x2, Wl, bl, wlr, blr, np.float64)
        gd_full, _ = orc.softcdet_grad(sd_full, t, theta, beta, alpha)

        def paired(a, b):
            y1, _ = orc.normalize(a.astype(np.float64) @ Wl.T.astype(np.float64) + bl, np.float64)
            y2, _ = orc.normalize(b.astype(np.float64) @ Wl.T.astype(np.float64) + bl, np.float64)
            return y1, y2

        dw_full, db_full = orc.dplda_backward(*paired(x1, x2), gd_full)
        sd_loc = orc.dplda_forward(sx1, sx2, Wl, bl, wlr, blr, np.float64)
        dsums = torch.from_numpy(_softcdet_sums(sd_loc, st, theta, alpha))
        dsums = m._reduce_sums(dsums).numpy()
        gd_loc, _ = orc.softcdet_grad(sd_loc, st, theta, beta, alpha, nt=dsums[0], nn=dsums[1])
        y1l, y2l = paired(sx1, sx2)
        xp = np.concatenate([y1l, y2l], axis=1)
        cnt, sm, sq = orc.weighted_moments(xp, gd_loc)
        dw, db = ops.dplda_fold_grad(torch.tensor([cnt]), torch.from_numpy(sm)[None], torch.from_numpy(sq)[None], D1,
                                     reduce=m.__dict__["_reduce_sums64"])
        np.testing.assert_allclose(dw.numpy(), dw_full.astype(np.float32), rtol=2e-6, atol=1e-9)
        np.testing.assert_allclose(db.numpy(), db_full.astype(np.float32), rtol=2e-6, atol=1e-9)
        open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    finally:
        torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_world_size_2_gloo(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))


def test_shard_bounds_cover_everything():
    from neuralplda_amd import dist as nd
    for n in (0, 1, 7, 8, 1000, 1001):
        for w in (1, 2, 3, 8):
            spans = [nd.shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            chunk = (n + w - 1) // w
            assert all(hi - lo <= chunk for lo, hi in spans)
    assert nd.world() == (0, 1)
    t = torch.arange(4.0)
    assert nd.allreduce_sum_(t) is t and nd.sharded_apply(lambda lo, hi: torch.arange(lo, hi), 5).tolist() == [0, 1, 2, 3, 4]


def _worker_short_last(rank, world, port, tmp):
    """World size 4 with n = 9 units: chunks of ceil(9 / 4) = 3 -> shards [0, 3) [3, 6) [6, 9) and an EMPTY shard on the last
    rank, for all three partitionings of SURVEY 8(e): the trial list, the AS-norm rows, the training minibatch."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from neuralplda_amd import dist as nd
    nd.init("gloo")
    try:
        rng = np.random.default_rng(3)
        p = _params(rng)
        n = 9
        lo, hi = nd.shard_bounds(n, world, rank)
        assert (lo, hi) == [(0, 3), (3, 6), (6, 9), (9, 9)][rank]
        # ---- 1. scoring: the last rank scores nothing and still takes part in the all-gather ------------------------------
        x = rng.standard_normal((40, 64)).astype(np.float32)
        i1, i2 = rng.integers(0, 40, n), rng.integers(0, 40, n)
        full = orc.forward(x[i1], x[i2], p)

        def score(lo_, hi_):
            if hi_ == lo_:
                return torch.empty(0, dtype=torch.float32)
            return torch.from_numpy(orc.forward(x[i1[lo_:hi_]], x[i2[lo_:hi_]], p))

        got = nd.sharded_apply(score, n)
        np.testing.assert_array_equal(got.numpy(), full)
        assert nd.sharded_apply(score, n, gather=False).shape == (hi - lo,)
        # ---- 2. AS-norm: 9 rows, the last rank holds none; ONE all-gather of (9, 4) statistics, then trial shards ----------
        R, M = 9, 60
        zr = orc.extract_plda_embeddings(rng.standard_normal((R, 64)).astype(np.float32), p)
        zc = orc.extract_plda_embeddings(rng.standard_normal((M, 64)).astype(np.float32), p)
        stats_full = orc.cohort_stats(orc.cohort_scores(zr, zc, p), topn=10)
        if hi > lo:
            local = torch.from_numpy(orc.cohort_stats(orc.cohort_scores(zr[lo:hi], zc, p), topn=10))
        else:
            local = torch.empty((0, 4), dtype=torch.float64)
        stats = nd.all_gather_rows(local, R)
        np.testing.assert_allclose(stats.numpy(), stats_full, rtol=1e-12)
        # ---- 3. data-parallel SoftCdet on a 9-pair minibatch: the empty shard contributes zeros to both all-reduces --------
        B = 9
        x1 = rng.standard_normal((B, 64)).astype(np.float32)
        x2 = rng.standard_normal((B, 64)).astype(np.float32)
        t = np.asarray([1, 0, 0, 1, 0, 0, 0, 1, 0], dtype=np.float64)
        theta, beta, alpha = [-0.5, -0.3], [99.0, 199.0], 15.0
        s_full = orc.forward(x1, x2, p, np.float64)
        g_full, dth_full = orc.softcdet_grad(s_full, t, theta, beta, alpha)
        grads_full = orc.backward(x1, x2, g_full, p)
        sx1, sx2, st = (v.numpy() for v in nd.shard_batch((torch.from_numpy(x1), torch.from_numpy(x2), torch.from_numpy(t))))
        assert sx1.shape[0] == hi - lo
        keys = ("W1", "b1", "W2", "b2", "P_sqrt", "Q")
        if hi > lo:
            s_loc = orc.forward(sx1, sx2, p, np.float64)
            own = _softcdet_sums(s_loc, st, theta, alpha)
        else:
            own = np.zeros(2 + 4 * len(theta))
        sums = nd.allreduce_sum_(torch.from_numpy(own)).numpy()
        np.testing.assert_allclose(sums, _softcdet_sums(s_full, t, theta, alpha), rtol=1e-12)
        if hi > lo:
            g_loc, _ = orc.softcdet_grad(s_loc, st, theta, beta, alpha, nt=sums[0], nn=sums[1])
            np.testing.assert_allclose(g_loc, g_full[lo:hi], rtol=1e-12)
            gl = orc.backward(sx1, sx2, g_loc, p)
            flat = np.concatenate([gl[k].ravel() for k in keys])
        else:
            flat = np.zeros(sum(grads_full[k].size for k in keys))
        flat = nd.allreduce_sum_(torch.from_numpy(flat)).numpy()
        np.testing.assert_allclose(flat, np.concatenate([grads_full[k].ravel() for k in keys]), rtol=1e-9, atol=1e-14)
        open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    finally:
        torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_world_size_4_gloo_with_an_empty_last_shard(tmp_path):
    world = 4
    port = _free_port()
    mp.spawn(_worker_short_last, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))
