"""The RCCL ("nccl") code path of neuralplda_amd.dist on real hardware with a one-rank group (the GPU box has one GPU;
world_size 2 runs on gloo in tests/test_dist_cpu.py): process-group init with device_id, sharded scoring + all-gather,
AS-norm row statistics gathered across the group, and a data-parallel training step (all-reduce of the loss sums and of the
flat gradient) must reproduce the single-process results bit for bit."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class NC:
    xvector_dim, layer1_LDA_dim, layer2_PLDA_spkfactor_dim = 512, 150, 150
    beta, alpha, device, loss = [99.0, 199.0], 15.0, "cuda", "SoftCdet"


@pytest.fixture()
def one_rank_group():
    import torch.distributed as td
    from neuralplda_amd import dist as ndist
    if td.is_initialized():
        pytest.skip("a process group already exists")
    env = {"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533"}
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    ndist.init("nccl")
    yield ndist
    td.destroy_process_group()
    for k, v in saved.items():  # (bench.py's launcher checks in later tests read these)
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def test_rccl_single_rank_paths(hip_lib, one_rank_group):
    ndist = one_rank_group
    from neuralplda_amd import adaptive_score_normalization as asn, models, train
    import torch.distributed as td
    assert ndist.world() == (0, 1) and td.get_backend() == "nccl"
    torch.manual_seed(0)
    m = models.NeuralPlda(NC()).cuda()
    gen = torch.Generator(device="cuda").manual_seed(1)
    x1 = torch.randn(5000, 512, device="cuda", generator=gen)
    x2 = torch.randn(5000, 512, device="cuda", generator=gen)
    with torch.no_grad():
        ref = m(x1, x2)
        got = ndist.sharded_apply(lambda lo, hi: m(x1[lo:hi], x2[lo:hi]), 5000)
    assert torch.equal(got, ref)
    # collectives on device tensors through RCCL
    t = torch.arange(10, dtype=torch.float64, device="cuda")
    ndist.allreduce_sum_(t)
    assert torch.equal(t, torch.arange(10, dtype=torch.float64, device="cuda"))
    rows = torch.randn(7, 4, dtype=torch.float64, device="cuda", generator=gen)
    assert torch.equal(ndist.all_gather_rows(rows, 7), rows)
    # AS-norm with the group argument == without
    xr = torch.randn(64, 512, device="cuda", generator=gen)
    xc = torch.randn(900, 512, device="cuda", generator=gen)
    ie, it = np.arange(200) % 20, 20 + np.arange(200) % 44
    with torch.no_grad():
        raw = m(xr[torch.from_numpy(ie).cuda()], xr[torch.from_numpy(it).cuda()]).double().cpu().numpy()
    a = asn.asnorm_scores(m, xr, xc, raw, ie, it, topN=100)
    b = asn.asnorm_scores(m, xr, xc, raw, ie, it, topN=100, group=td.group.WORLD)
    assert torch.equal(a, b)
    # data-parallel wrapper: loss and gradients equal the plain model's
    tgt = (torch.rand(5000, device="cuda", generator=gen) < 0.2).float()
    m2 = models.NeuralPlda(NC()).cuda()
    m2.load_state_dict(m.state_dict())
    ndist.make_data_parallel(m2)
    L1 = m.loss(m(x1, x2), tgt)
    L1.backward()
    xs1, xs2, ts = ndist.shard_batch((x1, x2, tgt))
    L2 = m2.loss(m2(xs1, xs2), ts)
    L2.backward()
    assert L1.item() == L2.item()
    for (k, p1), (_, p2) in zip(m.named_parameters(), m2.named_parameters()):
        assert (p1.grad is None) == (p2.grad is None), k
        if p1.grad is not None:
            assert torch.equal(p1.grad, p2.grad), k
    step = train.FusedTrainStep(m2, 1e-4, batch_size=5000, graph=False)
    assert np.isfinite(step(xs1, xs2, ts).item())


def test_data_parallel_fused_steps_replay_from_a_graph(hip_lib, one_rank_group):
    """The data-parallel fused steps carry their two collectives INSIDE the captured HIP graph (RCCL collectives are
    stream operations): with a one-rank RCCL group the graph-replayed DP step must follow the plain fused step bit for
    bit — NeuralPlda (loss sums + flat gradient) and DPlda (loss sums + folded fp64 unit gradient)."""
    ndist = one_rank_group
    from neuralplda_amd import models, train
    gen = torch.Generator(device="cuda").manual_seed(3)
    B = 2048
    batches = [(torch.randn(B, 512, device="cuda", generator=gen), torch.randn(B, 512, device="cuda", generator=gen),
                (torch.rand(B, device="cuda", generator=gen) < 0.2).float()) for _ in range(4)]
    torch.manual_seed(0)
    m_plain = models.NeuralPlda(NC()).cuda()
    m_dp = models.NeuralPlda(NC()).cuda()
    m_dp.load_state_dict(m_plain.state_dict())
    ndist.make_data_parallel(m_dp)
    s_plain = train.FusedTrainStep(m_plain, 1e-3, batch_size=B, graph=True)
    s_dp = train.FusedTrainStep(m_dp, 1e-3, batch_size=B, graph=True)
    assert s_dp.reduce_sums is not None and s_dp.reduce_flat is not None and s_dp.use_graph
    for x1, x2, t in batches:
        a, b = s_plain(x1, x2, t), s_dp(x1, x2, t)
        assert abs(float(a) - float(b)) <= 1e-6 * abs(float(a))
    assert s_dp._graph is not None
    for (k, p1), (_, p2) in zip(m_plain.state_dict().items(), m_dp.state_dict().items()):
        assert torch.allclose(p1, p2, rtol=1e-6, atol=1e-8), k

    # the head's end-to-end step (bf16 rows in, dL/dx out) in its one-collective data-parallel form, graph-replayed
    h_plain = models.NeuralPlda(NC()).cuda()
    h_dp = models.NeuralPlda(NC()).cuda()
    h_dp.load_state_dict(h_plain.state_dict())
    ndist.make_data_parallel(h_dp)
    hs_plain = train.HeadStepWithInputGrads(h_plain, 1e-3, batch_size=B, graph=True)
    hs_dp = train.HeadStepWithInputGrads(h_dp, 1e-3, batch_size=B, graph=True)
    assert hs_dp._dp_head
    for x1, x2, t in batches:
        (a, da1, da2), (b, db1, db2) = hs_plain(x1.bfloat16(), x2.bfloat16(), t), hs_dp(x1.bfloat16(), x2.bfloat16(), t)
        assert abs(float(a) - float(b)) <= 1e-6 * abs(float(a)) and torch.equal(da1, db1) and torch.equal(da2, db2)
    assert hs_dp._g is not None
    for (k, p1), (_, p2) in zip(h_plain.state_dict().items(), h_dp.state_dict().items()):
        assert torch.allclose(p1, p2, rtol=1e-6, atol=1e-8), k

    class NCD:
        xvector_dim, layer1_LDA_dim, layer2_PLDA_spkfactor_dim = 512, 170, 170
        beta, alpha, device, loss = [99.0, 199.0], 15.0, "cuda", "SoftCdet"

    torch.manual_seed(1)
    d_plain = models.DPlda(NCD()).cuda()
    d_dp = models.DPlda(NCD()).cuda()
    d_dp.load_state_dict(d_plain.state_dict())
    for d in (d_plain, d_dp):
        d.centering_and_LDA.weight.requires_grad = False
        d.centering_and_LDA.bias.requires_grad = False
    ndist.make_data_parallel(d_dp)
    f_plain = train.FusedDPldaStep(d_plain, 1e-3, batch_size=B, graph=True)
    f_dp = train.FusedDPldaStep(d_dp, 1e-3, batch_size=B, graph=True)
    assert f_dp.reduce_sums is not None and f_dp.reduce_flat is not None
    for x1, x2, t in batches:
        a, b = f_plain(x1, x2, t), f_dp(x1, x2, t)
        assert abs(float(a) - float(b)) <= 1e-6 * abs(float(a))
    for (k, p1), (_, p2) in zip(d_plain.state_dict().items(), d_dp.state_dict().items()):
        assert torch.allclose(p1, p2, rtol=1e-6, atol=1e-8), k
    # autograd path of the DP DPlda model: same gradients as the plain one
    x1, x2, t = batches[0]
    for d in (d_plain, d_dp):
        d.zero_grad()
        d.loss(d(x1, x2), t).backward()
    assert torch.allclose(d_plain.logistic_regres.weight.grad, d_dp.logistic_regres.weight.grad, rtol=1e-6, atol=1e-9)


# ---- two ranks on ONE device (gloo carries the collectives; the compute is the HIP library's) ---------------------------------
def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _two_rank_train_worker(rank, world, port, tmp):
    """train.train() in its data-parallel form on `world` gloo ranks sharing cuda:0: 129 trials at batch size 64 leave a
    tail batch of ONE pair, i.e. an EMPTY shard on rank 1 — every rank must still issue the step's single all-reduce and
    count the step.  (Binary cross-entropy: SoftCdet divides by the batch's target AND non-target counts, one of which is
    zero in a one-pair batch — NaN in the reference too.)  Rank 1's generator is deliberately out of step (the epoch's seed is broadcast from rank 0).  The
    parameters after the epoch must be the same on both ranks and follow the single-process run on the global batches."""
    import contextlib
    import io
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    import torch.distributed as td
    from neuralplda_amd import dist as ndist, models, sv_trials_loaders as svl, train
    torch.cuda.set_device(0)
    ndist.init("gloo")
    try:
        rng = np.random.default_rng(21)
        n_utt, n_trials, B = 200, 129, 64
        ids = [f"u{u:04d}" for u in range(n_utt)]
        xv = rng.standard_normal((n_utt, 512)).astype(np.float32)
        mega = {u: xv[i] for i, u in enumerate(ids)}
        num_to_id = dict(enumerate(ids))
        a, b = rng.integers(0, n_utt, n_trials), rng.integers(0, n_utt, n_trials)
        lab = (rng.random(n_trials) < 0.3).astype(np.float32)
        ds = svl.TrialIndexDataset(torch.from_numpy(a), torch.from_numpy(b), torch.from_numpy(lab))
        loader = svl._loader(ds, B)

        class Conf:
            xvector_dim, layer1_LDA_dim, layer2_PLDA_spkfactor_dim = 512, 150, 150
            beta, alpha, device, loss, log_interval = [99.0, 199.0], 15.0, "cuda", "crossentropy", 1

        def fresh():
            torch.manual_seed(0)
            return models.NeuralPlda(Conf()).cuda()

        def run(m, seed):
            step = train.FusedTrainStep(m, 1e-3, weight_decay=1e-5, batch_size=B, graph=False)
            torch.manual_seed(seed)
            out = io.StringIO()
            with contextlib.redirect_stdout(out):
                train.train(Conf, m, torch.device("cuda"), loader, mega, num_to_id, None, 1, step_fn=step)
            return step, out.getvalue()

        m_dp = ndist.make_data_parallel(fresh())
        step, log_dp = run(m_dp, 11 if rank == 0 else 12345)  # rank 1's own seed would give another permutation
        assert step._dp_call and int(step.step_count[0].item()) == 3
        sd = {k: v.detach().cpu() for k, v in m_dp.state_dict().items()}
        others = [None] * world
        td.all_gather_object(others, sd)
        for k in sd:
            assert torch.equal(others[0][k], others[1][k]), k  # the ranks stay in step, bit for bit
        if rank == 0:
            m_one = fresh()
            _, log_one = run(m_one, 11)
            l_one = [float(ln.rsplit(" ", 1)[-1]) for ln in log_one.strip().splitlines()]
            l_dp = [float(ln.rsplit(" ", 1)[-1]) for ln in log_dp.strip().splitlines()]
            assert len(l_one) == len(l_dp) == 3 and np.allclose(l_one, l_dp, rtol=2e-5), (l_one, l_dp)
            # Adam divides by sqrt(v) + 1e-8: where a gradient element is itself rounding noise, the two summation orders
            # (one batch / two shards + all-reduce) can step it differently by up to lr — so: nearly all elements agree to
            # fp32 rounding, and none differs by more than the three steps' worth of lr
            for (k, p1), (_, p2) in zip(m_one.state_dict().items(), m_dp.state_dict().items()):
                d = (p1 - p2).abs()
                tol = 2e-6 + 2e-5 * p1.abs()
                assert float((d <= tol).float().mean()) >= 0.995 and float(d.max()) <= 3.1e-3, (k, float(d.max()))
        open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    except Exception:
        import traceback
        open(os.path.join(tmp, f"err{rank}"), "w").write(traceback.format_exc())
    finally:
        try:
            td.barrier()
        except Exception:
            pass
        td.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_gloo_ranks_on_one_device_train_with_an_empty_tail_shard(hip_lib, tmp_path):
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_two_rank_train_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    errs = {r: (tmp_path / f"err{r}").read_text() for r in range(world) if (tmp_path / f"err{r}").exists()}
    assert not errs, errs
    assert all((tmp_path / f"ok{r}").exists() for r in range(world))
