#!/usr/bin/env python3
"""Round-2 golden vectors, again produced by RUNNING THE REFERENCE ITSELF (build container only; plain arrays out).

    python tests/golden/make_golden_r2.py

Kept separate from make_golden.py so that the RNG streams of the round-1 fixtures do not move.

G11  gradients w.r.t. the INPUTS by the reference's autograd (utils/models.py:366-382 is differentiable in x — the E2E
     model, :251-268, back-propagates into its extractor through exactly these lines):
       * NeuralPlda small (64->24->20): x1.grad, x2.grad under SoftCdet and crossentropy (fp32 + fp64 re-evaluation);
         extract_plda_embeddings alone with a random upstream gradient (x.grad and the four layer gradients);
         forward_from_plda_embeddings alone (z1.grad, z2.grad, P_sqrt.grad, Q.grad);
       * NeuralPlda Kaldi-initialised 512->170->170 on the G2 inputs: x1.grad, x2.grad;
       * DPlda small (64->24) with the LDA trainable: x1.grad, x2.grad, dW1, db1 next to d wlr, d blr (fp32 + fp64).
G12  the reference's OWN driver functions train() and validate() (xvector_NeuralPlda_pytorch.py:30-83, imported — the
     module is guarded by __main__) on a tiny seeded set: validate(update_thresholds=True) -> thresholds, one epoch of
     train() with torch.optim.Adam(lr, weight_decay=1e-5) (:139) -> per-batch losses and the final state dict, then
     validate() again -> (minc, thresholds, cdet, softcdet as logged).
G7b  the GaussianBackend statistics of g7_gb_kaldi170.npz stored as arrays (round 1 stored only their seed).
"""
import copy
import io
import os
import sys
import tempfile
import types
from contextlib import redirect_stdout

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REF)
sys.path.insert(1, ROOT)
sys.path.insert(2, HERE)
sys.modules.setdefault("kaldi_io", types.ModuleType("kaldi_io"))

from utils import models as refm  # noqa: E402
from utils import sv_trials_loaders as refl  # noqa: E402
import xvector_NeuralPlda_pytorch as refdrv  # noqa: E402  (train / validate; main guarded by __main__)

from make_golden import NC, kaldi_init_model, params_of, save  # noqa: E402

torch.set_num_threads(4)


def g11():
    rng = np.random.default_rng(1111)
    ncs = NC(D0=64, D1=24, D2=20)
    torch.manual_seed(11)
    ms = refm.NeuralPlda(ncs)
    with torch.no_grad():
        ms.threshold[99.0].fill_(-0.5)
        ms.threshold[199.0].fill_(-0.3)
        ms.threshold_Xent.fill_(0.25)
    ps = params_of(ms)
    B = 200
    x1 = rng.standard_normal((B, 64)).astype(np.float32)
    x2 = rng.standard_normal((B, 64)).astype(np.float32)
    x1[5] = 0.0
    t = (rng.random(B) < 0.15).astype(np.float32)
    out = {}
    msd = copy.deepcopy(ms).double()
    msd.threshold = {99.0: msd.Th99, 199.0: msd.Th199}
    msd.alpha = torch.tensor(15.0, dtype=torch.float64)
    for tag, mdl, cast in (("f32", ms, lambda a: torch.from_numpy(a)), ("f64", msd, lambda a: torch.from_numpy(a).double())):
        for lossname in ("SoftCdet", "crossentropy"):
            mdl.lossfn = lossname
            mdl.zero_grad()
            a, b = cast(x1).requires_grad_(True), cast(x2).requires_grad_(True)
            L = mdl.loss(mdl(a, b), cast(t))
            L.backward()
            out[f"{lossname}_{tag}_L"] = L.detach().numpy()
            out[f"{lossname}_{tag}_dx1"] = a.grad.numpy().copy()
            out[f"{lossname}_{tag}_dx2"] = b.grad.numpy().copy()
            out[f"{lossname}_{tag}_dW1"] = mdl.centering_and_LDA.weight.grad.numpy().copy()
        # extract_plda_embeddings alone
        Gz = rng.standard_normal((B, 20)).astype(np.float32) if tag == "f32" else Gz  # noqa: F821 (same upstream for both)
        mdl.zero_grad()
        a = cast(x1).requires_grad_(True)
        z = mdl.extract_plda_embeddings(a)
        (z * cast(Gz)).sum().backward()
        out[f"embed_{tag}_z"] = z.detach().numpy()
        out[f"embed_{tag}_dx"] = a.grad.numpy().copy()
        out[f"embed_{tag}_dW1"] = mdl.centering_and_LDA.weight.grad.numpy().copy()
        out[f"embed_{tag}_db1"] = mdl.centering_and_LDA.bias.grad.numpy().copy()
        out[f"embed_{tag}_dW2"] = mdl.centering_and_wccn_plda.weight.grad.numpy().copy()
        out[f"embed_{tag}_db2"] = mdl.centering_and_wccn_plda.bias.grad.numpy().copy()
        # forward_from_plda_embeddings alone
        z1v = rng.standard_normal((B, 20)).astype(np.float32) if tag == "f32" else z1v  # noqa: F821
        z2v = rng.standard_normal((B, 20)).astype(np.float32) if tag == "f32" else z2v  # noqa: F821
        gs = rng.standard_normal(B).astype(np.float32) if tag == "f32" else gs  # noqa: F821
        mdl.zero_grad()
        za, zb = cast(z1v).requires_grad_(True), cast(z2v).requires_grad_(True)
        s = mdl.forward_from_plda_embeddings(za, zb)
        (s * cast(gs)).sum().backward()
        out[f"embscore_{tag}_s"] = s.detach().numpy()
        out[f"embscore_{tag}_dz1"] = za.grad.numpy().copy()
        out[f"embscore_{tag}_dz2"] = zb.grad.numpy().copy()
        out[f"embscore_{tag}_dP_sqrt"] = mdl.P_sqrt.grad.numpy().copy()
        out[f"embscore_{tag}_dQ"] = mdl.Q.grad.numpy().copy()
    save("g11_input_grads_small.npz", **ps, x1=x1, x2=x2, t=t, theta=np.asarray([-0.5, -0.3]), theta_xent=0.25,
         beta=np.asarray([99.0, 199.0]), alpha=15.0, Gz=Gz, z1=z1v, z2=z2v, gs=gs, **out)

    # Kaldi-initialised 170-d model on the G2 inputs
    g2 = np.load(os.path.join(HERE, "g2_forward_kaldi170.npz"))
    g3 = np.load(os.path.join(HERE, "g3_loss_kaldi170.npz"))
    torch.manual_seed(1)
    mk = kaldi_init_model(NC())
    with torch.no_grad():
        mk.threshold[99.0].fill_(-0.9)
        mk.threshold[199.0].fill_(-0.8)
    mk.lossfn = "SoftCdet"
    a = torch.from_numpy(g2["x1"]).requires_grad_(True)
    b = torch.from_numpy(g2["x2"]).requires_grad_(True)
    L = mk.loss(mk(a, b), torch.from_numpy(g3["t"]))
    L.backward()
    mkd = copy.deepcopy(mk).double()
    mkd.threshold = {99.0: mkd.Th99, 199.0: mkd.Th199}
    mkd.alpha = torch.tensor(15.0, dtype=torch.float64)
    mkd.zero_grad()
    ad = torch.from_numpy(g2["x1"]).double().requires_grad_(True)
    bd = torch.from_numpy(g2["x2"]).double().requires_grad_(True)
    Ld = mkd.loss(mkd(ad, bd), torch.from_numpy(g3["t"]).double())
    Ld.backward()
    save("g11_input_grads_kaldi170.npz", L=L.detach().numpy(), dx1=a.grad.numpy(), dx2=b.grad.numpy(),
         L64=Ld.detach().numpy(), dx1_64=ad.grad.numpy(), dx2_64=bd.grad.numpy())

    # DPlda with the LDA trainable
    ncd = NC(D0=64, D1=24, D2=24)
    torch.manual_seed(12)
    dp = refm.DPlda(ncd)
    with torch.no_grad():
        dp.threshold[99.0].fill_(0.2)
        dp.threshold[199.0].fill_(0.35)
    xd1 = rng.standard_normal((60, 64)).astype(np.float32)
    xd2 = rng.standard_normal((60, 64)).astype(np.float32)
    td = (rng.random(60) < 0.3).astype(np.float32)
    outd = {}
    for tag, mdl, cast in (("f32", dp, lambda a: torch.from_numpy(a)),
                           ("f64", copy.deepcopy(dp).double(), lambda a: torch.from_numpy(a).double())):
        if tag == "f64":
            mdl.threshold = {99.0: mdl.Th99, 199.0: mdl.Th199}
            mdl.alpha = torch.tensor(15.0, dtype=torch.float64)
        for lossname in ("SoftCdet", "crossentropy"):
            mdl.lossfn = lossname
            mdl.zero_grad()
            a, b = cast(xd1).requires_grad_(True), cast(xd2).requires_grad_(True)
            L = mdl.loss(mdl(a, b), cast(td))
            L.backward()
            outd[f"{lossname}_{tag}_L"] = L.detach().numpy()
            outd[f"{lossname}_{tag}_dx1"] = a.grad.numpy().copy()
            outd[f"{lossname}_{tag}_dx2"] = b.grad.numpy().copy()
            outd[f"{lossname}_{tag}_dW1"] = mdl.centering_and_LDA.weight.grad.numpy().copy()
            outd[f"{lossname}_{tag}_db1"] = mdl.centering_and_LDA.bias.grad.numpy().copy()
            outd[f"{lossname}_{tag}_dwlr"] = mdl.logistic_regres.weight.grad.numpy().copy()
            outd[f"{lossname}_{tag}_dblr"] = mdl.logistic_regres.bias.grad.numpy().copy()
    save("g11_dplda_input_grads.npz", W1=dp.centering_and_LDA.weight.detach().numpy(),
         b1=dp.centering_and_LDA.bias.detach().numpy(), wlr=dp.logistic_regres.weight.detach().numpy(),
         blr=dp.logistic_regres.bias.detach().numpy(), x1=xd1, x2=xd2, t=td, theta=np.asarray([0.2, 0.35]),
         beta=np.asarray([99.0, 199.0]), alpha=15.0, **outd)


class DrvConf:
    """The NpldaConf fields train() / validate() read (xvector_NeuralPlda_pytorch.py:44-50,75-82)."""
    log_interval = 1
    loss = "SoftCdet"
    beta = [99.0, 199.0]


def g12():
    rng = np.random.default_rng(1212)
    S, U, D0 = 24, 5, 64
    nutt = S * U
    utt_ids = [f"spk{u // U:03d}-utt{u:04d}" for u in range(nutt)]
    spk_mean = rng.standard_normal((S, D0))
    xv = (np.repeat(spk_mean, U, axis=0) + 0.8 * rng.standard_normal((nutt, D0))).astype(np.float32)
    mega = {u: xv[i] for i, u in enumerate(utt_ids)}
    num_to_id = {i: u for i, u in enumerate(utt_ids)}
    id_to_num = {u: i for i, u in enumerate(utt_ids)}

    def trials(n, ntgt):
        a = rng.integers(0, nutt, n)
        b = rng.integers(0, nutt, n)
        ta = rng.integers(0, S, ntgt) * U + rng.integers(0, U, ntgt)
        tb = (ta // U) * U + rng.integers(0, U, ntgt)
        a, b = np.concatenate([a, ta]), np.concatenate([b, tb])
        lab = (a // U == b // U).astype(int)
        return [f"{utt_ids[i]}\t{utt_ids[j]}\t{l}" for i, j, l in zip(a, b, lab)]

    train_lines, val_lines = trials(520, 120), trials(300, 60)
    ncs = NC(D0=D0, D1=24, D2=20)
    with tempfile.TemporaryDirectory() as td:
        trf, vaf = os.path.join(td, "train_trials.tsv"), os.path.join(td, "val_trials.tsv")
        open(trf, "w").write("\n".join(train_lines) + "\n")
        open(vaf, "w").write("\n".join(val_lines) + "\n")
        np.random.seed(12)
        torch.manual_seed(12)
        model = refm.NeuralPlda(ncs)
        p0 = {k: v.numpy().copy() for k, v in model.state_dict().items()}
        train_loader = refl.combine_trials_and_get_loader([trf], id_to_num, subsample_factors=[1.01], batch_size=64)
        valid = refl.get_trials_loaders_dict([vaf], id_to_num, subsample_factors=[1.01], batch_size=320)
        vkey = list(valid.keys())[0]
        dev = torch.device("cpu")
        nc = DrvConf()
        sink = io.StringIO()
        # 1) threshold initialisation exactly as the script does it (:142)
        torch.manual_seed(120)
        with redirect_stdout(sink):
            minc0, th0 = refdrv.validate(nc, model, dev, mega, num_to_id, valid[vkey], update_thresholds=True)
        th_init = np.asarray([model.Th99.item(), model.Th199.item()])
        # 2) one epoch of the reference's train() under Adam(lr, weight_decay=1e-5) (:139); per-batch losses recorded by
        #    wrapping model.loss (train() itself only prints running means)
        losses = []
        orig_loss = model.loss

        def rec(o, t_):
            L = orig_loss(o, t_)
            losses.append(float(L.item()))
            return L
        model.loss = rec
        optimizer = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5)
        torch.manual_seed(121)
        with redirect_stdout(sink):
            refdrv.train(nc, model, dev, train_loader, mega, num_to_id, optimizer, 1)
        model.loss = orig_loss
        p1 = {k: v.numpy().copy() for k, v in model.state_dict().items()}
        # 3) validate() after the epoch
        torch.manual_seed(122)
        with redirect_stdout(sink):
            minc1, th1 = refdrv.validate(nc, model, dev, mega, num_to_id, valid[vkey])
        log = sink.getvalue()
        save("g12_reference_driver.npz", xvec=xv, utt_ids=np.asarray(utt_ids),
             train_trials_text=np.asarray(open(trf).read()), val_trials_text=np.asarray(open(vaf).read()),
             val_key=np.asarray(vkey), keys=np.asarray(list(p0.keys())),
             **{"p0_" + k: v for k, v in p0.items()}, **{"p1_" + k: v for k, v in p1.items()},
             minc0=np.asarray(float(minc0)), th0=np.asarray([th0[99.0].item(), th0[199.0].item()]), th_init=th_init,
             losses=np.asarray(losses), lr=1e-3, weight_decay=1e-5, batch_size=64, val_batch_size=320,
             seeds=np.asarray([12, 120, 121, 122]),
             minc1=np.asarray(float(minc1)), th1=np.asarray([th1[99.0].item(), th1[199.0].item()]),
             driver_stdout=np.asarray(log))


def g7b():
    rg = np.random.default_rng(77)
    A = rg.standard_normal((340, 340)).astype(np.float32)
    Lt = (A @ A.T / 340 + np.eye(340, dtype=np.float32)).astype(np.float32)
    A = rg.standard_normal((340, 340)).astype(np.float32)
    Ln = (A @ A.T / 340 + 0.5 * np.eye(340, dtype=np.float32)).astype(np.float32)
    mt = (0.05 * rg.standard_normal(340)).astype(np.float32)
    mn = (0.05 * rg.standard_normal(340)).astype(np.float32)
    g1 = np.load(os.path.join(HERE, "g1_kaldi_params.npz"))
    g2 = np.load(os.path.join(HERE, "g2_forward_kaldi170.npz"))
    gb2 = refm.GaussianBackend(NC())
    with torch.no_grad():
        gb2.centering_and_LDA.weight.copy_(torch.from_numpy(g1["W1"]))
        gb2.centering_and_LDA.bias.copy_(torch.from_numpy(g1["b1"]))
    gb2.paired_cov_inv_target, gb2.paired_cov_inv_nontarget = torch.from_numpy(Lt), torch.from_numpy(Ln)
    gb2.paired_mean_target, gb2.paired_mean_nontarget = torch.from_numpy(mt), torch.from_numpy(mn)
    with torch.no_grad():
        s = gb2.forward(torch.from_numpy(g2["x1"]), torch.from_numpy(g2["x2"])).numpy()
        s64 = copy.deepcopy(gb2).double()
        s64.paired_cov_inv_target, s64.paired_cov_inv_nontarget = torch.from_numpy(Lt).double(), torch.from_numpy(Ln).double()
        s64.paired_mean_target, s64.paired_mean_nontarget = torch.from_numpy(mt).double(), torch.from_numpy(mn).double()
        s64v = s64.forward(torch.from_numpy(g2["x1"]).double(), torch.from_numpy(g2["x2"]).double()).numpy()
    save("g7_gb_kaldi170.npz", seed=77, s=s, s64=s64v, Lt=Lt, Ln=Ln, mt=mt, mn=mn)


def g4b():
    """The G4 trajectory (three Adam steps as xvector_NeuralPlda_pytorch.py:35-43,139 takes them) re-run by the
    reference in float64 from the same start: the yardstick that is free of the reference's own fp32 gradient noise."""
    g4 = np.load(os.path.join(HERE, "g4_adam_small.npz"))
    g3 = np.load(os.path.join(HERE, "g3_loss_grad_small.npz"))
    ma = refm.NeuralPlda(NC(D0=64, D1=24, D2=20)).double()
    ma.load_state_dict({str(k): torch.from_numpy(g4["p0_" + str(k)]).double() for k in g4["keys"]})
    ma.threshold = {99.0: ma.Th99, 199.0: ma.Th199}
    ma.alpha = torch.tensor(15.0, dtype=torch.float64)
    ma.lossfn = "SoftCdet"
    opt = torch.optim.Adam(ma.parameters(), lr=1e-4, weight_decay=1e-5)
    losses = []
    for step in range(3):
        opt.zero_grad()
        lo, hi = step * 128, (step + 1) * 128
        o = ma(torch.from_numpy(g3["x1"][lo:hi]).double(), torch.from_numpy(g3["x2"][lo:hi]).double())
        L = ma.loss(o, torch.from_numpy(g3["t"][lo:hi]).double())
        losses.append(L.item())
        L.backward()
        opt.step()
    save("g4_adam_small_f64.npz", losses=np.asarray(losses),
         **{"p3_" + k: v.numpy().copy() for k, v in ma.state_dict().items()})


if __name__ == "__main__":
    which = sys.argv[1:] or ["g11", "g12", "g7b", "g4b"]
    for name in which:
        globals()[name]()
