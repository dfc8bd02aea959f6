#!/usr/bin/env python3
"""validate() end to end at the reference's batch size (5 x batch_size = 20 480 trial pairs per forward call,
xvector_NeuralPlda_pytorch.py:125 / :56-83): 1 M validation trials over a 100 k-utterance table — device gather, fused
forward, scores collected, `minc` on the device.  Run once as is and once with NPLDA_FWD_NO_MID=1 (the round-2 dispatch:
20 480 pairs on the streaming kernel's part-filled round) to see what the balanced-tile kernel is worth to the loop.
usage: validate_e2e.py [D=170]"""
import contextlib, io, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from neuralplda_amd import models, ops, sv_trials_loaders as svl, train

D = int(sys.argv[1]) if len(sys.argv) > 1 else 170


class NC:
    xvector_dim, layer1_LDA_dim, layer2_PLDA_spkfactor_dim = 512, D, D
    beta, alpha, device, loss = [99.0, 199.0], 15.0, "cuda", "SoftCdet"
    log_interval, batch_size, lr = 100, 4096, 1e-4


def main():
    n_utt, n_trials = 100_000, 1_000_000
    rng = np.random.default_rng(0)
    ids = [f"spk{u // 5:05d}-utt{u:07d}" for u in range(n_utt)]
    xv = rng.standard_normal((n_utt, 512)).astype(np.float32)
    mega = {u: xv[i] for i, u in enumerate(ids)}
    num_to_id = dict(enumerate(ids))
    id_to_num = {u: i for i, u in num_to_id.items()}
    a, b = rng.integers(0, n_utt, n_trials), rng.integers(0, n_utt, n_trials)
    lab = (a // 5 == b // 5).astype(int)
    lab[rng.random(n_trials) < 0.05] = 1
    torch.manual_seed(0)
    m = models.NeuralPlda(NC()).cuda()
    with tempfile.TemporaryDirectory() as td:
        tf = os.path.join(td, "valid.tsv")
        with open(tf, "w") as f:
            f.write("\n".join(f"{ids[i]}\t{ids[j]}.wav\t{l}" for i, j, l in zip(a, b, lab)) + "\n")
        loaders = svl.get_trials_loaders_dict([tf], id_to_num, subsample_factors=[1.01], batch_size=5 * NC.batch_size)
    loader = loaders["valid"]
    svl.xvector_table(mega).on("cuda")
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            mc, th = train.validate(NC, m, torch.device("cuda"), mega, num_to_id, loader)
        torch.cuda.synchronize()
        t_val = time.perf_counter() - t0
    # the forward calls alone, at validate()'s batch size
    B = 5 * NC.batch_size
    x1 = torch.randn(B, 512, device="cuda"); x2 = torch.randn(B, 512, device="cuda")
    with torch.no_grad():
        for _ in range(50):
            m(x1, x2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            m(x1, x2)
        torch.cuda.synchronize()
    t_fwd = (time.perf_counter() - t0) / 200
    # gather + score against the fused form (what validate()'s device-resident loop runs per batch)
    table = svl.xvector_table(mega).on("cuda")
    r1 = torch.randint(0, n_utt, (B,), device="cuda"); r2 = torch.randint(0, n_utt, (B,), device="cuda")

    def tm(fn, n=200):
        for _ in range(30):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e6
    with torch.no_grad():
        t_gs = tm(lambda: m(ops.gather_rows(table, r1), ops.gather_rows(table, r2)))
        t_fu = tm(lambda: m.forward_rows(table, r1, r2))
    print(f"D={D} one validate() batch of {B} pairs by table rows: gather x2 + forward {t_gs:.1f} us, fused {t_fu:.1f} us")
    name = ops._lib.load().nplda_score_pairs_kernel_name(B, 512, D, D).decode().split(" ")[0]
    mode = "round-2 dispatch (NPLDA_FWD_NO_MID=1)" if os.environ.get("NPLDA_FWD_NO_MID") == "1" else "round-3 dispatch"
    print(f"D={D} {mode}: validate() over {len(loader.dataset)} trials in batches of {B}: {t_val * 1e3:.1f} ms "
          f"({len(loader.dataset) / t_val:.3e} trials/s, minC {float(mc):.4f}); model(x1, x2) on {B} pairs: {t_fwd * 1e6:.1f} us "
          f"({B / t_fwd:.3e} pairs/s, {name})")


if __name__ == "__main__":
    main()
