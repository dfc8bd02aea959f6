#!/usr/bin/env python3
"""End-to-end training epoch through the reference's own loop shape (train() over a DataLoader of index batches):
1 M trials over a 100 k-utterance table, batch 2048, SoftCdet, fused step.  Prints loader construction time and epoch
wall-clock."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from neuralplda_amd import models, sv_trials_loaders as svl, train


class NC:
    xvector_dim, layer1_LDA_dim, layer2_PLDA_spkfactor_dim = 512, 170, 170
    beta, alpha, device, loss = [99.0, 199.0], 15.0, "cuda", "SoftCdet"
    log_interval, batch_size, lr = 100, 2048, 1e-4


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
    lab[rng.random(n_trials) < 0.1] = 1
    torch.manual_seed(0)
    m = models.NeuralPlda(NC()).cuda()
    with tempfile.TemporaryDirectory() as td:
        tf = os.path.join(td, "train.tsv")
        with open(tf, "w") as f:
            f.write("\n".join(f"{ids[i]}\t{ids[j]}\t{l}" for i, j, l in zip(a, b, lab)) + "\n")
        t0 = time.perf_counter()
        loader = svl.combine_trials_and_get_loader([tf], id_to_num, subsample_factors=[1.01], batch_size=NC.batch_size)
        t_load = time.perf_counter() - t0
    svl.xvector_table(mega).on("cuda")
    opt = train.make_optimizer(m, NC.lr)
    step = train.FusedTrainStep(m, NC.lr, weight_decay=1e-5, batch_size=NC.batch_size, graph=True)
    import io, contextlib
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            train.train(NC, m, torch.device("cuda"), loader, mega, num_to_id, opt, 1, step_fn=step)
        torch.cuda.synchronize()
        t_epoch = time.perf_counter() - t0
    nb = len(loader)
    # the parts of that epoch: setting the epoch up (permutation, index arrays to the device, batch records) and stepping
    table, row_map = train._device_table(mega, num_to_id, torch.device("cuda"))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    records, tail = loader.device_epoch(torch.device("cuda"), row_map)
    step.begin_epoch(table, records)
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t0
    n = records.shape[0]
    t0 = time.perf_counter()
    for _ in range(n):
        step.step_record()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_steps = time.perf_counter() - t0
    step.pop_loss_mean()
    print(f"  of which epoch set-up {t_setup * 1e3:.1f} ms; {n} steps: host loop {t_host / n * 1e6:.1f} us/step, "
          f"device-complete {t_steps / n * 1e6:.1f} us/step")
    print(f"loader from a {n_trials}-trial TSV: {t_load:.3f} s")
    print(f"epoch: {nb} batches of {NC.batch_size} in {t_epoch:.3f} s = {t_epoch / nb * 1e3:.3f} ms/step, "
          f"{len(loader.dataset) / t_epoch:.3e} pairs/s")


if __name__ == "__main__":
    main()
