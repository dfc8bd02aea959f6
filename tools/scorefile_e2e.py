#!/usr/bin/env python3
"""End-to-end score-file generation (the reference's generate_voices_scores / generate_sre_scores workflow) at a
realistic size: 100 k utterances, 500 k trials.  Prints wall-clock of the whole call and of its phases."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from neuralplda_amd import models, scorefile_generator as sg, sv_trials_loaders as svl, textio


class NC:
    xvector_dim, layer1_LDA_dim, layer2_PLDA_spkfactor_dim = 512, 170, 170
    beta, alpha, device, loss = [99.0, 199.0], 15.0, "cuda", "SoftCdet"


def main():
    n_utt, n_trials = 100_000, 500_000
    rng = np.random.default_rng(0)
    ids = [f"spk{u // 5:05d}-utt{u:07d}" for u in range(n_utt)]
    xv = rng.standard_normal((n_utt, 512)).astype(np.float32)
    mega = {u: xv[i] for i, u in enumerate(ids)}
    a, b = rng.integers(0, n_utt, n_trials), rng.integers(0, n_utt, n_trials)
    torch.manual_seed(0)
    m = models.NeuralPlda(NC()).cuda()
    with tempfile.TemporaryDirectory() as td:
        tf = os.path.join(td, "trials.tsv")
        with open(tf, "w") as f:
            f.write("modelid\tsegmentid\tside\n")
            f.write("\n".join(f"{ids[i]}\t{ids[j]}.sph\ta" for i, j in zip(a, b)) + "\n")
        out = os.path.join(td, "scores.tsv")
        t0 = time.perf_counter()
        tab = svl.xvector_table(mega)            # one-off: dict -> (N, 512) matrix (+ device copy on first use)
        tab.on("cuda")
        torch.cuda.synchronize()
        t_table = time.perf_counter() - t0
        for rep in range(2):                      # second call = steady state (table and id blob cached)
            t0 = time.perf_counter()
            sg.generate_sre_scores(out, tf, mega, m, torch.device("cuda"))
            torch.cuda.synchronize()
            t_call = time.perf_counter() - t0
        text = open(tf, "rb").read()
        t0 = time.perf_counter(); rows, nc = textio.scan(text); t_scan = time.perf_counter() - t0
        t0 = time.perf_counter(); r1, r2, _, _, _ = textio.lookup(text, tab.idblob, 1, 2, 2, rows=rows); t_look = time.perf_counter() - t0
        t0 = time.perf_counter(); S = sg._score_rows(m, tab, r1, r2, torch.device("cuda")); torch.cuda.synchronize(); t_score = time.perf_counter() - t0
        t0 = time.perf_counter(); textio.write_scores(out, text, S, skip_rows=1, keep_cols=nc, header="h"); t_write = time.perf_counter() - t0
        size = os.path.getsize(out)
    print(f"x-vector table build + upload (one-off): {t_table:.3f} s")
    print(f"generate_sre_scores, {n_trials} trials over {n_utt} utterances: {t_call:.3f} s  ({n_trials / t_call:.3e} trials/s, "
          f"{size / 1e6:.1f} MB written)")
    print(f"  phases: scan {t_scan * 1e3:.1f} ms, id lookup {t_look * 1e3:.1f} ms, embed + indexed scoring + copy back "
          f"{t_score * 1e3:.1f} ms, write {t_write * 1e3:.1f} ms")


if __name__ == "__main__":
    main()
