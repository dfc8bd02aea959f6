"""Seeded synthetic data shared by tests, the golden generator and bench.py.

speaker_structured_xvectors(): S speakers x U utterances of 512-d x-vectors whose PLDA latent
follows the Kaldi model (y_s ~ N(0, diag Psi), z = y_s + c*eps), mapped back through the PLDA
diagonalising transform and the (pseudo-inverted) LDA, plus noise in the LDA null space.
"""
import numpy as np


def speaker_structured_xvectors(W1, b1, diag_transform, plda_mean, psi, S, U, c=2.0, seed=7):
    r = np.random.default_rng(seed)
    W1 = np.asarray(W1, np.float64)
    b1 = np.asarray(b1, np.float64)
    D1 = W1.shape[0]
    ys = r.standard_normal((S, D1)) * np.sqrt(np.asarray(psi, np.float64))
    zl = np.repeat(ys, U, axis=0) + c * r.standard_normal((S * U, D1))
    u = np.linalg.solve(np.asarray(diag_transform, np.float64), zl.T).T + np.asarray(plda_mean, np.float64)
    W1p = np.linalg.pinv(W1)
    x = (W1p @ (u - b1).T).T
    nperp = r.standard_normal((S * U, W1.shape[1]))
    nperp = nperp - (W1p @ (W1 @ nperp.T)).T
    spk = np.repeat(np.arange(S), U)
    return (x + nperp).astype(np.float32), spk


def trial_list(spk, n_random, n_target_speakers, U, seed=7):
    """Random pairs plus every same-speaker pair of the first n_target_speakers speakers."""
    r = np.random.default_rng(seed + 1)
    n = spk.shape[0]
    ia = r.integers(0, n, n_random)
    ib = r.integers(0, n, n_random)
    ta, tb = [], []
    for sp in range(n_target_speakers):
        for p in range(U):
            for q in range(p + 1, U):
                ta.append(sp * U + p)
                tb.append(sp * U + q)
    ia = np.concatenate([ia, np.asarray(ta, dtype=np.int64)])
    ib = np.concatenate([ib, np.asarray(tb, dtype=np.int64)])
    t = (spk[ia] == spk[ib]).astype(np.float32)
    return ia, ib, t
