"""Score-file writers — counterpart of utils/scorefile_generator.py:22-56.

Same signatures and the same TSV layouts as the reference (`sre`: header line + every input column +
an `LLR` column; `voices`: first two columns + score, no header), but the model is NOT forced to the
CPU: every distinct utterance of the trials file is embedded once on the device
(nplda_embed_f32) and the trial list is scored from index pairs (nplda_score_indexed_f32) — the
"Regime B" path of SURVEY.md §8d.  A trial list whose length is a multiple of `batch_size` works
(the reference crashes on the empty last chunk, utils/scorefile_generator.py:29-33).
"""
import os

import numpy as np
import torch

from . import ops
from .sv_trials_loaders import xvector_table

__all__ = ["generate_sre_scores", "generate_voices_scores", "score_trials"]


def _model_packed(model, dev):
    prm = [t.detach().float().to(dev) for t in (model.centering_and_LDA.weight, model.centering_and_LDA.bias,
                                                model.centering_and_wccn_plda.weight,
                                                model.centering_and_wccn_plda.bias, model.P_sqrt, model.Q)]
    return ops.pack_params(*prm)


def _pick_device(model, device):
    device = torch.device(device) if device is not None else None
    if device is not None and device.type == "cuda":
        return device
    p = next(model.parameters())
    if p.is_cuda:
        return p.device
    if not torch.cuda.is_available():
        from . import _lib
        raise _lib.NpldaHipError("score generation needs a HIP device (there is no CPU implementation)")
    return torch.device("cuda", torch.cuda.current_device())


def score_trials(model, mega_dict, ids1, ids2, device=None, batch_size=1 << 22):
    """Scores of the trials (ids1[i], ids2[i]) as a float32 numpy array.  Ids are normalised like
    load_xvec_trials_from_idbatch does (basename, extension stripped: utils/sv_trials_loaders.py:432)."""
    if hasattr(model, "centering_and_wccn_plda"):
        dev = _pick_device(model, device)
        tab = xvector_table(mega_dict)
        norm = lambda d: os.path.splitext(os.path.basename(d))[0]  # noqa: E731
        n1 = [norm(d) for d in ids1]
        n2 = [norm(d) for d in ids2]
        uniq = {}
        for u in n1:
            uniq.setdefault(u, len(uniq))
        for u in n2:
            uniq.setdefault(u, len(uniq))
        try:
            rows = np.fromiter((tab.row_of[u] for u in uniq), dtype=np.int64, count=len(uniq))
        except KeyError as e:
            raise KeyError(f"utterance {e.args[0]!r} is not in mega_dict") from None
        i1 = np.fromiter((uniq[u] for u in n1), dtype=np.int64, count=len(n1))
        i2 = np.fromiter((uniq[u] for u in n2), dtype=np.int64, count=len(n2))
        if len(i1) == 0:
            return np.zeros(0, np.float32)
        with torch.no_grad():
            packed = _model_packed(model, dev)
            x = tab.gather(rows, dev)                       # device gather of the distinct utterances
            z, q = ops.embed(x, packed)                     # embed each utterance ONCE
            out = []
            for lo in range(0, len(i1), batch_size):        # index pairs -> scores
                out.append(ops.score_indexed(z, q, torch.from_numpy(i1[lo:lo + batch_size]),
                                             torch.from_numpy(i2[lo:lo + batch_size]), packed))
            return torch.cat(out).cpu().numpy()
    # models without the NPLDA head (GaussianBackend): dense batched forward on the device
    from .sv_trials_loaders import load_xvec_trials_from_idbatch
    dev = _pick_device(model, device)
    trials = np.c_[np.asarray(ids1), np.asarray(ids2)]
    out = []
    with torch.no_grad():
        for lo in range(0, len(trials), min(batch_size, 1 << 18)):
            x1, x2 = load_xvec_trials_from_idbatch(mega_dict, trials[lo:lo + min(batch_size, 1 << 18)], dev)
            out.append(model.forward(x1, x2))
    return torch.cat(out).cpu().numpy() if out else np.zeros(0, np.float32)


def generate_sre_scores(score_filename, trials_file, mega_dict, model, device, batch_size=102400):
    """utils/scorefile_generator.py:22-39: header + input columns + LLR."""
    trials = np.genfromtxt(trials_file, dtype='str')
    trials = trials.reshape(-1, trials.shape[-1]) if trials.ndim == 2 else trials.reshape(1, -1)
    header = '\t'.join(trials[0]) + '\tLLR'
    trials = trials[1:]
    was_training = model.training
    model = model.eval()
    S = score_trials(model, mega_dict, trials[:, 0], trials[:, 1], device)
    scores = np.asarray(S).astype(str)
    np.savetxt(score_filename, np.c_[trials, scores], header=header, fmt='%s', delimiter='\t', comments='')
    if was_training:
        model.train()


def generate_voices_scores(score_filename, trials_file, mega_dict, model, device, batch_size=102400):
    """utils/scorefile_generator.py:41-56: first two columns + score, no header."""
    trials = np.genfromtxt(trials_file, dtype='str')
    trials = (trials.reshape(-1, trials.shape[-1]) if trials.ndim == 2 else trials.reshape(1, -1))[:, :2]
    was_training = model.training
    model = model.eval()
    S = score_trials(model, mega_dict, trials[:, 0], trials[:, 1], device)
    scores = np.asarray(S).astype(str)
    np.savetxt(score_filename, np.c_[trials, scores], fmt='%s', delimiter='\t', comments='')
    if was_training:
        model.train()
