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


def _score_rows(model, tab, r1, r2, device, batch_size=1 << 22):
    """Scores of trials given as x-vector table rows (int64 numpy).  NPLDA: each distinct utterance is embedded once,
    then index pairs are scored; other models: dense batched forward of gathered rows."""
    if len(r1) == 0:
        return np.zeros(0, np.float32)
    dev = _pick_device(model, device)
    with torch.no_grad():
        if hasattr(model, "centering_and_wccn_plda"):
            used = np.zeros(len(tab.ids), dtype=bool)
            used[r1] = True
            used[r2] = True
            rows = np.flatnonzero(used)
            remap = np.cumsum(used) - 1
            j1, j2 = remap[r1], remap[r2]
            packed = _model_packed(model, dev)
            z, q = ops.embed(tab.gather(rows, dev), packed)
            out = [ops.score_indexed(z, q, torch.from_numpy(j1[lo:lo + batch_size]),
                                     torch.from_numpy(j2[lo:lo + batch_size]), packed)
                   for lo in range(0, len(j1), batch_size)]
        else:
            step = min(batch_size, 1 << 18)
            out = [model.forward(tab.gather(r1[lo:lo + step], dev), tab.gather(r2[lo:lo + step], dev))
                   for lo in range(0, len(r1), step)]
        return torch.cat(out).float().cpu().numpy()


def _generate(score_filename, trials_file, mega_dict, model, device, skip_rows, keep_cols, with_header):
    from . import textio
    with open(trials_file, "rb") as fh:
        text = fh.read()
    rows, ncols = textio.scan(text)
    if rows < skip_rows or (rows > skip_rows and ncols < 2):
        raise ValueError(f"{trials_file}: not a trials file")
    tab = xvector_table(mega_dict)
    r1, r2, _, _, bad = textio.lookup(text, tab.idblob, skip_rows, textio.BASENAME_SPLITEXT,
                                      textio.BASENAME_SPLITEXT, rows=rows)
    if bad >= 0:  # the reference raises KeyError at utils/sv_trials_loaders.py:433
        toks = textio.row_tokens(text, bad + skip_rows)
        norm = [os.path.splitext(os.path.basename(d))[0] for d in toks[:2]]
        missing = [u for u in norm if u not in tab.row_of]
        raise KeyError(f"utterance {missing[0] if missing else norm!r} is not in mega_dict")
    was_training = model.training
    model = model.eval()
    S = _score_rows(model, tab, r1, r2, device)
    header = None
    if with_header:
        header = '\t'.join(textio.row_tokens(text, 0)) + '\tLLR'
    textio.write_scores(score_filename, text, S, skip_rows=skip_rows, keep_cols=ncols if keep_cols is None else keep_cols,
                        header=header)
    if was_training:
        model.train()


# ---- binary score files: at 1e9 pairs/s the text is the bottleneck (~30 bytes a line) -----------------------------
# Layout: 16-byte header  b"NPLDASCR" | uint32 version = 1 | uint32 flags (bit 0: the trials file had a header row),
#         uint64 count, uint64 size of the trials file, 16 bytes = MD5 of the trials file (which names the rows: the scores are
#         in ITS row order), then `count` little-endian float32 scores.  The TSV writers above stay the interchange format; this
#         one is for pipelines that keep the trials file and want the numbers (AS-norm, calibration, metrics).
_BIN_MAGIC = b"NPLDASCR"


def generate_scores_binary(score_filename, trials_file, mega_dict, model, device=None, skip_rows=0):
    """Score `trials_file` like generate_sre_scores (skip_rows=1) / generate_voices_scores (skip_rows=0) but write the scores
    as a binary file tied to the trials file by size and MD5; returns the float32 scores."""
    import hashlib
    import struct
    from . import textio
    with open(trials_file, "rb") as fh:
        text = fh.read()
    rows, ncols = textio.scan(text)
    if rows < skip_rows or (rows > skip_rows and ncols < 2):
        raise ValueError(f"{trials_file}: not a trials file")
    tab = xvector_table(mega_dict)
    r1, r2, _, _, bad = textio.lookup(text, tab.idblob, skip_rows, textio.BASENAME_SPLITEXT,
                                      textio.BASENAME_SPLITEXT, rows=rows)
    if bad >= 0:
        toks = textio.row_tokens(text, bad + skip_rows)
        raise KeyError(f"utterance in {toks[:2]!r} is not in mega_dict")
    was_training = model.training
    model = model.eval()
    S = np.ascontiguousarray(_score_rows(model, tab, r1, r2, device), dtype="<f4")
    if was_training:
        model.train()
    with open(score_filename, "wb") as fh:
        fh.write(_BIN_MAGIC + struct.pack("<IIQQ", 1, 1 if skip_rows else 0, S.size, len(text)) + hashlib.md5(text).digest())
        fh.write(S.tobytes())
    return S


def load_scores_binary(score_filename, trials_file=None):
    """The float32 scores of a generate_scores_binary file; with `trials_file`, checks that it is the file the scores were
    made for (size and MD5) and raises ValueError otherwise."""
    import hashlib
    import struct
    with open(score_filename, "rb") as fh:
        head = fh.read(48)
        if len(head) != 48 or head[:8] != _BIN_MAGIC:
            raise ValueError(f"{score_filename}: not a binary score file")
        version, flags, count, tsize = struct.unpack("<IIQQ", head[8:32])
        if version != 1:
            raise ValueError(f"{score_filename}: unknown version {version}")
        S = np.frombuffer(fh.read(4 * count), dtype="<f4")
    if S.size != count:
        raise ValueError(f"{score_filename}: truncated ({S.size} of {count} scores)")
    if trials_file is not None:
        with open(trials_file, "rb") as fh:
            text = fh.read()
        if len(text) != tsize or hashlib.md5(text).digest() != head[32:48]:
            raise ValueError(f"{score_filename} was not made from {trials_file}")
    return S.astype(np.float32)


def generate_sre_scores(score_filename, trials_file, mega_dict, model, device, batch_size=102400):
    """utils/scorefile_generator.py:22-39: header + input columns + LLR.  One native pass reads the trials file and
    resolves both id columns to table rows (nplda_text_lookup), one writes the TSV (nplda_scores_write)."""
    _generate(score_filename, trials_file, mega_dict, model, device, skip_rows=1, keep_cols=None, with_header=True)


def generate_voices_scores(score_filename, trials_file, mega_dict, model, device, batch_size=102400):
    """utils/scorefile_generator.py:41-56: first two columns + score, no header."""
    _generate(score_filename, trials_file, mega_dict, model, device, skip_rows=0, keep_cols=2, with_header=False)
