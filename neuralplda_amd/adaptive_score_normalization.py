"""Adaptive score normalisation — counterpart of utils/adaptive_score_normalization.py.

The reference file is a module-level script with two hard-coded paths (:17-18) that reads a raw-score
TSV and a cohort-score TSV (written by Kaldi), sorts every cohort row on the host and loops over the
trials in Python.  Two entry points here:

* `normalize_scorefile(raw, cohort, ...)` — the same file-in / file-out contract (same input layouts,
  same four output files `<raw>_znorm.tsv`, `_tnorm.tsv`, `_snorm.tsv`, `_asnorm1.tsv`, same header
  handling, `.sph` stripped from test ids), with the per-row statistics and the per-trial
  normalisation done by the HIP kernels nplda_row_stats_f32 / nplda_asnorm_apply_f64.
* `asnorm_scores(model, ...)` — the full device pipeline the reference does not have: embed enroll,
  test and cohort utterances once, build the cohort score matrix on MFMA tiles, reduce it to row
  statistics (top-N included) and normalise the trial scores; rows shard across GPUs with one
  all-gather of the (R, 4) statistics (neuralplda_amd.dist).

Reference semantics kept by default: "top-N" = the N SMALLEST cohort scores (ascending sort then [:N],
:32-36), population std (ddof = 0).  `select="highest"` gives the conventional AS-norm.
Cohort scores are held in fp32 on the device (Kaldi's text scores carry <= 7 significant digits);
statistics and normalised scores are fp64 like the reference script.
"""
import numpy as np
import torch

from . import dist as ndist
from . import ops

__all__ = ["ASnorm_topN", "normalize_scorefile", "asnorm_scores", "cohort_row_stats"]

ASnorm_topN = 500  # utils/adaptive_score_normalization.py:12


def _device(device=None):
    if device is not None:
        return torch.device(device)
    if not torch.cuda.is_available():
        from . import _lib
        raise _lib.NpldaHipError("adaptive score normalisation needs a HIP device (no CPU implementation)")
    return torch.device("cuda", torch.cuda.current_device())


def normalize_scorefile(raw_score_filename, cohort_score_filename, topN=ASnorm_topN, select="lowest", device=None,
                        write=True):
    """File-level equivalent of running utils/adaptive_score_normalization.py with its two paths set.
    Returns {"znorm","tnorm","snorm","asnorm1"} -> float64 numpy arrays (and writes the four TSVs)."""
    dev = _device(device)
    raw_tab = np.genfromtxt(raw_score_filename, dtype='str')
    header = raw_tab[0]
    raw_tab = raw_tab[1:]
    trials_enroll, trials_test = raw_tab[:, 0], raw_tab[:, 1]
    raw_scores = raw_tab[:, -1].astype(float)
    trials_test = np.asarray([w.replace('.sph', '') for w in trials_test])
    coh = np.genfromtxt(cohort_score_filename, dtype='str', skip_header=1)
    num_unlabelled = len(np.unique(coh[:, 1]))
    cohort_matrix = coh[:, -1].astype(float).reshape(-1, num_unlabelled)
    row_ids = coh[:, 0].reshape(-1, num_unlabelled)[:, 0]
    row_of = dict(zip(row_ids, range(len(row_ids))))  # later duplicates win, as in the reference's dict(zip())
    S = torch.from_numpy(np.ascontiguousarray(cohort_matrix, dtype=np.float32)).to(dev)
    stats = ops.row_stats(S, topn=topN, select=select)
    try:
        ie = np.fromiter((row_of[e] for e in trials_enroll), dtype=np.int64, count=len(trials_enroll))
        it = np.fromiter((row_of[t] for t in trials_test), dtype=np.int64, count=len(trials_test))
    except KeyError as e:
        raise KeyError(f"id {e.args[0]!r} of the trial list has no row in the cohort score file") from None
    out = ops.asnorm_apply(torch.from_numpy(raw_scores), ie, it, stats).cpu().numpy()
    res = {k: out[:, c] for c, k in enumerate(("znorm", "tnorm", "snorm", "asnorm1"))}
    if write:
        for k, col in res.items():
            np.savetxt(raw_score_filename + f'_{k}.tsv', np.c_[raw_tab[:, :-1], col.astype(str)],
                       header='\t'.join(header), fmt='%s', delimiter='\t')
    return res


def cohort_row_stats(z_rows, q_rows, z_coh, q_coh, packed, topN=ASnorm_topN, select="lowest", group=None):
    """(R, 4) float64 statistics of every row against the whole cohort; rows are sharded over the ranks
    of `group` (each rank needs complete cohort rows for its top-N) and the statistics all-gathered."""
    R = z_rows.shape[0]
    rank, ws = ndist.world(group)
    lo, hi = ndist.shard_bounds(R, ws, rank)
    local = ops.cohort_stats(z_rows[lo:hi], q_rows[lo:hi], z_coh, q_coh, packed, topn=topN, select=select)
    return ndist.all_gather_rows(local, R, group)


def asnorm_scores(model, x_rows, x_cohort, raw, ie, it, topN=ASnorm_topN, select="lowest", group=None):
    """Device pipeline: x_rows (R, D0) enroll+test x-vectors, x_cohort (M, D0), raw (T,) raw trial scores,
    ie / it (T,) row indices of each trial's enroll / test utterance.  Returns (T, 4) float64 on the device
    (columns znorm, tnorm, snorm, asnorm1).  With a process group, rows AND trials are sharded."""
    dev = x_rows.device
    prm = [t.detach().float().to(dev) for t in (model.centering_and_LDA.weight, model.centering_and_LDA.bias,
                                                model.centering_and_wccn_plda.weight,
                                                model.centering_and_wccn_plda.bias, model.P_sqrt, model.Q)]
    with torch.no_grad():
        packed = ops.pack_params(*prm)
        zc, qc = ops.embed(x_cohort, packed)
        rank, ws = ndist.world(group)
        R = x_rows.shape[0]
        lo, hi = ndist.shard_bounds(R, ws, rank)
        zr, qr = ops.embed(x_rows[lo:hi], packed)
        local = ops.cohort_stats(zr, qr, zc, qc, packed, topn=topN, select=select)
        stats = ndist.all_gather_rows(local, R, group)  # the ONE exchange step: R x 4 doubles
        raw = torch.as_tensor(raw)
        T = raw.shape[0]
        ie_t, it_t = torch.as_tensor(ie), torch.as_tensor(it)

        def shard(tlo, thi):
            return ops.asnorm_apply(raw[tlo:thi], ie_t[tlo:thi], it_t[tlo:thi], stats)

        return ndist.sharded_apply(shard, T, group=group, device=dev)
