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

__all__ = ["ASnorm_topN", "normalize_scorefile", "asnorm_scores", "cohort_row_stats", "CohortState"]

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
    from . import textio
    dev = _device(device)
    with open(raw_score_filename, "rb") as fh:
        raw_text = fh.read()
    with open(cohort_score_filename, "rb") as fh:
        fh.readline()  # skip_header=1: the first LINE, before any comment / blank-line handling (:27)
        coh_text = fh.read()
    raw_rows, raw_cols = textio.scan(raw_text)
    coh_rows, coh_cols = textio.scan(coh_text)
    if raw_rows < 1 or raw_cols < 3 or coh_cols < 3:
        raise ValueError("score files need at least (id, id, score) columns and a header row")
    header = textio.row_tokens(raw_text, 0)
    n_trials = raw_rows - 1
    raw_scores = textio.column_f64(raw_text, -1, n_trials, skip_rows=1)
    num_unlabelled = textio.count_unique(coh_text, 1)
    if num_unlabelled == 0 or coh_rows % num_unlabelled:
        raise ValueError("cohort score file: rows are not a whole number of cohort blocks")  # reshape(-1, M) fails (:32)
    R = coh_rows // num_unlabelled
    cohort_matrix = textio.column_f64(coh_text, -1, coh_rows).reshape(R, num_unlabelled)
    row_ids = textio.column_tokens(coh_text, 0, R, stride=num_unlabelled)
    blob = textio.IdBlob(row_ids)  # a repeated id resolves to its last block, as the reference's dict(zip()) does
    S = torch.from_numpy(np.ascontiguousarray(cohort_matrix, dtype=np.float32)).to(dev)
    stats = ops.row_stats(S, topn=topN, select=select)
    ie, it, _, _, bad = textio.lookup(raw_text, blob, 1, textio.RAW, textio.STRIP_SPH, rows=raw_rows)
    if bad >= 0:
        toks = textio.row_tokens(raw_text, bad + 1)
        cand = [toks[0], toks[1].replace('.sph', '')]
        known = set(row_ids)
        miss = [c for c in cand if c not in known]
        raise KeyError(f"id {(miss[0] if miss else cand)!r} of the trial list has no row in the cohort score file")
    out = ops.asnorm_apply(torch.from_numpy(raw_scores), ie, it, stats).cpu().numpy()
    res = {k: np.ascontiguousarray(out[:, c]) for c, k in enumerate(("znorm", "tnorm", "snorm", "asnorm1"))}
    if write:
        for k, col in res.items():  # np.savetxt's default comment prefix "# " goes in front of the header (:81-84)
            textio.write_scores(raw_score_filename + f'_{k}.tsv', raw_text, col, skip_rows=1, keep_cols=raw_cols - 1,
                                header='# ' + '\t'.join(header))
    return res


def _model_params(model, dev):
    return [t.detach().float().to(dev) for t in (model.centering_and_LDA.weight, model.centering_and_LDA.bias,
                                                 model.centering_and_wccn_plda.weight,
                                                 model.centering_and_wccn_plda.bias, model.P_sqrt, model.Q)]


class CohortState:
    """A cohort made ready ONCE per (model, cohort, top-N) for every trial list scored against it
    (utils/adaptive_score_normalization.py:27-36 reads one cohort score table for all of its trials): the parameter
    image, the cohort's embeddings `z_coh`, `q_coh` (utils/models.py:366-370 per cohort utterance) and what
    nplda_cohort_stats_f32 derives from the cohort alone (Gram matrix, first moments, the covariance image its row
    thresholds are proposed from: nplda_cohort_prepare_f32).  Without it every `asnorm_scores` call — and every RANK of a
    row-sharded call — embeds the cohort and redoes that pre-pass: 63 us of the 250 us a rank spends on an eighth of
    cfg3.  Build with `CohortState.build(model, x_cohort, topN)`; hand it to `asnorm_scores(..., cohort=state)` or
    `cohort_row_stats(..., prepared=state.prepared)`.  Valid while the model's parameters are unchanged (`check(model)`,
    which `asnorm_scores` calls, raises once they are not)."""

    def __init__(self, packed, z_coh, q_coh, prepared, topN, stamp=None):
        self.packed, self.z_coh, self.q_coh, self.prepared, self.topN = packed, z_coh, q_coh, prepared, int(topN)
        self.stamp = stamp  # (storage address, autograd version) of the six parameters it was built from

    @staticmethod
    def _stamp(model):
        return tuple((t.data_ptr(), t._version) for t in (
            model.centering_and_LDA.weight, model.centering_and_LDA.bias, model.centering_and_wccn_plda.weight,
            model.centering_and_wccn_plda.bias, model.P_sqrt, model.Q))

    def check(self, model):
        """Raise if `model`'s parameters were modified in place (an optimiser step, load_state_dict: the autograd version
        counters say so) or replaced since build(): the state's image and embeddings are the OLD model's."""
        if model is not None and self.stamp is not None and self._stamp(model) != self.stamp:
            raise ValueError("the model's parameters changed since this CohortState was built: build it again")

    @classmethod
    def build(cls, model, x_cohort, topN=ASnorm_topN):
        dev = x_cohort.device
        with torch.no_grad():
            packed = ops.pack_params(*_model_params(model, dev))
            zc, qc = ops.embed(x_cohort, packed)
            return cls(packed, zc, qc, ops.cohort_prepare(zc, qc, packed, topn=topN), topN, cls._stamp(model))


def cohort_row_stats(z_rows, q_rows, z_coh, q_coh, packed, topN=ASnorm_topN, select="lowest", group=None, prepared=None):
    """(R, 4) float64 statistics of every row against the whole cohort; rows are sharded over the ranks
    of `group` (each rank needs complete cohort rows for its top-N) and the statistics all-gathered.
    prepared: ops.cohort_prepare(z_coh, q_coh, packed, topN) (or CohortState.prepared) — the cohort-only pre-pass, once."""
    R = z_rows.shape[0]
    rank, ws = ndist.world(group)
    lo, hi = ndist.shard_bounds(R, ws, rank)
    local = ops.cohort_stats(z_rows[lo:hi], q_rows[lo:hi], z_coh, q_coh, packed, topn=topN, select=select, prepared=prepared)
    return ndist.all_gather_rows(local, R, group)


def asnorm_scores(model, x_rows, x_cohort, raw, ie, it, topN=ASnorm_topN, select="lowest", group=None, cohort=None):
    """Device pipeline: x_rows (R, D0) enroll+test x-vectors, x_cohort (M, D0), raw (T,) raw trial scores,
    ie / it (T,) row indices of each trial's enroll / test utterance.  Returns (T, 4) float64 on the device
    (columns znorm, tnorm, snorm, asnorm1).  With a process group, rows AND trials are sharded.
    cohort: a CohortState built once for this (model, x_cohort, topN) — x_cohort is then not touched (may be None)."""
    dev = x_rows.device
    with torch.no_grad():
        rank, ws = ndist.world(group)
        R = x_rows.shape[0]
        lo, hi = ndist.shard_bounds(R, ws, rank)
        if cohort is not None:
            if cohort.topN != int(topN):
                raise ValueError("the CohortState was prepared for another top-N")
            cohort.check(model)
            packed, zc, qc = cohort.packed, cohort.z_coh, cohort.q_coh
            zr, qr = ops.embed(x_rows[lo:hi], packed)
            local = ops.cohort_stats(zr, qr, zc, qc, packed, topn=topN, select=select, prepared=cohort.prepared)
        else:
            packed = ops.pack_params(*_model_params(model, dev))
            # this rank's rows and the (replicated) cohort in ONE embedding launch (nplda_embed_pair_f32)
            (zr, qr), (zc, qc) = ops.embed_pair(x_rows[lo:hi], x_cohort, packed)
            local = ops.cohort_stats(zr, qr, zc, qc, packed, topn=topN, select=select)
        stats = ndist.all_gather_rows(local, R, group)  # the ONE exchange step: R x 4 doubles
        raw = torch.as_tensor(raw)
        T = raw.shape[0]
        ie_t, it_t = torch.as_tensor(ie), torch.as_tensor(it)

        def shard(tlo, thi):
            return ops.asnorm_apply(raw[tlo:thi], ie_t[tlo:thi], it_t[tlo:thi], stats)

        return ndist.sharded_apply(shard, T, group=group, device=dev)
