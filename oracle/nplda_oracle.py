"""CPU ORACLE — test infrastructure, NOT product code.

A plain-NumPy restatement of the Neural-PLDA hot path of iiscleap/NeuralPlda, each function
citing the reference lines it follows.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this module; nothing under neuralplda_amd/ does (the product
path fails loudly when the HIP library is missing — it never falls back to this).

Pinning: the reference has no tests and no golden vectors of its own ("parity unpinned by the
reference's own tests").  This oracle is pinned instead against outputs of the
reference itself, generated in the build container by tests/golden/make_golden.py (which imports
/root/reference/utils/models.py) and committed as tests/golden/*.npz; see
tests/test_oracle_golden.py.

All functions take/return NumPy arrays and compute in the dtype of `dtype` (float32 mirrors the
reference's arithmetic; float64 is used as the tighter yardstick).
"""
import numpy as np

EPS_NORMALIZE = 1e-12  # torch.nn.functional.normalize default eps (utils/models.py:368)


def _as(a, dtype):
    return np.ascontiguousarray(np.asarray(a), dtype=dtype)


class Params:
    """The six tensors of NeuralPlda (utils/models.py:351-354), nn.Linear layout (out, in)."""

    def __init__(self, W1, b1, W2, b2, P_sqrt, Q):
        self.W1, self.b1, self.W2, self.b2, self.P_sqrt, self.Q = W1, b1, W2, b2, P_sqrt, Q

    def astype(self, dtype):
        return Params(*[_as(t, dtype) for t in (self.W1, self.b1, self.W2, self.b2, self.P_sqrt, self.Q)])

    def tensors(self):
        return (self.W1, self.b1, self.W2, self.b2, self.P_sqrt, self.Q)


def layer1(x, p, dtype=np.float32):
    """u = W1 x + b1 (utils/models.py:367, nn.Linear)."""
    return _as(x, dtype) @ _as(p.W1, dtype).T + _as(p.b1, dtype)


def normalize(u, dtype=np.float32):
    """F.normalize(u): u / max(||u||_2, 1e-12) row-wise (utils/models.py:368). Returns (y, norm)."""
    u = _as(u, dtype)
    nrm = np.sqrt(np.sum(u * u, axis=1, keepdims=True, dtype=dtype))
    den = np.maximum(nrm, dtype(EPS_NORMALIZE))
    return u / den, nrm[:, 0]


def extract_plda_embeddings(x, p, dtype=np.float32, with_intermediates=False):
    """z = W2 normalize(W1 x + b1) + b2 (utils/models.py:366-370)."""
    u = layer1(x, p, dtype)
    y, nrm = normalize(u, dtype)
    z = y @ _as(p.W2, dtype).T + _as(p.b2, dtype)
    if with_intermediates:
        return z, (u, y, nrm)
    return z


def forward_from_plda_embeddings(z1, z2, p, dtype=np.float32):
    """S = sum(z1 Q z1) + sum(z2 Q z2) + 2 sum(z1 P z2), P = P_sqrt^2 (utils/models.py:372-376)."""
    z1, z2 = _as(z1, dtype), _as(z2, dtype)
    P = _as(p.P_sqrt, dtype) * _as(p.P_sqrt, dtype)
    Q = _as(p.Q, dtype)
    return (z1 * Q * z1).sum(axis=1) + (z2 * Q * z2).sum(axis=1) + 2 * (z1 * P * z2).sum(axis=1)


def forward(x1, x2, p, dtype=np.float32):
    """NeuralPlda.forward (utils/models.py:378-382)."""
    x1 = _as(x1, dtype).reshape(-1, np.asarray(p.W1).shape[1])
    x2 = _as(x2, dtype).reshape(-1, np.asarray(p.W1).shape[1])
    z1 = extract_plda_embeddings(x1, p, dtype)
    z2 = extract_plda_embeddings(x2, p, dtype)
    return forward_from_plda_embeddings(z1, z2, p, dtype)


def self_term(z, p, dtype=np.float32):
    """q_n = sum_d Q_d z_nd^2 — the per-utterance half of utils/models.py:375."""
    z = _as(z, dtype)
    return (z * _as(p.Q, dtype) * z).sum(axis=1)


def score_indexed(z, i1, i2, p, dtype=np.float32):
    """utils/models.py:372-376 evaluated on rows gathered from an embedding table
    (the gather is utils/sv_trials_loaders.py:418-426 moved after the embedding)."""
    z = _as(z, dtype)
    return forward_from_plda_embeddings(z[np.asarray(i1)], z[np.asarray(i2)], p, dtype)


# ---------------------------------------------------------------------------------------------
# losses (utils/models.py:384-399) and their hand-derived gradients
# ---------------------------------------------------------------------------------------------

def _sigmoid(v):
    out = np.empty_like(v)
    pos = v >= 0
    out[pos] = 1.0 / (1.0 + np.exp(-v[pos]))
    ev = np.exp(v[~pos])
    out[~pos] = ev / (1.0 + ev)
    return out


def softcdet(s, t, theta, beta, alpha, dtype=np.float32):
    """utils/models.py:384-388.  theta, beta: sequences of length K."""
    s, t = _as(s, dtype), _as(t, dtype)
    a = dtype(alpha)
    nt, nn = t.sum(dtype=dtype), (1 - t).sum(dtype=dtype)
    losses = []
    for th, b in zip(theta, beta):
        th = dtype(th)
        miss = (_sigmoid(a * (th - s)) * t).sum(dtype=dtype) / nt
        fa = (_sigmoid(a * (s - th)) * (1 - t)).sum(dtype=dtype) / nn
        losses.append(miss + dtype(b) * fa)
    return dtype(sum(losses) / dtype(len(losses)))


def softcdet_grad(s, t, theta, beta, alpha, dtype=np.float64, nt=None, nn=None):
    """Gradient of softcdet w.r.t. s (g) and theta (verified against autograd when
    the golden fixtures were generated).  nt/nn override the target/non-target counts (data
    parallelism: the counts are batch-global)."""
    s, t = _as(s, dtype), _as(t, dtype)
    a = dtype(alpha)
    nt = t.sum() if nt is None else dtype(nt)
    nn = (1 - t).sum() if nn is None else dtype(nn)
    K = len(beta)
    g = np.zeros_like(s)
    dtheta = np.zeros(K, dtype=dtype)
    for k, (th, b) in enumerate(zip(theta, beta)):
        sg = _sigmoid(a * (dtype(th) - s))
        d = sg * (1 - sg)
        g += (-a * t * d / nt + dtype(b) * a * (1 - t) * d / nn) / K
        dtheta[k] = (a * (t * d).sum() / nt - dtype(b) * a * ((1 - t) * d).sum() / nn) / K
    return g, dtheta


def crossentropy(s, t, theta_xent, dtype=np.float32):
    """F.binary_cross_entropy(sigmoid(s - theta), t), mean reduction (utils/models.py:390-393).
    torch clamps each log term at -100."""
    s, t = _as(s, dtype), _as(t, dtype)
    pr = _sigmoid(s - dtype(theta_xent))
    with np.errstate(divide="ignore"):
        lp = np.maximum(np.log(pr), dtype(-100))
        lq = np.maximum(np.log(1 - pr), dtype(-100))
    return dtype(-(t * lp + (1 - t) * lq).mean(dtype=dtype))


def crossentropy_grad(s, t, theta_xent, dtype=np.float64, n_total=None):
    s, t = _as(s, dtype), _as(t, dtype)
    n = s.shape[0] if n_total is None else n_total
    g = (_sigmoid(s - dtype(theta_xent)) - t) / dtype(n)
    return g, np.asarray([-g.sum()], dtype=dtype)


def backward(x1, x2, g, p, dtype=np.float64):
    """Gradients of sum_i g_i * s_i w.r.t. all six parameter tensors.
    Returns dict(W1,b1,W2,b2,P_sqrt,Q)."""
    p = p.astype(dtype)
    g = _as(g, dtype)[:, None]
    z1, (u1, y1, n1) = extract_plda_embeddings(x1, p, dtype, True)
    z2, (u2, y2, n2) = extract_plda_embeddings(x2, p, dtype, True)
    P = p.P_sqrt * p.P_sqrt
    dQ = (g * (z1 * z1 + z2 * z2)).sum(axis=0)
    dPs = 4 * p.P_sqrt * (g * z1 * z2).sum(axis=0)
    dz1 = 2 * g * (p.Q * z1 + P * z2)
    dz2 = 2 * g * (p.Q * z2 + P * z1)
    dW2 = dz1.T @ y1 + dz2.T @ y2
    db2 = (dz1 + dz2).sum(axis=0)

    du1, du2 = _normalize_bwd(dz1 @ p.W2, y1, n1), _normalize_bwd(dz2 @ p.W2, y2, n2)
    x1d, x2d = _as(x1, dtype), _as(x2, dtype)
    dW1 = du1.T @ x1d + du2.T @ x2d
    db1 = (du1 + du2).sum(axis=0)
    return dict(W1=dW1, b1=db1, W2=dW2, b2=db2, P_sqrt=dPs, Q=dQ)


def _normalize_bwd(dy, y, nrm):
    """Backward of F.normalize (utils/models.py:368): du = (dy - y (y . dy)) / max(||u||, eps); where ||u|| < eps the
    denominator is the constant eps."""
    den = np.maximum(nrm, EPS_NORMALIZE)[:, None]
    du = (dy - y * (y * dy).sum(axis=1, keepdims=True)) / den
    return np.where((nrm < EPS_NORMALIZE)[:, None], dy / den, du)


def embed_backward(x, gz, p, dtype=np.float64):
    """Backward of z = extract_plda_embeddings(x) (utils/models.py:366-370) for an upstream gradient gz = dL/dz:
    dict(W1, b1, W2, b2, x) — what the reference's autograd returns for these three lines (pinned by golden G11)."""
    p = p.astype(dtype)
    gz = _as(gz, dtype)
    _, (u, y, nrm) = extract_plda_embeddings(x, p, dtype, True)
    du = _normalize_bwd(gz @ p.W2, y, nrm)
    xd = _as(x, dtype)
    return dict(W1=du.T @ xd, b1=du.sum(axis=0), W2=gz.T @ y, b2=gz.sum(axis=0), x=du @ p.W1)


def embscore_backward(z1, z2, g, p, dtype=np.float64):
    """Backward of s = forward_from_plda_embeddings(z1, z2) (utils/models.py:372-376) for g = dL/ds:
    dict(z1, z2, P_sqrt, Q)."""
    z1, z2, g = _as(z1, dtype), _as(z2, dtype), _as(g, dtype)[:, None]
    ps, Q = _as(p.P_sqrt, dtype), _as(p.Q, dtype)
    P = ps * ps
    return dict(z1=2 * g * (Q * z1 + P * z2), z2=2 * g * (Q * z2 + P * z1),
                P_sqrt=4 * ps * (g * z1 * z2).sum(axis=0), Q=(g * (z1 * z1 + z2 * z2)).sum(axis=0))


def input_grads(x1, x2, g, p, dtype=np.float64):
    """dL/dx1, dL/dx2 of sum_i g_i s_i, s = forward(x1, x2) (utils/models.py:378-382): the chain of the two functions
    above — the gradient the E2E model (utils/models.py:251-268) passes to its x-vector extractor."""
    p = p.astype(dtype)
    z1 = extract_plda_embeddings(x1, p, dtype)
    z2 = extract_plda_embeddings(x2, p, dtype)
    d = embscore_backward(z1, z2, g, p, dtype)
    return embed_backward(x1, d["z1"], p, dtype)["x"], embed_backward(x2, d["z2"], p, dtype)["x"]


# ---------------------------------------------------------------------------------------------
# metrics (utils/models.py:401-436)
# ---------------------------------------------------------------------------------------------

def cdet(s, t, theta, beta, dtype=np.float32):
    """Hard detection cost at the model thresholds, strict inequalities (utils/models.py:401-404)."""
    s, t = _as(s, dtype), _as(t, dtype)
    nt, nn = t.sum(dtype=dtype), (1 - t).sum(dtype=dtype)
    losses = []
    for th, b in zip(theta, beta):
        miss = ((s < dtype(th)).astype(dtype) * t).sum(dtype=dtype) / nt
        fa = ((s > dtype(th)).astype(dtype) * (1 - t)).sum(dtype=dtype) / nn
        losses.append(miss + dtype(b) * fa)
    return dtype(sum(losses) / dtype(len(losses)))


def minc_reference(s, t, beta):
    """Bit-compatible restatement of NeuralPlda.minc (utils/models.py:406-436) INCLUDING its
    quirks (utils/models.py:23-27 `arr2val`): the "count" is the LAST INDEX of torch.where (count-1)
    and 1.0 when empty; thresholds are the target scores only.  O(N log N) via searchsorted instead
    of the reference's O(N_tgt * N) Python loop.  Returns (minc_avg float32, {beta: threshold})."""
    s = _as(s, np.float32)
    t = _as(t, np.float32)
    st = np.sort(s[t > 0.5])
    sn = np.sort(s[t < 0.5])  # ascending; the reference keeps it descending, only counts matter
    # pmiss_arr[i] = last index of {j: st[j] < st[i]} = count-1, or 1.0 if none
    c_lt = np.searchsorted(st, st, side="left")  # number of targets strictly below st[i]
    pmiss_arr = np.where(c_lt > 0, c_lt - 1, 1).astype(np.float32)
    # pfa_arr[i] = last index of {j: sn_desc[j] >= st[i]} = count-1, or 1.0 if none
    c_ge = sn.shape[0] - np.searchsorted(sn, st, side="left")
    pfa_arr = np.where(c_ge > 0, c_ge - 1, 1).astype(np.float32)
    pmiss = pmiss_arr / np.float32(t.sum(dtype=np.float32))
    pfa = pfa_arr / np.float32((1 - t).sum(dtype=np.float32))
    mincs, ths = [], {}
    for b in beta:
        c = pmiss + np.float32(b) * pfa
        idx = int(np.argmin(c))
        mincs.append(c[idx])
        ths[b] = st[idx]
    return np.float32(sum(mincs) / np.float32(len(mincs))), ths


def minc_exact(s, t, beta):
    """Exact minimum detection cost: sweep every distinct score as threshold (decide target iff
    s >= th), P_miss = #{tgt < th}/N_t, P_fa = #{non >= th}/N_n, plus the accept-nothing point."""
    s = _as(s, np.float64)
    t = _as(t, np.float64)
    st = np.sort(s[t > 0.5])
    sn = np.sort(s[t < 0.5])
    th = np.unique(np.concatenate([s, [np.inf]]))
    pmiss = np.searchsorted(st, th, side="left") / max(st.shape[0], 1)
    pfa = (sn.shape[0] - np.searchsorted(sn, th, side="left")) / max(sn.shape[0], 1)
    out, ths = [], {}
    for b in beta:
        c = pmiss + b * pfa
        i = int(np.argmin(c))
        out.append(c[i])
        ths[b] = th[i]
    return float(np.mean(out)), ths


def eer(s, t):
    """Equal error rate by the same threshold sweep (linear interpolation at the crossing)."""
    s = _as(s, np.float64)
    t = _as(t, np.float64)
    st = np.sort(s[t > 0.5])
    sn = np.sort(s[t < 0.5])
    th = np.unique(s)
    pmiss = np.searchsorted(st, th, side="left") / st.shape[0]
    pfa = (sn.shape[0] - np.searchsorted(sn, th, side="left")) / sn.shape[0]
    d = pmiss - pfa
    i = int(np.argmax(d >= 0))
    if i == 0:
        return float((pmiss[0] + pfa[0]) / 2)
    x0, x1 = d[i - 1], d[i]
    w = -x0 / (x1 - x0) if x1 != x0 else 0.5
    return float(pmiss[i - 1] + w * (pmiss[i] - pmiss[i - 1]))


# ---------------------------------------------------------------------------------------------
# adaptive score normalisation (utils/adaptive_score_normalization.py:27-73)
# ---------------------------------------------------------------------------------------------

def cohort_stats(C, topn=500, select="lowest"):
    """Per-row (mean, std, mean_top, std_top) of a cohort score matrix C (R, M), float64, population
    std (ddof=0).  `lowest` follows the reference: rows are sorted ASCENDING and the FIRST topn kept
    (adaptive_score_normalization.py:32-36).  Returns (R, 4) float64."""
    C = np.sort(_as(C, np.float64), axis=1)
    top = C[:, :topn] if select == "lowest" else C[:, -topn:]
    return np.stack([C.mean(axis=1), C.std(axis=1), top.mean(axis=1), top.std(axis=1)], axis=1)


def cohort_scores(z_rows, z_coh, p, dtype=np.float32):
    """Cohort score matrix C[r, m] = NeuralPlda score of (row r, cohort m), i.e.
    forward_from_plda_embeddings on the expanded pair list (utils/models.py:372-376)."""
    z_rows, z_coh = _as(z_rows, dtype), _as(z_coh, dtype)
    P = _as(p.P_sqrt, dtype) * _as(p.P_sqrt, dtype)
    qr, qc = self_term(z_rows, p, dtype), self_term(z_coh, p, dtype)
    return qr[:, None] + qc[None, :] + 2 * ((z_rows * P) @ z_coh.T)


def asnorm_apply(raw, ie, it, stats):
    """z/t/s/as-norm of raw scores given per-row stats (adaptive_score_normalization.py:65-73).
    ie/it index the enroll/test row of each trial in `stats` (R, 4). Returns (T, 4) float64:
    columns znorm, tnorm, snorm, asnorm1."""
    raw = _as(raw, np.float64)
    se, st_ = stats[np.asarray(ie)], stats[np.asarray(it)]
    zn = (raw - se[:, 0]) / se[:, 1]
    tn = (raw - st_[:, 0]) / st_[:, 1]
    sn = (zn + tn) / 2
    an = ((raw - se[:, 2]) / se[:, 3] + (raw - st_[:, 2]) / st_[:, 3]) / 2
    return np.stack([zn, tn, sn, an], axis=1)


# ---------------------------------------------------------------------------------------------
# GaussianBackend.forward (utils/models.py:584-593)
# ---------------------------------------------------------------------------------------------

def gb_forward(x1, x2, W1, b1, mu_t, Lam_t, mu_n, Lam_n, dtype=np.float32):
    W1, b1 = _as(W1, dtype), _as(b1, dtype)
    y1, _ = normalize(_as(x1, dtype) @ W1.T + b1, dtype)
    y2, _ = normalize(_as(x2, dtype) @ W1.T + b1, dtype)
    x = np.concatenate([y1, y2], axis=1)
    dt = x - _as(mu_t, dtype)
    dn = x - _as(mu_n, dtype)
    St = (-(dt @ _as(Lam_t, dtype)) * dt).sum(axis=1)
    Snt = (-(dn @ _as(Lam_n, dtype)) * dn).sum(axis=1)
    return St - Snt


# ---------------------------------------------------------------------------------------------
# DPlda.forward (utils/models.py:479-495)
# ---------------------------------------------------------------------------------------------

def dplda_forward(x1, x2, W1, b1, wlr, blr, dtype=np.float32):
    """LDA + normalize, explicit outer-product features [y1 y2^T + y2 y1^T, y1 y1^T + y2 y2^T, y1 + y2] and a
    single linear unit (utils/models.py:484-495).  wlr: (1, 2 D^2 + D), blr: (1,)."""
    W1, b1 = _as(W1, dtype), _as(b1, dtype)
    y1, _ = normalize(_as(x1, dtype) @ W1.T + b1, dtype)
    y2, _ = normalize(_as(x2, dtype) @ W1.T + b1, dtype)
    return dplda_from_embeddings(y1, y2, wlr, blr, dtype)


def dplda_from_embeddings(y1, y2, wlr, blr, dtype=np.float32):
    y1, y2 = _as(y1, dtype), _as(y2, dtype)
    B = y1.shape[0]
    between = (y1[:, :, None] * y2[:, None, :] + y2[:, :, None] * y1[:, None, :]).reshape(B, -1)
    within = (y1[:, :, None] * y1[:, None, :] + y2[:, :, None] * y2[:, None, :]).reshape(B, -1)
    feats = np.concatenate([between, within, y1 + y2], axis=1)
    return feats @ _as(wlr, dtype).reshape(-1) + dtype(np.asarray(blr).reshape(-1)[0])


def dplda_backward(y1, y2, g, dtype=np.float64):
    """Gradient of sum_i g_i s_i w.r.t. DPlda's linear unit (utils/models.py:484-490): the features are
    [y1 y2^T + y2 y1^T, y1 y1^T + y2 y2^T, y1 + y2], so d wlr = g^T feats and d bias = sum g."""
    y1, y2, g = _as(y1, dtype), _as(y2, dtype), _as(g, dtype)
    B = y1.shape[0]
    between = (y1[:, :, None] * y2[:, None, :] + y2[:, :, None] * y1[:, None, :]).reshape(B, -1)
    within = (y1[:, :, None] * y1[:, None, :] + y2[:, :, None] * y2[:, None, :]).reshape(B, -1)
    feats = np.concatenate([between, within, y1 + y2], axis=1)
    return (g @ feats).reshape(1, -1), np.asarray([g.sum()], dtype=dtype)


def dplda_lda_backward(x1, x2, g, W1, b1, wlr, dtype=np.float64):
    """Gradient of sum_i g_i s_i, s = DPlda.forward(x1, x2) (utils/models.py:492-495), w.r.t. the LDA layer and the
    inputs: dict(W1, b1, x1, x2).  With x = [y1; y2] the score is x^T M x + x^T v + c, M = [[Ww, Wb], [Wb, Ww]],
    v = [ws; ws] (the linear unit's weight split [Wb | Ww | ws] as :484-490 concatenates the features), hence
    dL/dx = g ((M + M^T) x + v), then the F.normalize backward and the LDA's own two GEMMs."""
    W1, b1, g = _as(W1, dtype), _as(b1, dtype), _as(g, dtype)[:, None]
    D1 = W1.shape[0]
    w = _as(wlr, dtype).reshape(-1)
    Wb, Ww, ws = w[:D1 * D1].reshape(D1, D1), w[D1 * D1:2 * D1 * D1].reshape(D1, D1), w[2 * D1 * D1:]
    x1d, x2d = _as(x1, dtype), _as(x2, dtype)
    y1, n1 = normalize(x1d @ W1.T + b1, dtype)
    y2, n2 = normalize(x2d @ W1.T + b1, dtype)
    dy1 = g * (y1 @ (Ww + Ww.T) + y2 @ (Wb + Wb.T) + ws)
    dy2 = g * (y2 @ (Ww + Ww.T) + y1 @ (Wb + Wb.T) + ws)
    du1, du2 = _normalize_bwd(dy1, y1, n1), _normalize_bwd(dy2, y2, n2)
    return dict(W1=du1.T @ x1d + du2.T @ x2d, b1=(du1 + du2).sum(axis=0), x1=du1 @ W1, x2=du2 @ W1)


def weighted_moments(x, w, dtype=np.float64):
    """cnt = sum w, sum = w^T x, sq = x^T diag(w) x — the accumulators of
    xvector_GaussianBackend_pytorch.py:31-52 for one class mask w."""
    x, w = _as(x, dtype), _as(w, dtype)
    return w.sum(), w @ x, (x * w[:, None]).T @ x


def gb_fit(paired, t, dtype=np.float64):
    """Closed-form Gaussian-backend "training" (xvector_GaussianBackend_pytorch.py:30-56), from the paired rows
    x = forward_getpaired(x1, x2) and labels t.  Quirk kept: the non-target class divides BOTH its sum and its
    second moment by (count - 1), the target class by count (:53-56).  Returns mu_t, Lam_t, mu_n, Lam_n."""
    x, t = _as(paired, dtype), _as(t, dtype)
    ct, st, qt = weighted_moments(x, (t > 0.5).astype(dtype), dtype)
    cn, sn, qn = weighted_moments(x, (t < 0.5).astype(dtype), dtype)
    mu_t = st / ct
    Lam_t = np.linalg.inv(qt / ct - np.outer(mu_t, mu_t))
    mu_n = sn / (cn - 1)
    Lam_n = np.linalg.inv(qn / (cn - 1) - np.outer(mu_n, mu_n))
    return mu_t, Lam_t, mu_n, Lam_n


# ---------------------------------------------------------------------------------------------
# Kaldi PLDA -> (P, Q) (utils/Kaldi2NumpyUtils/kaldiPlda2numpydict.py:34-38, utils/models.py:450-457)
# ---------------------------------------------------------------------------------------------

def kaldi_psi_to_pq(psi):
    ac = np.asarray(psi, dtype=np.float64)
    tot = 1 + ac
    diagP = ac / (tot * (tot - ac * ac / tot))
    diagQ = (1 / tot) - 1 / (tot - ac * ac / tot)
    return diagP, diagQ


def kaldi_init_params(transform_mat, mean_vec, plda_mean, diag_transform, psi):
    """utils/models.py:450-457: fold centring into the biases. transform_mat is (D1, D0+1)."""
    T = np.asarray(transform_mat, dtype=np.float64)
    mean_vec = np.asarray(mean_vec, dtype=np.float64)
    Dt = np.asarray(diag_transform, dtype=np.float64)
    diagP, diagQ = kaldi_psi_to_pq(psi)
    W1 = T[:, :-1]
    b1 = T[:, -1] - T[:, :-1].dot(mean_vec)
    W2 = Dt
    b2 = -Dt.dot(np.asarray(plda_mean, dtype=np.float64))
    return Params(*[np.asarray(a, dtype=np.float32) for a in (W1, b1, W2, b2, np.sqrt(diagP), diagQ)])
