"""Thin tensor-level wrappers over the C ABI (include/nplda_hip.h).

Every function takes torch tensors that live on a HIP device, enqueues the kernel on torch's
current stream and returns torch tensors.  No CPU path exists: a CPU tensor is an error here
(neuralplda_amd.models stages CPU inputs through the device, it never computes on the host).
"""
import functools

import torch

from . import _lib


class PackedParams:
    """MFMA-fragment-ordered image of one parameter set (see csrc/nplda_common.h).  precision: "fp32" (exact
    fp32 MFMA, default) or "bf16x3" (split-bf16 image for the opt-in scoring kernels, csrc/nplda_fwd_bf16x3.h)."""

    __slots__ = ("buf", "D0", "D1", "D2", "ldz", "precision")

    def __init__(self, buf, D0, D1, D2, ldz, precision="fp32"):
        self.buf, self.D0, self.D1, self.D2, self.ldz, self.precision = buf, D0, D1, D2, ldz, precision

    @property
    def device(self):
        return self.buf.device


def _require_dev_f32(t, name):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise _lib.NpldaHipError(f"{name} must live on a HIP device (got {t.device}); there is no CPU path")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32 (got {t.dtype})")


def _rows(t, name, D0):
    """Return (tensor, ld) with unit inner stride, 16-byte-aligned rows of at least D0 floats."""
    _require_dev_f32(t, name)
    if t.dim() != 2 or t.shape[1] != D0:
        raise ValueError(f"{name} must have shape (B, {D0}), got {tuple(t.shape)}")
    ok = t.stride(1) == 1 and t.stride(0) >= D0 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0
    if t.shape[0] <= 1:
        ok = ok or t.is_contiguous() and t.data_ptr() % 16 == 0
    if not ok:
        t = t.contiguous()
        if t.data_ptr() % 16 != 0:  # pragma: no cover (torch allocations are >= 256-B aligned)
            t = t.clone()
    ld = t.stride(0) if t.shape[0] > 1 else max(D0, 4)
    return t, ld


def pack_params(W1, b1, W2, b2, P_sqrt, Q, precision="fp32"):
    """nplda_pack_params_f32 (or nplda_pack_params_bf16x3): nn.Linear-layout parameters -> PackedParams."""
    lib = _lib.load()
    if precision not in ("fp32", "bf16x3"):
        raise ValueError("precision must be 'fp32' or 'bf16x3'")
    for n, t in (("W1", W1), ("b1", b1), ("W2", W2), ("b2", b2), ("P_sqrt", P_sqrt), ("Q", Q)):
        _require_dev_f32(t, n)
    D1, D0 = W1.shape
    D2, D1b = W2.shape
    if D1b != D1 or b1.numel() != D1 or b2.numel() != D2 or P_sqrt.numel() != D2 or Q.numel() != D2:
        raise ValueError("inconsistent parameter shapes")
    if D0 % 4 != 0:
        raise ValueError(f"xvector_dim must be a multiple of 4 (got {D0}); pad the x-vectors")
    b3 = precision == "bf16x3"
    nbytes = (lib.nplda_bf16x3_packed_bytes if b3 else lib.nplda_packed_bytes)(D0, D1, D2)
    if nbytes == 0:
        raise _lib.NpldaHipError(
            f"model {D0}->{D1}->{D2} is outside the compiled kernel set (max dim {lib.nplda_max_dim()})")
    buf = torch.empty(nbytes // 4, dtype=torch.float32, device=W1.device)
    ts = [t.detach().contiguous() for t in (W1, b1, W2, b2, P_sqrt, Q)]
    fn = lib.nplda_pack_params_bf16x3 if b3 else lib.nplda_pack_params_f32
    with _lib.on_device(W1.device):
        code = fn(*[_lib.ptr(t) for t in ts], D0, D1, D2, _lib.ptr(buf), nbytes, _lib.current_stream())
    _lib.check(code, "nplda_pack_params_bf16x3" if b3 else "nplda_pack_params_f32")
    return PackedParams(buf, D0, D1, D2, lib.nplda_padded_dim(D1, D2), precision)


def score_pairs(x1, x2, packed):
    """nplda_score_pairs_f32: (B, D0), (B, D0) -> (B,) scores.  Two bfloat16 row batches (an extractor's output) go to
    nplda_score_pairs_bf16rows_f32 where the streaming kernels apply — no fp32 copy of the batch — and are widened otherwise;
    the scores are those of the widened rows either way."""
    lib = _lib.load()
    if (x1.dtype == torch.bfloat16 and x2.dtype == torch.bfloat16 and packed.precision == "fp32" and x1.is_cuda and x2.is_cuda
            and x1.dim() == 2 and x1.shape == x2.shape and x1.shape[1] == packed.D0):
        B = x1.shape[0]
        ok = all(t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 8 == 0 for t in (x1, x2)) and x1.stride(0) == x2.stride(0)
        if ok and B > 0:
            s = torch.empty(B, dtype=torch.float32, device=x1.device)
            with _lib.on_device(x1.device):
                code = lib.nplda_score_pairs_bf16rows_f32(x1.data_ptr(), x2.data_ptr(), B, x1.stride(0), _lib.ptr(packed.buf),
                                                          packed.D0, packed.D1, packed.D2, _lib.ptr(s), _lib.current_stream())
            if code == 0:
                return s
            if code != _lib.NPLDA_EUNSUPPORTED:
                _lib.check(code, "nplda_score_pairs_bf16rows_f32")
    # any remaining non-fp32 floating input (bf16 rows the streaming form did not take — a 'bf16x3' image, short or
    # strided batches —, fp16, fp64) is widened / narrowed here: the kernels below read fp32 rows
    if torch.is_tensor(x1) and x1.is_floating_point() and x1.dtype != torch.float32:
        x1 = x1.float()
    if torch.is_tensor(x2) and x2.is_floating_point() and x2.dtype != torch.float32:
        x2 = x2.float()
    x1, ld1 = _rows(x1, "x1", packed.D0)
    x2, ld2 = _rows(x2, "x2", packed.D0)
    if x1.shape[0] != x2.shape[0]:
        raise ValueError("x1 and x2 must have the same number of rows")
    if ld1 != ld2:
        x1, x2 = x1.contiguous(), x2.contiguous()
        ld1 = ld2 = packed.D0
    B = x1.shape[0]
    s = torch.empty(B, dtype=torch.float32, device=x1.device)
    if B == 0:
        return s
    fn = lib.nplda_score_pairs_bf16x3 if packed.precision == "bf16x3" else lib.nplda_score_pairs_f32
    with _lib.on_device(x1.device):
        code = fn(_lib.ptr(x1), _lib.ptr(x2), B, ld1, _lib.ptr(packed.buf), packed.D0, packed.D1, packed.D2,
                  _lib.ptr(s), _lib.current_stream())
    _lib.check(code, "nplda_score_pairs_" + ("bf16x3" if packed.precision == "bf16x3" else "f32"))
    return s


def score_pairs_rows(table, rows1, rows2, packed):
    """nplda_score_pairs_rows_f32: scores of the pairs (table[rows1], table[rows2]) of a resident (N, D0) x-vector matrix,
    the gather folded into the kernel; falls back to gather_rows + score_pairs where the fused form does not apply."""
    lib = _lib.load()
    if packed.precision != "fp32" or rows1.dtype != torch.int64 or rows2.dtype != torch.int64:
        return score_pairs(gather_rows(table, rows1), gather_rows(table, rows2), packed)
    table, ldt = _rows(table, "table", packed.D0)
    rows1, rows2 = rows1.contiguous(), rows2.contiguous()
    B = rows1.shape[0]
    if rows2.shape[0] != B:
        raise ValueError("rows1 and rows2 must have the same length")
    s = torch.empty(B, dtype=torch.float32, device=table.device)
    if B == 0:
        return s
    with _lib.on_device(table.device):
        code = lib.nplda_score_pairs_rows_f32(_lib.ptr(table), table.shape[0], ldt, _lib.ptr(rows1), _lib.ptr(rows2), B,
                                              _lib.ptr(packed.buf), packed.D0, packed.D1, packed.D2, _lib.ptr(s),
                                              _lib.current_stream())
    if code == -95:  # NPLDA_EUNSUPPORTED: a shape the balanced-tile kernel does not cover
        return score_pairs(gather_rows(table, rows1), gather_rows(table, rows2), packed)
    _lib.check(code, "nplda_score_pairs_rows_f32")
    return s


def embed(x, packed, want_q=True):
    """nplda_embed_f32: (N, D0) -> z table (N, ldz) [columns >= D2 are zero] and q (N,)."""
    lib = _lib.load()
    x, ld = _rows(x, "x", packed.D0)
    N = x.shape[0]
    z = torch.empty((N, packed.ldz), dtype=torch.float32, device=x.device)
    q = torch.empty(N, dtype=torch.float32, device=x.device) if want_q else None
    if N == 0:
        return z, q
    fn = lib.nplda_embed_bf16x3 if packed.precision == "bf16x3" else lib.nplda_embed_f32
    with _lib.on_device(x.device):
        code = fn(_lib.ptr(x), N, ld, _lib.ptr(packed.buf), packed.D0, packed.D1, packed.D2, _lib.ptr(z), packed.ldz,
                  _lib.ptr(q), _lib.current_stream())
    _lib.check(code, "nplda_embed_" + ("bf16x3" if packed.precision == "bf16x3" else "f32"))
    return z, q


def embed_rows(table, rows, packed):
    """nplda_embed_rows_f32: embed(table[rows]) -> (z (U, ldz), q (U,)) with the gather folded into the kernel; gather_rows +
    embed where the fused form does not apply (same values)."""
    lib = _lib.load()
    if packed.precision != "fp32" or rows.dtype != torch.int64:
        return embed(gather_rows(table, rows), packed)
    table, ldt = _rows(table, "table", packed.D0)
    rows = rows.contiguous()
    U = rows.shape[0]
    z = torch.empty((U, packed.ldz), dtype=torch.float32, device=table.device)
    q = torch.empty(U, dtype=torch.float32, device=table.device)
    if U == 0:
        return z, q
    with _lib.on_device(table.device):
        code = lib.nplda_embed_rows_f32(_lib.ptr(table), table.shape[0], ldt, _lib.ptr(rows), U, _lib.ptr(packed.buf),
                                        packed.D0, packed.D1, packed.D2, _lib.ptr(z), packed.ldz, _lib.ptr(q),
                                        _lib.current_stream())
    if code == _lib.NPLDA_EUNSUPPORTED:
        return embed(gather_rows(table, rows), packed)
    _lib.check(code, "nplda_embed_rows_f32")
    return z, q


def embed_pair(xa, xb, packed):
    """nplda_embed_pair_f32: the z tables and q vectors of two row sets, ((za, qa), (zb, qb)) — views of ONE (Na + Nb, ldz)
    table filled by one launch where the balanced-tile kernel applies (same values as two embed() calls)."""
    lib = _lib.load()
    if packed.precision != "fp32":
        return embed(xa, packed), embed(xb, packed)
    xa, lda = _rows(xa, "xa", packed.D0)
    xb, ldb = _rows(xb, "xb", packed.D0)
    if lda != ldb or xa.device != xb.device:
        return embed(xa, packed), embed(xb, packed)
    Na, Nb = xa.shape[0], xb.shape[0]
    z = torch.empty((Na + Nb, packed.ldz), dtype=torch.float32, device=xa.device)
    q = torch.empty(Na + Nb, dtype=torch.float32, device=xa.device)
    if Na + Nb > 0:
        with _lib.on_device(xa.device):
            code = lib.nplda_embed_pair_f32(_lib.ptr(xa), Na, _lib.ptr(xb), Nb, lda, _lib.ptr(packed.buf), packed.D0, packed.D1,
                                            packed.D2, _lib.ptr(z), packed.ldz, _lib.ptr(q), _lib.current_stream())
        _lib.check(code, "nplda_embed_pair_f32")
    return (z[:Na], q[:Na]), (z[Na:], q[Na:])


# ---- training ---------------------------------------------------------------------------------

LOSS_SOFTCDET, LOSS_BCE, LOSS_HARD_CDET = 0, 1, 2


def _need_fp32(packed, what):
    if packed.precision != "fp32":
        raise ValueError(f"{what} needs the fp32 parameter image (pack_params(..., precision='fp32'))")


def forward_train(x1, x2, packed):
    """nplda_forward_train_f32: scores plus the activations the backward needs.
    Returns (s, saved) with saved = (x1, x2, ld, y, z, rn) device tensors."""
    lib = _lib.load()
    _need_fp32(packed, "forward_train")
    x1, ld1 = _rows(x1, "x1", packed.D0)
    x2, ld2 = _rows(x2, "x2", packed.D0)
    if x1.shape[0] != x2.shape[0]:
        raise ValueError("x1 and x2 must have the same number of rows")
    if ld1 != ld2:
        x1, x2 = x1.contiguous(), x2.contiguous()
        ld1 = ld2 = packed.D0
    B = x1.shape[0]
    dev = x1.device
    s = torch.empty(B, dtype=torch.float32, device=dev)
    y = torch.empty((2 * B, packed.ldz), dtype=torch.float32, device=dev)
    z = torch.empty((2 * B, packed.ldz), dtype=torch.float32, device=dev)
    rn = torch.empty(2 * B, dtype=torch.float32, device=dev)
    if B > 0:
        with _lib.on_device(dev):
            code = lib.nplda_forward_train_f32(_lib.ptr(x1), _lib.ptr(x2), B, ld1, _lib.ptr(packed.buf), packed.D0,
                                               packed.D1, packed.D2, _lib.ptr(s), _lib.ptr(y), _lib.ptr(z),
                                               _lib.ptr(rn), packed.ldz, _lib.current_stream())
        _lib.check(code, "nplda_forward_train_f32")
    return s, (x1, x2, ld1, y, z, rn)


@functools.lru_cache(maxsize=64)
def _backward_sizes(rows, D0, D1, D2, want_dx):
    lib = _lib.load()
    return (lib.nplda_grad_floats(D0, D1, D2), lib.nplda_backward_ex_workspace_bytes(rows, D0, D1, D2, want_dx))


def backward(saved, g, packed, P_sqrt, want_dx=False):
    """nplda_backward_ex_f32: flat gradient [dW1 | db1 | dW2 | db2 | dP_sqrt | dQ] for dL/ds = g; with want_dx also
    the input gradients -> (flat, dx1, dx2), dx (B, D0) = du . W1."""
    lib = _lib.load()
    x1, x2, ld, y, z, rn = saved
    B = x1.shape[0]
    dev = x1.device
    _require_dev_f32(g, "g")
    g = g.contiguous()
    n, wsb = _backward_sizes(2 * B, packed.D0, packed.D1, packed.D2, 1 if want_dx else 0)
    flat = torch.empty(n, dtype=torch.float32, device=dev)
    ws = torch.empty(max(wsb // 4, 4), dtype=torch.float32, device=dev)
    ps = P_sqrt.detach().contiguous()
    dx1 = torch.empty((B, packed.D0), dtype=torch.float32, device=dev) if want_dx else None
    dx2 = torch.empty((B, packed.D0), dtype=torch.float32, device=dev) if want_dx else None
    with _lib.on_device(dev):
        code = lib.nplda_backward_ex_f32(_lib.ptr(x1), _lib.ptr(x2), B, ld, _lib.ptr(packed.buf), packed.D0, packed.D1,
                                         packed.D2, _lib.ptr(g), _lib.ptr(y), _lib.ptr(z), _lib.ptr(rn), packed.ldz,
                                         _lib.ptr(ps), _lib.ptr(ws), wsb, _lib.ptr(flat), _lib.ptr(dx1), _lib.ptr(dx2),
                                         packed.D0, _lib.current_stream())
    _lib.check(code, "nplda_backward_ex_f32")
    return (flat, dx1, dx2) if want_dx else flat


def embed_train(x, packed):
    """nplda_embed_train_f32: (N, D0) -> z (N, ldz) plus the saved rows (x, ld, y (N, ldz), rn (N)) of its backward."""
    lib = _lib.load()
    _need_fp32(packed, "embed_train")
    x, ld = _rows(x, "x", packed.D0)
    N = x.shape[0]
    dev = x.device
    z = torch.empty((N, packed.ldz), dtype=torch.float32, device=dev)
    y = torch.empty((N, packed.ldz), dtype=torch.float32, device=dev)
    rn = torch.empty(N, dtype=torch.float32, device=dev)
    if N > 0:
        with _lib.on_device(dev):
            code = lib.nplda_embed_train_f32(_lib.ptr(x), N, ld, _lib.ptr(packed.buf), packed.D0, packed.D1, packed.D2,
                                             _lib.ptr(z), _lib.ptr(y), _lib.ptr(rn), packed.ldz, _lib.current_stream())
        _lib.check(code, "nplda_embed_train_f32")
    return z, (x, ld, y, rn)


def embed_backward(saved, gz, packed, want_dx=False):
    """nplda_embed_backward_f32: gz = dL/dz (N, D2) -> (flat gradient with dP_sqrt = dQ = 0, dx (N, D0) or None)."""
    lib = _lib.load()
    x, ld, y, rn = saved
    N = x.shape[0]
    dev = x.device
    _require_dev_f32(gz, "gz")
    if gz.dim() != 2 or gz.shape != (N, packed.D2):
        raise ValueError(f"gz must be ({N}, {packed.D2})")
    if gz.stride(1) != 1 or (N > 1 and gz.stride(0) < packed.D2):
        gz = gz.contiguous()
    n = lib.nplda_grad_floats(packed.D0, packed.D1, packed.D2)
    flat = torch.empty(n, dtype=torch.float32, device=dev)
    wsb = lib.nplda_backward_ex_workspace_bytes(N, packed.D0, packed.D1, packed.D2, 1 if want_dx else 0)
    ws = torch.empty(max(wsb // 4, 4), dtype=torch.float32, device=dev)
    dx = torch.empty((N, packed.D0), dtype=torch.float32, device=dev) if want_dx else None
    with _lib.on_device(dev):
        code = lib.nplda_embed_backward_f32(_lib.ptr(x), N, ld, _lib.ptr(packed.buf), packed.D0, packed.D1, packed.D2,
                                            _lib.ptr(gz), gz.stride(0) if N > 1 else packed.D2, _lib.ptr(y),
                                            _lib.ptr(rn), packed.ldz, _lib.ptr(ws), wsb, _lib.ptr(flat), _lib.ptr(dx),
                                            packed.D0, _lib.current_stream())
    _lib.check(code, "nplda_embed_backward_f32")
    return flat, dx


def score_embeddings_bwd(z1, z2, P_sqrt, Q, g, want_dz1=True, want_dz2=True):
    """nplda_score_embeddings_bwd_f32 -> (dz1 or None, dz2 or None, dP_sqrt (D2), dQ (D2))."""
    lib = _lib.load()
    for n, t in (("z1", z1), ("z2", z2), ("P_sqrt", P_sqrt), ("Q", Q), ("g", g)):
        _require_dev_f32(t, n)
    D2 = Q.numel()
    if z1.stride(1) != 1:
        z1 = z1.contiguous()
    if z2.stride(1) != 1:
        z2 = z2.contiguous()
    B = z1.shape[0]
    dev = z1.device
    dz1 = torch.empty((B, D2), dtype=torch.float32, device=dev) if want_dz1 else None
    dz2 = torch.empty((B, D2), dtype=torch.float32, device=dev) if want_dz2 else None
    dP = torch.empty(D2, dtype=torch.float32, device=dev)
    dQ = torch.empty(D2, dtype=torch.float32, device=dev)
    wsb = lib.nplda_score_embeddings_bwd_workspace_bytes(B, D2)
    if wsb == 0:
        raise _lib.NpldaHipError(f"embedding dimension {D2} is outside the compiled kernel set")
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=dev)
    ld1 = z1.stride(0) if B > 1 else D2
    ld2 = z2.stride(0) if B > 1 else D2
    with _lib.on_device(dev):
        code = lib.nplda_score_embeddings_bwd_f32(_lib.ptr(z1), ld1, _lib.ptr(z2), ld2, B, D2,
                                                  _lib.ptr(P_sqrt.contiguous()), _lib.ptr(Q.contiguous()),
                                                  _lib.ptr(g.contiguous()), _lib.ptr(dz1), D2, _lib.ptr(dz2), D2,
                                                  _lib.ptr(dP), _lib.ptr(dQ), _lib.ptr(ws), wsb, _lib.current_stream())
    _lib.check(code, "nplda_score_embeddings_bwd_f32")
    return dz1, dz2, dP, dQ


def pack_matrix(src, mode=0):
    """nplda_pack_matrix_f32: fragment image of Wm = src (mode 0), src^T (1) or src + src^T (2) -> (frag, K, N)."""
    lib = _lib.load()
    _require_dev_f32(src, "src")
    if src.dim() != 2 or src.stride(1) != 1:
        raise ValueError("src must be a 2-D tensor with unit inner stride")
    K, N = (src.shape[1], src.shape[0]) if mode == 1 else (src.shape[0], src.shape[1])
    nbytes = lib.nplda_matrix_frag_bytes(K, N)
    if nbytes == 0:
        raise _lib.NpldaHipError(f"a {K} x {N} matrix is outside the resident-matrix GEMM (K <= 512, N % 4 == 0)")
    frag = torch.empty(nbytes // 4, dtype=torch.float32, device=src.device)
    with _lib.on_device(src.device):
        code = lib.nplda_pack_matrix_f32(_lib.ptr(src), src.stride(0), K, N, mode, _lib.ptr(frag), nbytes,
                                         _lib.current_stream())
    _lib.check(code, "nplda_pack_matrix_f32")
    return frag, K, N


def rows_matmul(rows, packed_matrix, bias=None, rowscale=None):
    """nplda_rows_matmul_f32: out[r] = rowscale[r] * (rows[r, :K] . Wm + bias) -> (R, N)."""
    lib = _lib.load()
    frag, K, N = packed_matrix
    rows, ld = _rows(rows, "rows", K)
    R = rows.shape[0]
    out = torch.empty((R, N), dtype=torch.float32, device=rows.device)
    if R == 0:
        return out
    with _lib.on_device(rows.device):
        code = lib.nplda_rows_matmul_f32(_lib.ptr(rows), ld, R, K, _lib.ptr(frag), N,
                                         _lib.ptr(bias.contiguous()) if bias is not None else None,
                                         _lib.ptr(rowscale.contiguous()) if rowscale is not None else None,
                                         _lib.ptr(out), N, _lib.current_stream())
    _lib.check(code, "nplda_rows_matmul_f32")
    return out


def lda_backward(x1, x2, paired, rn, dpaired, W1, want_w=True, want_dx=True):
    """Backward through y = normalize(LDA x) of a paired-row head (DPlda): dpaired = dL/d[y1 | y2] (B, 2 D1) ->
    (dW1 (D1, D0), db1 (D1), dx1, dx2): F.normalize backward on the paired rows, then the wgrad / dgrad GEMMs."""
    lib = _lib.load()
    D1, D0 = W1.shape
    x1, ld1 = _rows(x1, "x1", D0)
    x2, ld2 = _rows(x2, "x2", D0)
    if ld1 != ld2:
        x1, x2 = x1.contiguous(), x2.contiguous()
        ld1 = D0
    B = x1.shape[0]
    dev = x1.device
    Mp = lib.nplda_padded_dim(D1, D1)
    du = torch.empty((2 * B, Mp), dtype=torch.float32, device=dev)
    dW1 = db1 = dx1 = dx2 = None
    st = _lib.current_stream()
    with _lib.on_device(dev):
        _lib.check(lib.nplda_normalize_bwd_paired_f32(_lib.ptr(dpaired), dpaired.stride(0), _lib.ptr(paired),
                                                      paired.stride(0), _lib.ptr(rn), B, D1, _lib.ptr(du), Mp, st),
                   "nplda_normalize_bwd_paired_f32")
        if want_w:
            wsb = lib.nplda_lda_wgrad_workspace_bytes(B, D0, D1)
            ws = torch.empty(max(wsb // 4, 4), dtype=torch.float32, device=dev)
            out = torch.empty(D1 * D0 + D1, dtype=torch.float32, device=dev)
            _lib.check(lib.nplda_lda_wgrad_f32(_lib.ptr(x1), _lib.ptr(x2), B, ld1, _lib.ptr(du), Mp, D0, D1,
                                               _lib.ptr(ws), wsb, _lib.ptr(out), st), "nplda_lda_wgrad_f32")
            dW1, db1 = out[:D1 * D0].view(D1, D0), out[D1 * D0:]
        if want_dx:
            wsb = lib.nplda_lda_dgrad_workspace_bytes(D0, D1)
            ws = torch.empty(max(wsb // 4, 4), dtype=torch.float32, device=dev)
            dx1 = torch.empty((B, D0), dtype=torch.float32, device=dev)
            dx2 = torch.empty((B, D0), dtype=torch.float32, device=dev)
            _lib.check(lib.nplda_lda_dgrad_f32(_lib.ptr(du), Mp, B, _lib.ptr(W1.detach().contiguous()), D0, D1,
                                               _lib.ptr(ws), wsb, _lib.ptr(dx1), _lib.ptr(dx2), D0, st),
                       "nplda_lda_dgrad_f32")
    return dW1, db1, dx1, dx2


def split_flat_grad(flat, D0, D1, D2):
    """Views (dW1, db1, dW2, db2, dP_sqrt, dQ) into the flat gradient buffer."""
    w1, b1, w2, b2, ps, q = flat.split_with_sizes((D1 * D0, D1, D2 * D1, D2, D2, D2))  # (one call: six views)
    return w1.view(D1, D0), b1, w2.view(D2, D1), b2, ps, q


def _theta_array(thetas):
    import ctypes
    arr = (ctypes.c_void_p * len(thetas))()
    for i, t in enumerate(thetas):
        _require_dev_f32(t, "theta")
        arr[i] = t.data_ptr()
    return arr


def loss_sums(s, t, thetas, alpha, kind):
    """nplda_loss_sums_f32: fp64 vector of batch-global sums (additive across shards)."""
    lib = _lib.load()
    _require_dev_f32(s, "output")
    _require_dev_f32(t, "target")
    s, t = s.contiguous(), t.contiguous()
    if s.shape != t.shape or s.dim() != 1:
        raise ValueError("output and target must be 1-D tensors of the same length")
    K = len(thetas)
    ns = lib.nplda_loss_nsums(K, kind)
    if ns == 0:
        raise _lib.NpldaHipError(f"loss with {K} thresholds is not supported by the compiled kernels (max 4)")
    sums = torch.empty(ns, dtype=torch.float64, device=s.device)
    with _lib.on_device(s.device):
        code = lib.nplda_loss_sums_f32(_lib.ptr(s), _lib.ptr(t), s.shape[0], _theta_array(thetas), K, float(alpha),
                                       kind, _lib.ptr(sums), _lib.current_stream())
    _lib.check(code, "nplda_loss_sums_f32")
    return sums


def loss_finish(s, t, thetas, betas, alpha, kind, sums, want_grad=True):
    """nplda_loss_finish_f32: (loss 0-d tensor, g or None, dtheta (K,) or None)."""
    import ctypes
    lib = _lib.load()
    s, t = s.contiguous(), t.contiguous()
    K = len(thetas)
    dev = s.device
    loss = torch.empty((), dtype=torch.float32, device=dev)
    g = torch.empty_like(s) if want_grad else None
    dth = torch.empty(K, dtype=torch.float32, device=dev) if want_grad else None
    barr = (ctypes.c_float * max(K, 1))(*[float(b) for b in betas]) if kind != LOSS_BCE else None
    with _lib.on_device(dev):
        code = lib.nplda_loss_finish_f32(_lib.ptr(s), _lib.ptr(t), s.shape[0], _theta_array(thetas), barr, K,
                                         float(alpha), kind, _lib.ptr(sums), _lib.ptr(loss), _lib.ptr(g),
                                         _lib.ptr(dth), _lib.current_stream())
    _lib.check(code, "nplda_loss_finish_f32")
    return loss, g, dth


def loss_fwd_bwd(s, t, thetas, betas, alpha, kind, want_joint=False):
    """nplda_loss_fwd_bwd_f32: (loss 0-d tensor, g, dtheta (K,), sums[, the buffer g and dtheta live in]) of an unsharded batch — both loss passes in one
    call (one launch up to 4096 pairs), bit-identical to loss_sums + loss_finish."""
    import ctypes
    lib = _lib.load()
    _require_dev_f32(s, "output")
    _require_dev_f32(t, "target")
    s, t = s.contiguous(), t.contiguous()
    if s.shape != t.shape or s.dim() != 1:
        raise ValueError("output and target must be 1-D tensors of the same length")
    K = len(thetas)
    ns = lib.nplda_loss_nsums(K, kind)
    if ns == 0 or kind == LOSS_HARD_CDET:
        raise _lib.NpldaHipError("loss_fwd_bwd: unsupported loss kind / number of thresholds")
    dev = s.device
    sums = torch.empty(ns, dtype=torch.float64, device=dev)
    loss = torch.empty((), dtype=torch.float32, device=dev)
    # g and dtheta share a buffer ([g (B) | pad to 4 | dtheta (K)]: loss_joint_views): the backward scales both by the
    # incoming dL/dloss with ONE product
    B = s.shape[0]
    joint = torch.empty(((B + 3) & ~3) + K, dtype=torch.float32, device=dev)
    g, dth = loss_joint_views(joint, B, K)
    barr = (ctypes.c_float * max(K, 1))(*[float(b) for b in betas]) if kind != LOSS_BCE else None
    with _lib.on_device(dev):
        code = lib.nplda_loss_fwd_bwd_f32(_lib.ptr(s), _lib.ptr(t), s.shape[0], _theta_array(thetas), barr, K,
                                          float(alpha), kind, _lib.ptr(sums), _lib.ptr(loss), _lib.ptr(g),
                                          _lib.ptr(dth), _lib.current_stream())
    _lib.check(code, "nplda_loss_fwd_bwd_f32")
    return (loss, g, dth, sums, joint) if want_joint else (loss, g, dth, sums)


def loss_joint_views(joint, B, K):
    """(g, dtheta) inside loss_fwd_bwd's joint buffer (or a product of it)."""
    o = (B + 3) & ~3
    return joint[:B], joint[o:o + K]


def pack_params_into(packed, W1, b1, W2, b2, P_sqrt, Q):
    """nplda_pack_params_f32 into an EXISTING PackedParams buffer (its address is baked into captured graphs)."""
    lib = _lib.load()
    _need_fp32(packed, "pack_params_into")
    ts = [t.detach().contiguous() for t in (W1, b1, W2, b2, P_sqrt, Q)]
    for n, t in zip(("W1", "b1", "W2", "b2", "P_sqrt", "Q"), ts):
        _require_dev_f32(t, n)
    with _lib.on_device(packed.buf.device):
        code = lib.nplda_pack_params_f32(*[_lib.ptr(t) for t in ts], packed.D0, packed.D1, packed.D2, _lib.ptr(packed.buf),
                                         packed.buf.numel() * 4, _lib.current_stream())
    _lib.check(code, "nplda_pack_params_f32")
    return packed


def repack_params(packed, keys):
    """The image of `packed` refreshed from six parameter tensors that were validated when it was first packed (same
    storage, same device: `keys` = their (data_ptr, version, device) triples) — the raw launch, nothing else."""
    lib = _lib.load()
    b3 = packed.precision == "bf16x3"
    fn = lib.nplda_pack_params_bf16x3 if b3 else lib.nplda_pack_params_f32
    dev = keys[0][2]
    with _lib.on_device(dev):
        code = fn(keys[0][0], keys[1][0], keys[2][0], keys[3][0], keys[4][0], keys[5][0], packed.D0, packed.D1, packed.D2,
                  packed.buf.data_ptr(), packed.buf.numel() * 4, _lib.current_stream())
    _lib.check(code, "nplda_pack_params_bf16x3" if b3 else "nplda_pack_params_f32")
    return packed


def train_step_workspace(B, packed, rows=False):
    """Workspace tensor for train_step (rows=True: train_step_rows) at batch size B (None when the fused step does not
    cover this size / shape)."""
    lib = _lib.load()
    fn = lib.nplda_train_step_rows_workspace_bytes if rows else lib.nplda_train_step_workspace_bytes
    n = fn(B, packed.D0, packed.D1, packed.D2)
    if n == 0:
        return None
    return torch.empty(n // 4, dtype=torch.float32, device=packed.buf.device)


def train_step(x1, x2, target, params, thetas, betas, alpha, kind, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps,
               weight_decay, packed, ws, loss, grad_out=None, loss_sum=None):
    """nplda_train_step_f32: forward -> loss -> backward -> Adam on `params` (the six parameter tensors, updated IN
    PLACE) and `thetas`, `packed` refreshed to the updated parameters, `loss` (0-d device tensor) written; `loss_sum` (optional 1-element fp64 device tensor) += loss.  Three launches."""
    import ctypes
    lib = _lib.load()
    _need_fp32(packed, "train_step")
    x1, ld1 = _rows(x1, "x1", packed.D0)
    x2, ld2 = _rows(x2, "x2", packed.D0)
    if x1.shape[0] != x2.shape[0] or target.shape[0] != x1.shape[0]:
        raise ValueError("x1, x2 and target must have the same number of rows")
    if ld1 != ld2:
        x1, x2 = x1.contiguous(), x2.contiguous()
        ld1 = ld2 = packed.D0
    _require_dev_f32(target, "target")
    target = target.contiguous()
    if target.data_ptr() % 16:
        target = target.clone()
    for q in list(params) + list(thetas):
        _require_dev_f32(q, "parameter")
        if not q.is_contiguous():
            raise ValueError("train_step updates the parameter tensors in place: they must be contiguous")
    K = len(thetas)
    parr = (ctypes.c_void_p * 6)(*[q.data_ptr() for q in params])
    barr = (ctypes.c_float * max(K, 1))(*[float(b) for b in betas]) if kind != LOSS_BCE else None
    with _lib.on_device(x1.device):
        code = lib.nplda_train_step_f32(_lib.ptr(x1), _lib.ptr(x2), x1.shape[0], ld1, _lib.ptr(target), parr, packed.D0,
                                        packed.D1, packed.D2, _theta_array(thetas), barr, K, float(alpha), kind,
                                        _lib.ptr(exp_avg), _lib.ptr(exp_avg_sq), _lib.ptr(step), float(lr), float(beta1),
                                        float(beta2), float(eps), float(weight_decay), _lib.ptr(packed.buf), _lib.ptr(ws),
                                        ws.numel() * 4, _lib.ptr(loss), _lib.ptr(loss_sum) if loss_sum is not None else None,
                                        _lib.ptr(grad_out) if grad_out is not None else None, _lib.current_stream())
    _lib.check(code, "nplda_train_step_f32")
    return loss


def train_step_flat_floats(packed):
    """Length of the data-parallel step's flat buffer: the flat gradient + 4 x 18 floats of loss sums (16-bit limbs)."""
    return int(_lib.load().nplda_train_step_flat_floats(packed.D0, packed.D1, packed.D2))


def train_step_grad(x1, x2, target, params, thetas, betas, alpha, kind, step, packed, ws, flat, global_counts=None):
    """nplda_train_step_grad_f32: this rank's share of a data-parallel step up to the flat gradient (+ loss sums) in
    `flat`; `global_counts` = device float64 [N_t, N_n] of the GLOBAL minibatch (None: one rank).  Follow with a SUM
    all-reduce of `flat` and train_step_apply."""
    import ctypes
    lib = _lib.load()
    _need_fp32(packed, "train_step_grad")
    x1, ld1 = _rows(x1, "x1", packed.D0)
    x2, ld2 = _rows(x2, "x2", packed.D0)
    if x1.shape[0] != x2.shape[0] or target.shape[0] != x1.shape[0]:
        raise ValueError("x1, x2 and target must have the same number of rows")
    if ld1 != ld2:
        x1, x2 = x1.contiguous(), x2.contiguous()
        ld1 = ld2 = packed.D0
    _require_dev_f32(target, "target")
    target = target.contiguous()
    if target.data_ptr() % 16:
        target = target.clone()
    if global_counts is not None and (global_counts.dtype != torch.float64 or not global_counts.is_cuda
                                      or global_counts.numel() != 2 or not global_counts.is_contiguous()):
        raise ValueError("global_counts must be a contiguous device float64 tensor [N_t, N_n]")
    if flat.dtype != torch.float32 or not flat.is_cuda or flat.numel() < train_step_flat_floats(packed) or not flat.is_contiguous():
        raise ValueError("flat must be a contiguous device float32 tensor of train_step_flat_floats(packed) elements")
    K = len(thetas)
    parr = (ctypes.c_void_p * 6)(*[q.data_ptr() for q in params])
    barr = (ctypes.c_float * max(K, 1))(*[float(b) for b in betas]) if kind != LOSS_BCE else None
    with _lib.on_device(x1.device):
        code = lib.nplda_train_step_grad_f32(_lib.ptr(x1), _lib.ptr(x2), x1.shape[0], ld1, _lib.ptr(target),
                                             _lib.ptr(global_counts) if global_counts is not None else None, parr,
                                             packed.D0, packed.D1, packed.D2, _theta_array(thetas), barr, K, float(alpha),
                                             kind, _lib.ptr(step), _lib.ptr(packed.buf), _lib.ptr(ws), ws.numel() * 4,
                                             _lib.ptr(flat), _lib.current_stream())
    _lib.check(code, "nplda_train_step_grad_f32")
    return flat


def train_step_grad_rows(table, rows1, rows2, target, params, thetas, betas, alpha, kind, step, packed, ws, flat,
                         global_counts=None):
    """nplda_train_step_grad_rows_f32: train_step_grad on the pairs (table[rows1], table[rows2]); the first kernel gathers
    the rows itself.  `ws`: train_step_workspace(B, packed, rows=True)."""
    import ctypes
    lib = _lib.load()
    _need_fp32(packed, "train_step_grad_rows")
    table, ldt = _rows(table, "table", packed.D0)
    dev = table.device
    if rows1.dtype != torch.int64 or rows2.dtype != torch.int64 or rows1.device != dev or rows2.device != dev:
        raise TypeError("rows must be int64 tensors on the table's device")
    rows1, rows2 = rows1.contiguous(), rows2.contiguous()
    B = rows1.shape[0]
    if rows2.shape[0] != B or target.shape[0] != B:
        raise ValueError("rows1, rows2 and target must have the same length")
    _require_dev_f32(target, "target")
    target = target.contiguous()
    if target.data_ptr() % 16:
        target = target.clone()
    if global_counts is not None and (global_counts.dtype != torch.float64 or not global_counts.is_cuda
                                      or global_counts.numel() != 2 or not global_counts.is_contiguous()):
        raise ValueError("global_counts must be a contiguous device float64 tensor [N_t, N_n]")
    if flat.dtype != torch.float32 or not flat.is_cuda or flat.numel() < train_step_flat_floats(packed) or not flat.is_contiguous():
        raise ValueError("flat must be a contiguous device float32 tensor of train_step_flat_floats(packed) elements")
    K = len(thetas)
    parr = (ctypes.c_void_p * 6)(*[q.data_ptr() for q in params])
    barr = (ctypes.c_float * max(K, 1))(*[float(b) for b in betas]) if kind != LOSS_BCE else None
    with _lib.on_device(dev):
        code = lib.nplda_train_step_grad_rows_f32(_lib.ptr(table), table.shape[0], ldt, _lib.ptr(rows1), _lib.ptr(rows2), B,
                                                  _lib.ptr(target), _lib.ptr(global_counts) if global_counts is not None else None,
                                                  parr, packed.D0, packed.D1, packed.D2, _theta_array(thetas), barr, K,
                                                  float(alpha), kind, _lib.ptr(step), _lib.ptr(packed.buf), _lib.ptr(ws),
                                                  ws.numel() * 4, _lib.ptr(flat), _lib.current_stream())
    _lib.check(code, "nplda_train_step_grad_rows_f32")
    return flat


def train_step_grad_dx(x1, x2, target, params, thetas, betas, alpha, kind, step, packed, ws, flat, dx1, dx2,
                       global_counts=None):
    """nplda_train_step_grad_dx_f32: train_step_grad that also writes dL/dx1, dL/dx2 (float32 or bfloat16, as x1 / x2) — the
    data-parallel form of train_step_dx's first half.  `ws`: train_step_dx_workspace."""
    import ctypes
    lib = _lib.load()
    _need_fp32(packed, "train_step_grad_dx")
    bf = x1.dtype == torch.bfloat16
    for t in (x1, x2, dx1, dx2):
        if t.dtype != x1.dtype or t.dtype not in (torch.float32, torch.bfloat16) or not t.is_cuda:
            raise ValueError("train_step_grad_dx: x1, x2, dx1, dx2 must be device tensors, all float32 or all bfloat16")
        if t.dim() != 2 or t.shape[1] != packed.D0 or t.stride(1) != 1 or t.stride(0) % 4 or t.data_ptr() % 16:
            raise ValueError("train_step_grad_dx: (B, D0) rows with unit inner stride, 16-byte aligned")
    B = x1.shape[0]
    if x2.shape[0] != B or target.shape[0] != B or dx1.shape[0] != B or dx2.shape[0] != B:
        raise ValueError("x1, x2, target, dx1, dx2 must have the same number of rows")
    if x1.stride(0) != x2.stride(0) or dx1.stride(0) != dx2.stride(0):
        raise ValueError("train_step_grad_dx: x1 / x2 (and dx1 / dx2) must share their row stride")
    _require_dev_f32(target, "target")
    if not target.is_contiguous() or target.data_ptr() % 16:
        raise ValueError("train_step_grad_dx: target must be contiguous and 16-byte aligned")
    if global_counts is not None and (global_counts.dtype != torch.float64 or not global_counts.is_cuda
                                      or global_counts.numel() != 2 or not global_counts.is_contiguous()):
        raise ValueError("global_counts must be a contiguous device float64 tensor [N_t, N_n]")
    if flat.dtype != torch.float32 or not flat.is_cuda or flat.numel() < train_step_flat_floats(packed) or not flat.is_contiguous():
        raise ValueError("flat must be a contiguous device float32 tensor of train_step_flat_floats(packed) elements")
    K = len(thetas)
    parr = (ctypes.c_void_p * 6)(*[q.data_ptr() for q in params])
    barr = (ctypes.c_float * max(K, 1))(*[float(b) for b in betas]) if kind != LOSS_BCE else None
    with _lib.on_device(x1.device):
        code = lib.nplda_train_step_grad_dx_f32(x1.data_ptr(), x2.data_ptr(), B, x1.stride(0), 1 if bf else 0, _lib.ptr(target),
                                                _lib.ptr(global_counts) if global_counts is not None else None, parr,
                                                packed.D0, packed.D1, packed.D2, _theta_array(thetas), barr, K, float(alpha),
                                                kind, _lib.ptr(step), _lib.ptr(packed.buf), _lib.ptr(ws), ws.numel() * 4,
                                                _lib.ptr(flat), dx1.data_ptr(), dx2.data_ptr(), dx1.stride(0),
                                                _lib.current_stream())
    _lib.check(code, "nplda_train_step_grad_dx_f32")
    return flat


def train_step_apply(flat, params, thetas, betas, alpha, kind, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps,
                     weight_decay, packed, loss, loss_sum=None):
    """nplda_train_step_apply_f32: the update half of the data-parallel step from the all-reduced `flat`."""
    import ctypes
    lib = _lib.load()
    _need_fp32(packed, "train_step_apply")
    K = len(thetas)
    parr = (ctypes.c_void_p * 6)(*[q.data_ptr() for q in params])
    barr = (ctypes.c_float * max(K, 1))(*[float(b) for b in betas]) if kind != LOSS_BCE else None
    with _lib.on_device(flat.device):
        code = lib.nplda_train_step_apply_f32(_lib.ptr(flat), parr, packed.D0, packed.D1, packed.D2, _theta_array(thetas),
                                              barr, K, float(alpha), kind, _lib.ptr(exp_avg), _lib.ptr(exp_avg_sq),
                                              _lib.ptr(step), float(lr), float(beta1), float(beta2), float(eps),
                                              float(weight_decay), _lib.ptr(packed.buf), _lib.ptr(loss),
                                              _lib.ptr(loss_sum) if loss_sum is not None else None, _lib.current_stream())
    _lib.check(code, "nplda_train_step_apply_f32")
    return loss


def train_step_dx(x1, x2, target, params, thetas, betas, alpha, kind, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps,
                  weight_decay, packed, ws, loss, dx1, dx2, loss_sum=None):
    """nplda_train_step_dx_f32: train_step that also writes dL/dx1, dL/dx2 into `dx1`, `dx2` (B, D0).  x1 / x2 and dx1 / dx2
    are all float32 or all bfloat16 (the head's arithmetic is fp32 either way).  Four launches.  `ws`: a float32 tensor of
    nplda_train_step_dx_workspace_bytes (train_step_dx_workspace)."""
    import ctypes
    lib = _lib.load()
    _need_fp32(packed, "train_step_dx")
    bf = x1.dtype == torch.bfloat16
    for t in (x1, x2, dx1, dx2):
        if t.dtype != x1.dtype or t.dtype not in (torch.float32, torch.bfloat16) or not t.is_cuda:
            raise ValueError("train_step_dx: x1, x2, dx1, dx2 must be device tensors, all float32 or all bfloat16")
        if t.dim() != 2 or t.shape[1] != packed.D0 or t.stride(1) != 1 or t.stride(0) % 4 or t.data_ptr() % 16:
            raise ValueError("train_step_dx: (B, D0) rows with unit inner stride, 16-byte aligned")
    B = x1.shape[0]
    if x2.shape[0] != B or target.shape[0] != B or dx1.shape[0] != B or dx2.shape[0] != B:
        raise ValueError("x1, x2, target, dx1, dx2 must have the same number of rows")
    if x1.stride(0) != x2.stride(0) or dx1.stride(0) != dx2.stride(0):
        raise ValueError("train_step_dx: x1 / x2 (and dx1 / dx2) must share their row stride")
    _require_dev_f32(target, "target")
    if not target.is_contiguous() or target.data_ptr() % 16:
        raise ValueError("train_step_dx: target must be contiguous and 16-byte aligned")
    for q in list(params) + list(thetas):
        _require_dev_f32(q, "parameter")
        if not q.is_contiguous():
            raise ValueError("train_step_dx updates the parameter tensors in place: they must be contiguous")
    K = len(thetas)
    parr = (ctypes.c_void_p * 6)(*[q.data_ptr() for q in params])
    barr = (ctypes.c_float * max(K, 1))(*[float(b) for b in betas]) if kind != LOSS_BCE else None
    with _lib.on_device(x1.device):
        code = lib.nplda_train_step_dx_f32(_lib.ptr(x1), _lib.ptr(x2), B, x1.stride(0), 1 if bf else 0, _lib.ptr(target), parr,
                                           packed.D0, packed.D1, packed.D2, _theta_array(thetas), barr, K, float(alpha), kind,
                                           _lib.ptr(exp_avg), _lib.ptr(exp_avg_sq), _lib.ptr(step), float(lr), float(beta1),
                                           float(beta2), float(eps), float(weight_decay), _lib.ptr(packed.buf), _lib.ptr(ws),
                                           ws.numel() * 4, _lib.ptr(loss), _lib.ptr(loss_sum) if loss_sum is not None else None,
                                           None, _lib.ptr(dx1), _lib.ptr(dx2), dx1.stride(0), _lib.current_stream())
    _lib.check(code, "nplda_train_step_dx_f32")
    return loss


def train_step_dx_workspace(B, packed, bf16):
    """Workspace tensor for train_step_dx (None when the fused step does not cover this size / shape)."""
    n = _lib.load().nplda_train_step_dx_workspace_bytes(B, packed.D0, packed.D1, packed.D2, 1 if bf16 else 0)
    if n == 0:
        return None
    return torch.empty(n // 4, dtype=torch.float32, device=packed.buf.device)


def train_step_rows(table, rows1, rows2, target, params, thetas, betas, alpha, kind, exp_avg, exp_avg_sq, step, lr, beta1,
                    beta2, eps, weight_decay, packed, ws, loss, grad_out=None, loss_sum=None):
    """nplda_train_step_rows_f32: train_step on the pairs (table[rows1], table[rows2]) of a resident x-vector matrix; the
    first kernel gathers the rows itself.  rows1 / rows2: int64 device tensors with values in [0, len(table))."""
    import ctypes
    lib = _lib.load()
    _need_fp32(packed, "train_step_rows")
    table, ldt = _rows(table, "table", packed.D0)
    dev = table.device
    if rows1.dtype != torch.int64 or rows2.dtype != torch.int64 or rows1.device != dev or rows2.device != dev:
        raise TypeError("rows must be int64 tensors on the table's device")
    rows1, rows2 = rows1.contiguous(), rows2.contiguous()
    B = rows1.shape[0]
    if rows2.shape[0] != B or target.shape[0] != B:
        raise ValueError("rows1, rows2 and target must have the same length")
    _require_dev_f32(target, "target")
    target = target.contiguous()
    if target.data_ptr() % 16:
        target = target.clone()
    for q in list(params) + list(thetas):
        _require_dev_f32(q, "parameter")
        if not q.is_contiguous():
            raise ValueError("train_step updates the parameter tensors in place: they must be contiguous")
    K = len(thetas)
    parr = (ctypes.c_void_p * 6)(*[q.data_ptr() for q in params])
    barr = (ctypes.c_float * max(K, 1))(*[float(b) for b in betas]) if kind != LOSS_BCE else None
    with _lib.on_device(dev):
        code = lib.nplda_train_step_rows_f32(_lib.ptr(table), table.shape[0], ldt, _lib.ptr(rows1), _lib.ptr(rows2), B,
                                             _lib.ptr(target), parr, packed.D0, packed.D1, packed.D2, _theta_array(thetas),
                                             barr, K, float(alpha), kind, _lib.ptr(exp_avg), _lib.ptr(exp_avg_sq),
                                             _lib.ptr(step), float(lr), float(beta1), float(beta2), float(eps),
                                             float(weight_decay), _lib.ptr(packed.buf), _lib.ptr(ws), ws.numel() * 4,
                                             _lib.ptr(loss), _lib.ptr(loss_sum) if loss_sum is not None else None,
                                             _lib.ptr(grad_out) if grad_out is not None else None,
                                             _lib.current_stream())
    _lib.check(code, "nplda_train_step_rows_f32")
    return loss


def train_step_records(table, cursor, stage, B, params, thetas, betas, alpha, kind, exp_avg, exp_avg_sq, step, lr, beta1, beta2,
                       eps, weight_decay, packed, ws, loss, grad_out=None, loss_sum=None):
    """nplda_train_step_records_f32: train_step_rows on the record in `stage` (uint8 device tensor of 20 B bytes:
    [rows1 (B int64) | rows2 (B int64) | labels (B float32)]); the step's last kernel copies the epoch's next record there
    (cursor: int64 device tensor [address of record 0, next record to stage, record count])."""
    import ctypes
    lib = _lib.load()
    _need_fp32(packed, "train_step_records")
    table, ldt = _rows(table, "table", packed.D0)
    if cursor.dtype != torch.int64 or cursor.device != table.device or cursor.numel() != 3 or not cursor.is_contiguous():
        raise TypeError("cursor must be a contiguous int64 tensor of 3 elements on the table's device")
    if stage.dtype != torch.uint8 or stage.device != table.device or stage.numel() != 20 * int(B) or not stage.is_contiguous():
        raise TypeError("stage must be a contiguous uint8 tensor of 20 B bytes on the table's device")
    for q in list(params) + list(thetas):
        _require_dev_f32(q, "parameter")
        if not q.is_contiguous():
            raise ValueError("train_step updates the parameter tensors in place: they must be contiguous")
    K = len(thetas)
    parr = (ctypes.c_void_p * 6)(*[q.data_ptr() for q in params])
    barr = (ctypes.c_float * max(K, 1))(*[float(b) for b in betas]) if kind != LOSS_BCE else None
    with _lib.on_device(table.device):
        code = lib.nplda_train_step_records_f32(_lib.ptr(table), table.shape[0], ldt, _lib.ptr(cursor), _lib.ptr(stage), int(B), parr,
                                                packed.D0, packed.D1, packed.D2, _theta_array(thetas), barr, K,
                                                float(alpha), kind, _lib.ptr(exp_avg), _lib.ptr(exp_avg_sq), _lib.ptr(step),
                                                float(lr), float(beta1), float(beta2), float(eps), float(weight_decay),
                                                _lib.ptr(packed.buf), _lib.ptr(ws), ws.numel() * 4, _lib.ptr(loss),
                                                _lib.ptr(loss_sum) if loss_sum is not None else None,
                                                _lib.ptr(grad_out) if grad_out is not None else None, _lib.current_stream())
    _lib.check(code, "nplda_train_step_records_f32")
    return loss


# ---- indexed scoring / gather -------------------------------------------------------------------

def _idx(t, name, dev):
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(t)
    if t.dtype != torch.int64:
        t = t.long()
    if t.device != dev:
        t = t.to(dev, non_blocking=True)
    return t.contiguous()


def score_indexed(z, q, i1, i2, packed):
    """nplda_score_indexed_f32: z (N, ldz), q (N) from embed(); i1, i2 int64 (B) -> (B,) scores.  q=None: the self terms
    are formed from the z rows inside the kernel (sum_d Q_d z_d^2: two scattered 4-byte reads per pair less)."""
    lib = _lib.load()
    _need_fp32(packed, "score_indexed")
    _require_dev_f32(z, "z")
    if q is not None:
        _require_dev_f32(q, "q")
        q = q.contiguous()
    if z.dim() != 2 or z.stride(1) != 1 or z.stride(0) != packed.ldz or (q is not None and z.shape[0] != q.shape[0]):
        raise ValueError("z must be the (N, ldz) table returned by embed() and q its (N,) self terms")
    dev = z.device
    i1, i2 = _idx(i1, "i1", dev), _idx(i2, "i2", dev)
    if i1.shape != i2.shape or i1.dim() != 1:
        raise ValueError("i1 and i2 must be 1-D index tensors of the same length")
    B = i1.shape[0]
    s = torch.empty(B, dtype=torch.float32, device=dev)
    if B == 0:
        return s
    with _lib.on_device(dev):
        code = lib.nplda_score_indexed_f32(_lib.ptr(z), packed.ldz, _lib.ptr(q), z.shape[0], _lib.ptr(i1),
                                           _lib.ptr(i2), B, _lib.ptr(packed.buf), packed.D0, packed.D1, packed.D2,
                                           _lib.ptr(s), _lib.current_stream())
    _lib.check(code, "nplda_score_indexed_f32")
    return s


def score_embeddings(z1, z2, P_sqrt, Q):
    """nplda_score_embeddings_f32: forward_from_plda_embeddings on explicit (B, D2) tensors."""
    lib = _lib.load()
    for n, t in (("z1", z1), ("z2", z2), ("P_sqrt", P_sqrt), ("Q", Q)):
        _require_dev_f32(t, n)
    D2 = Q.numel()
    if z1.dim() != 2 or z1.shape != z2.shape or z1.shape[1] != D2:
        raise ValueError(f"z1, z2 must both be (B, {D2})")
    if z1.stride(1) != 1:
        z1 = z1.contiguous()
    if z2.stride(1) != 1:
        z2 = z2.contiguous()
    B = z1.shape[0]
    s = torch.empty(B, dtype=torch.float32, device=z1.device)
    if B == 0:
        return s
    ld1 = z1.stride(0) if B > 1 else D2
    ld2 = z2.stride(0) if B > 1 else D2
    with _lib.on_device(z1.device):
        code = lib.nplda_score_embeddings_f32(_lib.ptr(z1), ld1, _lib.ptr(z2), ld2, B, D2,
                                              _lib.ptr(P_sqrt.contiguous()), _lib.ptr(Q.contiguous()), _lib.ptr(s),
                                              _lib.current_stream())
    _lib.check(code, "nplda_score_embeddings_f32")
    return s


def gather_rows(table, idx, out=None):
    """nplda_gather_rows_f32: out[r] = table[idx[r]] for a resident (N, D0) float32 matrix.  `out`: an existing
    contiguous (B, D0) float32 device tensor to fill (static buffers of a captured step)."""
    lib = _lib.load()
    _require_dev_f32(table, "table")
    if table.dim() != 2 or table.stride(1) != 1 or table.stride(0) % 4 != 0 or table.shape[1] % 4 != 0:
        raise ValueError("table must be (N, D0) float32 with unit inner stride and D0 % 4 == 0")
    dev = table.device
    idx = _idx(idx, "idx", dev)
    B, D0 = idx.shape[0], table.shape[1]
    if out is None:
        out = torch.empty((B, D0), dtype=torch.float32, device=dev)
    else:
        _require_dev_f32(out, "out")
        if out.shape != (B, D0) or not out.is_contiguous() or out.device != dev:
            raise ValueError("out must be a contiguous (len(idx), D0) float32 tensor on the table's device")
    if B == 0:
        return out
    with _lib.on_device(dev):
        code = lib.nplda_gather_rows_f32(_lib.ptr(table), table.stride(0), table.shape[0], _lib.ptr(idx), B, D0,
                                         _lib.ptr(out), D0, _lib.current_stream())
    _lib.check(code, "nplda_gather_rows_f32")
    return out


_BAD_FLAGS = {}
KEYERROR_DEFERRED = False  # compat.install(lean=True): see gather_pairs_mapped


def _bad_flag(dev, stream=0):
    """The gather kernels' error word of a device: one int32 in PINNED host memory (the kernel ORs into it over the bus,
    only when a trial number is bad; the host reads it with a load, no copy) -> (tensor, its numpy view)."""
    hit = _BAD_FLAGS.get((dev, stream))  # (a word per stream: whose launch raised it is then unambiguous)
    if hit is None:
        t = torch.zeros(1, dtype=torch.int32).pin_memory()
        hit = _BAD_FLAGS[(dev, stream)] = (t, t.numpy())
    return hit


def _raise_bad(view):
    bad = int(view[0])
    if bad:
        view[0] = 0
        raise KeyError("trial index is outside num_to_id_dict" if bad & 1
                       else "trial index refers to an utterance that is not in mega_dict")


def check_trial_indices(device=None):
    """Wait for the gathers enqueued so far and raise the KeyError of any bad trial number among them (the deferred mode's
    explicit check: validate() and train() at the end of every epoch call it).  In deferred mode the step on a bad batch is
    NOT recoverable: its rows are NaN, so the loss, the gradients and — after the optimiser's step — every parameter are NaN
    by the time the KeyError surfaces; the error says which run to throw away, it does not save it."""
    synced = set()
    for (dev, _), (_, view) in list(_BAD_FLAGS.items()):
        if device is None or torch.device(device) == dev:
            if dev not in synced:
                torch.cuda.synchronize(dev)  # (every stream of the device: a word per stream)
                synced.add(dev)
            _raise_bad(view)


def gather_pairs_mapped(table, num_map, num1, num2, deferred=None):
    """nplda_gather_pairs_mapped_f32: (table[num_map[num1]], table[num_map[num2]]) as two (B, D0) float32 tensors in ONE
    launch — load_xvec_trials_from_numbatch on the device.  Raises the reference's KeyError when a number is outside the
    map or names an utterance that is not in the table: the kernel raises a word in pinned host memory, read after a
    stream synchronise (no copy).  deferred=True (default: KEYERROR_DEFERRED) skips that synchronise — the host keeps
    running ahead of the device — and raises for the launches that HAVE finished: the KeyError of a bad batch then surfaces
    at the next call (or at check_trial_indices()), its rows being NaN in the meantime."""
    lib = _lib.load()
    _require_dev_f32(table, "table")
    if table.dim() != 2 or table.stride(1) != 1 or table.stride(0) % 4 != 0 or table.shape[1] % 4 != 0:
        raise ValueError("table must be (N, D0) float32 with unit inner stride and D0 % 4 == 0")
    dev = table.device
    num1, num2 = _idx(num1.reshape(-1), "num1", dev), _idx(num2.reshape(-1), "num2", dev)
    if num1.shape != num2.shape:
        raise ValueError("num1 and num2 must have the same length")
    if num_map.dtype != torch.int64 or num_map.device != dev or not num_map.is_contiguous():
        raise ValueError("num_map must be a contiguous int64 tensor on the table's device")
    B, D0 = num1.shape[0], table.shape[1]
    out = torch.empty((2, B, D0), dtype=torch.float32, device=dev)
    if B == 0:
        return out[0], out[1]
    if deferred is None:
        deferred = KEYERROR_DEFERRED
    with _lib.on_device(dev):  # (the launch goes to the CURRENT device: dev may not be it in a multi-GPU process)
        st = _lib.current_stream(dev)
        flag, view = _bad_flag(dev, st)
        if deferred:
            _raise_bad(view)  # (an earlier batch's)
        code = lib.nplda_gather_pairs_mapped_f32(table.data_ptr(), table.stride(0), table.shape[0], num_map.data_ptr(),
                                                 num_map.numel(), num1.data_ptr(), num2.data_ptr(), B, D0, out[0].data_ptr(),
                                                 out[1].data_ptr(), D0, flag.data_ptr(), st)
        _lib.check(code, "nplda_gather_pairs_mapped_f32")
        if not deferred:
            torch.cuda.current_stream(dev).synchronize()
    if not deferred:
        _raise_bad(view)
    return out[0], out[1]


# ---- adaptive score normalisation -----------------------------------------------------------------

class PreparedCohort:
    """What nplda_cohort_stats_f32 derives from the cohort alone (nplda_cohort_prepare_f32), kept for the calls that follow:
    `state` is None when the shape takes the spilling path (nothing to prepare)."""
    __slots__ = ("state", "M", "topn", "ldz", "key")

    def __init__(self, state, M, topn, ldz, key):
        self.state, self.M, self.topn, self.ldz, self.key = state, M, topn, ldz, key


def _cohort_key(z_coh, q_coh, packed):
    """What a PreparedCohort was derived from: the cohort's embedding tables and the packed model image (addresses +
    in-place versions).  cohort_stats(prepared=...) recomputes it: a state prepared for ANOTHER cohort or model of the same
    size would otherwise feed its first moments and covariance image into this call's row means and thresholds."""
    def ver(t):
        return 0 if t.is_inference() else t._version
    return (z_coh.data_ptr(), ver(z_coh), q_coh.data_ptr(), ver(q_coh), packed.buf.data_ptr(), ver(packed.buf))


def cohort_prepare(z_coh, q_coh, packed, topn=500):
    """nplda_cohort_prepare_f32: the cohort-only part of cohort_stats (Gram matrix, first moments, the covariance image the
    row thresholds are proposed from), once per (model, cohort, top-N) -> PreparedCohort for cohort_stats(prepared=...)."""
    lib = _lib.load()
    _need_fp32(packed, "cohort_prepare")
    _require_dev_f32(z_coh, "z_coh")
    _require_dev_f32(q_coh, "q_coh")
    if z_coh.stride(0) != packed.ldz:
        raise ValueError("z_coh must come from embed() (row stride = packed.ldz)")
    M = z_coh.shape[0]
    key = _cohort_key(z_coh, q_coh, packed)
    nb = lib.nplda_cohort_state_bytes(M, int(topn), packed.D1, packed.D2) if M > 0 else 0
    if nb == 0:
        return PreparedCohort(None, M, int(topn), packed.ldz, key)
    state = torch.empty((nb + 255) // 4 + 64, dtype=torch.float32, device=z_coh.device)
    off = (-state.data_ptr()) % 256 // 4
    state = state[off:off + (nb + 3) // 4]  # 256-byte aligned view
    qc = q_coh.contiguous()
    with _lib.on_device(z_coh.device):
        code = lib.nplda_cohort_prepare_f32(_lib.ptr(z_coh), _lib.ptr(qc), M, packed.ldz, _lib.ptr(packed.buf), packed.D0,
                                            packed.D1, packed.D2, int(topn), _lib.ptr(state), nb, _lib.current_stream())
    _lib.check(code, "nplda_cohort_prepare_f32")
    return PreparedCohort(state, M, int(topn), packed.ldz, key)


def cohort_stats(z_rows, q_rows, z_coh, q_coh, packed, topn=500, select="lowest", max_ws_bytes=None, force_spill=False,
                 return_fallback_rows=False, prepared=None):
    """nplda_cohort_stats_f32: (R, 4) float64 rows of (mean, std, mean_top, std_top).  force_spill (tests / A-B timing):
    hand the call a workspace just below the fused path's minimum, so that it materialises the score matrix.
    return_fallback_rows (diagnostics): also return how many rows of the (last chunk of the) fused path were handed to the
    general path because the proposed threshold did not bracket their N-th smallest score (reads the workspace: a sync)."""
    lib = _lib.load()
    _need_fp32(packed, "cohort_stats")
    for n, t in (("z_rows", z_rows), ("q_rows", q_rows), ("z_coh", z_coh), ("q_coh", q_coh)):
        _require_dev_f32(t, n)
    if z_rows.stride(0) != packed.ldz or z_coh.stride(0) != packed.ldz:
        raise ValueError("z tables must come from embed() (row stride = packed.ldz)")
    if select not in ("lowest", "highest"):
        raise ValueError("select must be 'lowest' (reference semantics) or 'highest'")
    R, M = z_rows.shape[0], z_coh.shape[0]
    dev = z_rows.device
    stats = torch.empty((R, 4), dtype=torch.float64, device=dev)
    if R == 0:
        return stats
    wsb = lib.nplda_cohort_workspace_bytes_ex(R, M, int(topn), packed.D1, packed.D2)
    if force_spill:
        wsb = lib.nplda_cohort_workspace_bytes(R, M)
    if max_ws_bytes is not None:
        # never below one score row, nor below what keeps the call on the path its shape selects (fused / spilling)
        floor = max(256 + ((M + 3) // 4 * 4) * 4,
                    lib.nplda_cohort_fused_min_workspace_bytes(M, int(topn), packed.D1, packed.D2))
        wsb = max(min(wsb, int(max_ws_bytes)), floor)
    if force_spill:
        fmin = lib.nplda_cohort_fused_min_workspace_bytes(M, int(topn), packed.D1, packed.D2)
        if fmin:
            wsb = min(wsb, fmin - 256)
            if wsb < 256 + ((M + 3) // 4 * 4) * 4:
                raise ValueError("no workspace size selects the spilling path for this shape")
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=dev)
    use_prep = prepared is not None and prepared.state is not None and not force_spill
    if prepared is not None and (prepared.M != M or prepared.topn != int(topn) or prepared.ldz != packed.ldz):
        raise ValueError("prepared cohort does not belong to this cohort table / top-N")
    if use_prep and prepared.key != _cohort_key(z_coh, q_coh, packed):  # (state None: nothing was derived, nothing to mismatch)
        raise ValueError("prepared cohort was derived from other cohort tables / another model image (or they were "
                         "modified since): call cohort_prepare() again")
    with _lib.on_device(dev):
        if use_prep:
            st = prepared.state
            code = lib.nplda_cohort_stats_prepared_f32(
                _lib.ptr(z_rows), _lib.ptr(q_rows.contiguous()), R, _lib.ptr(z_coh), _lib.ptr(q_coh.contiguous()), M, packed.ldz,
                _lib.ptr(packed.buf), packed.D0, packed.D1, packed.D2, int(topn), 1 if select == "lowest" else 0, _lib.ptr(stats),
                _lib.ptr(ws), wsb, _lib.ptr(st), st.numel() * 4, _lib.current_stream())
        else:
            code = lib.nplda_cohort_stats_f32(_lib.ptr(z_rows), _lib.ptr(q_rows.contiguous()), R, _lib.ptr(z_coh),
                                              _lib.ptr(q_coh.contiguous()), M, packed.ldz, _lib.ptr(packed.buf), packed.D0,
                                              packed.D1, packed.D2, int(topn), 1 if select == "lowest" else 0,
                                              _lib.ptr(stats), _lib.ptr(ws), wsb, _lib.current_stream())
    _lib.check(code, "nplda_cohort_stats_prepared_f32" if use_prep else "nplda_cohort_stats_f32")
    if return_fallback_rows:
        fused = lib.nplda_cohort_fused_min_workspace_bytes(M, int(topn), packed.D1, packed.D2)
        return stats, (int(ws.view(torch.int32)[8].item()) if fused and wsb >= fused else None)
    return stats


def row_stats(S, topn=500, select="lowest"):
    """nplda_row_stats_f32 on an explicit (R, M) float32 score matrix."""
    lib = _lib.load()
    _require_dev_f32(S, "S")
    if S.dim() != 2:
        raise ValueError("S must be (R, M)")
    if S.stride(1) != 1:
        S = S.contiguous()
    R, M = S.shape
    stats = torch.empty((R, 4), dtype=torch.float64, device=S.device)
    if R == 0:
        return stats
    with _lib.on_device(S.device):
        code = lib.nplda_row_stats_f32(_lib.ptr(S), S.stride(0) if R > 1 else M, R, M, int(topn),
                                       1 if select == "lowest" else 0, _lib.ptr(stats), _lib.current_stream())
    _lib.check(code, "nplda_row_stats_f32")
    return stats


def asnorm_apply(raw, ie, it, stats):
    """nplda_asnorm_apply_f64: raw (T,) -> (T, 4) float64 columns znorm, tnorm, snorm, asnorm1."""
    lib = _lib.load()
    dev = stats.device
    if not stats.is_cuda or stats.dtype != torch.float64 or stats.dim() != 2 or stats.shape[1] != 4:
        raise ValueError("stats must be the (R, 4) float64 device tensor returned by cohort_stats/row_stats")
    raw = torch.as_tensor(raw).to(dev).double().contiguous()
    ie, it = _idx(ie, "ie", dev), _idx(it, "it", dev)
    T = raw.shape[0]
    out = torch.empty((T, 4), dtype=torch.float64, device=dev)
    if T == 0:
        return out
    with _lib.on_device(dev):
        code = lib.nplda_asnorm_apply_f64(_lib.ptr(raw), _lib.ptr(ie), _lib.ptr(it), T, _lib.ptr(stats.contiguous()),
                                          stats.shape[0], _lib.ptr(out), _lib.current_stream())
    _lib.check(code, "nplda_asnorm_apply_f64")
    return out


# ---- GaussianBackend ---------------------------------------------------------------------------------

def gb_pack(W1, b1, mu_t, Lam_t, mu_n, Lam_n):
    """gb_pack_params_f32 -> (buffer, D0, D1)."""
    lib = _lib.load()
    ts = []
    for n, t in (("W1", W1), ("b1", b1), ("mu_t", mu_t), ("Lam_t", Lam_t), ("mu_n", mu_n), ("Lam_n", Lam_n)):
        _require_dev_f32(t, n)
        ts.append(t.detach().contiguous())
    D1, D0 = W1.shape
    if mu_t.numel() != 2 * D1 or Lam_t.shape != (2 * D1, 2 * D1) or mu_n.numel() != 2 * D1 or Lam_n.shape != (2 * D1, 2 * D1):
        raise ValueError("paired statistics must have dimension 2 * layer1_LDA_dim")
    if D0 % 4 != 0:
        raise ValueError(f"xvector_dim must be a multiple of 4 (got {D0})")
    nbytes = lib.gb_packed_bytes(D0, D1)
    if nbytes == 0:
        raise _lib.NpldaHipError(f"GaussianBackend {D0}->{D1} is outside the compiled kernel set")
    buf = torch.empty(nbytes // 4, dtype=torch.float32, device=W1.device)
    with _lib.on_device(W1.device):
        code = lib.gb_pack_params_f32(*[_lib.ptr(t) for t in ts], D0, D1, _lib.ptr(buf), nbytes, _lib.current_stream())
    _lib.check(code, "gb_pack_params_f32")
    return buf, D0, D1


def _gb_call(x1, x2, packed, want_s, want_paired, want_rn=False):
    lib = _lib.load()
    buf, D0, D1 = packed
    x1, ld1 = _rows(x1, "x1", D0)
    x2, ld2 = _rows(x2, "x2", D0)
    if x1.shape[0] != x2.shape[0]:
        raise ValueError("x1 and x2 must have the same number of rows")
    if ld1 != ld2:
        x1, x2 = x1.contiguous(), x2.contiguous()
        ld1 = ld2 = D0
    B = x1.shape[0]
    s = torch.empty(B, dtype=torch.float32, device=x1.device) if want_s else None
    paired = torch.empty((B, 2 * D1), dtype=torch.float32, device=x1.device) if want_paired else None
    rn = torch.empty(2 * B, dtype=torch.float32, device=x1.device) if want_rn else None
    if B > 0:
        with _lib.on_device(x1.device):
            code = lib.gb_score_pairs_ex_f32(_lib.ptr(x1), _lib.ptr(x2), B, ld1, _lib.ptr(buf), D0, D1, _lib.ptr(s),
                                             _lib.ptr(paired), _lib.ptr(rn), _lib.current_stream())
        _lib.check(code, "gb_score_pairs_ex_f32")
    return (s, paired, rn) if want_rn else (s, paired)


def gb_score_pairs(x1, x2, W1, b1, mu_t, Lam_t, mu_n, Lam_n):
    """GaussianBackend.forward: (B, D0) x 2 -> (B,)."""
    return _gb_call(x1, x2, gb_pack(W1, b1, mu_t, Lam_t, mu_n, Lam_n), True, False)[0]


def gb_paired(x1, x2, W1, b1):
    """GaussianBackend.forward_getpaired: (B, D0) x 2 -> (B, 2 D1) = [normalize(LDA x1), normalize(LDA x2)]."""
    D1 = W1.shape[0]
    z = torch.zeros(2 * D1, dtype=torch.float32, device=W1.device)
    Z = torch.zeros((2 * D1, 2 * D1), dtype=torch.float32, device=W1.device)
    return _gb_call(x1, x2, gb_pack(W1, b1, z, Z, z, Z), False, True)[1]


def quadform_pack(W1, b1, M, v, c):
    """gb_pack_quadform_f32: image for S = x^T M x + x^T v + c on x = [normalize(LDA x1); normalize(LDA x2)]."""
    lib = _lib.load()
    for n, t in (("W1", W1), ("b1", b1), ("M", M)):
        _require_dev_f32(t, n)
    D1, D0 = W1.shape
    if M.shape != (2 * D1, 2 * D1) or (v is not None and v.numel() != 2 * D1):
        raise ValueError("M must be (2 D1, 2 D1) and v (2 D1)")
    if D0 % 4 != 0:
        raise ValueError(f"xvector_dim must be a multiple of 4 (got {D0})")
    nbytes = lib.gb_packed_bytes(D0, D1)
    if nbytes == 0:
        raise _lib.NpldaHipError(f"model {D0}->{D1} is outside the compiled kernel set")
    buf = torch.empty(nbytes // 4, dtype=torch.float32, device=W1.device)
    ts = [W1.detach().contiguous(), b1.detach().contiguous(), M.detach().contiguous(),
          None if v is None else v.detach().contiguous()]
    with _lib.on_device(W1.device):
        code = lib.gb_pack_quadform_f32(_lib.ptr(ts[0]), _lib.ptr(ts[1]), _lib.ptr(ts[2]), _lib.ptr(ts[3]), float(c), D0,
                                        D1, _lib.ptr(buf), nbytes, _lib.current_stream())
    _lib.check(code, "gb_pack_quadform_f32")
    return buf, D0, D1


def dplda_pack(W1, b1, wlr, blr, out=None):
    """gb_pack_dplda_f32: the quadratic-form image straight from DPlda's parameters (no host sync, no temporaries).
    out: an image returned earlier, refilled in place (its address may be baked into a captured graph)."""
    lib = _lib.load()
    for n, t in (("W1", W1), ("b1", b1), ("wlr", wlr), ("blr", blr)):
        _require_dev_f32(t, n)
    D1, D0 = W1.shape
    if wlr.numel() != 2 * D1 * D1 + D1 or blr.numel() != 1:
        raise ValueError("logistic_regres must be Linear(2 D1^2 + D1, 1)")
    if D0 % 4 != 0:
        raise ValueError(f"xvector_dim must be a multiple of 4 (got {D0})")
    nbytes = lib.gb_packed_bytes(D0, D1)
    if nbytes == 0:
        raise _lib.NpldaHipError(f"model {D0}->{D1} is outside the compiled kernel set")
    if out is not None and (out[1] != D0 or out[2] != D1 or out[0].numel() * 4 < nbytes or out[0].device != W1.device):
        out = None
    buf = out[0] if out is not None else torch.empty(nbytes // 4, dtype=torch.float32, device=W1.device)
    ts = [t.detach().contiguous() for t in (W1, b1, wlr, blr)]
    with _lib.on_device(W1.device):
        code = lib.gb_pack_dplda_f32(*[_lib.ptr(t) for t in ts], D0, D1, _lib.ptr(buf), nbytes, _lib.current_stream())
    _lib.check(code, "gb_pack_dplda_f32")
    return buf, D0, D1


def quadform_score_pairs(x1, x2, packed):
    """(B, D0) x 2 -> (B,) scores of the packed quadratic form (kernel: MODE_GB of the fused forward)."""
    return _gb_call(x1, x2, packed, True, False)[0]


def quadform_score_rows(y1, y2, packed):
    """gb_score_rows_f32: like quadform_score_pairs with the normalisation step skipped."""
    lib = _lib.load()
    buf, D0, D1 = packed
    y1, ld1 = _rows(y1, "y1", D0)
    y2, ld2 = _rows(y2, "y2", D0)
    if y1.shape[0] != y2.shape[0]:
        raise ValueError("y1 and y2 must have the same number of rows")
    if ld1 != ld2:
        y1, y2 = y1.contiguous(), y2.contiguous()
        ld1 = ld2 = D0
    B = y1.shape[0]
    s = torch.empty(B, dtype=torch.float32, device=y1.device)
    if B > 0:
        with _lib.on_device(y1.device):
            code = lib.gb_score_rows_f32(_lib.ptr(y1), _lib.ptr(y2), B, ld1, _lib.ptr(buf), D0, D1, _lib.ptr(s),
                                         _lib.current_stream())
        _lib.check(code, "gb_score_rows_f32")
    return s


def dplda_quadform(wlr, blr, D1):
    """DPlda's single linear unit over [y1 y2^T + y2 y1^T, y1 y1^T + y2 y2^T, y1 + y2] (utils/models.py:484-490)
    as (M, v, c) of x^T M x + x^T v + c, x = [y1; y2]:  M = [[Ww, Wb], [Wb, Ww]], v = [ws; ws], c = bias."""
    w = wlr.detach().reshape(-1)
    n = D1 * D1
    if w.numel() != 2 * n + D1:
        raise ValueError("logistic_regres.weight must have 2 D1^2 + D1 inputs")
    Wb, Ww, ws = w[:n].reshape(D1, D1), w[n:2 * n].reshape(D1, D1), w[2 * n:]
    M = torch.cat([torch.cat([Ww, Wb], 1), torch.cat([Wb, Ww], 1)], 0).contiguous()
    return M, torch.cat([ws, ws]).contiguous(), (float(blr.detach().reshape(-1)[0]) if blr is not None else 0.0)


def dplda_quadform_image(wlr, D1):
    """nplda_dplda_quadform_f32: ((frag, K, N) of M + M^T, v) for rows_matmul — dplda_quadform + pack_matrix(mode=2)
    in one launch (the input-side backward of DPlda's quadratic form)."""
    lib = _lib.load()
    _require_dev_f32(wlr, "wlr")
    w = wlr.detach().reshape(-1)
    if w.numel() != 2 * D1 * D1 + D1:
        raise ValueError("logistic_regres.weight must have 2 D1^2 + D1 inputs")
    w = w.contiguous()
    K = 2 * D1
    nbytes = lib.nplda_matrix_frag_bytes(K, K)
    if nbytes == 0:
        raise _lib.NpldaHipError(f"a {K} x {K} matrix is outside the resident-matrix GEMM (K <= 512, N % 4 == 0)")
    frag = torch.empty(nbytes // 4, dtype=torch.float32, device=w.device)
    v = torch.empty(K, dtype=torch.float32, device=w.device)
    with _lib.on_device(w.device):
        code = lib.nplda_dplda_quadform_f32(_lib.ptr(w), D1, _lib.ptr(frag), nbytes, _lib.ptr(v), _lib.current_stream())
    _lib.check(code, "nplda_dplda_quadform_f32")
    return (frag, K, K), v


def dplda_grad(paired, g, D1):
    """nplda_dplda_grad_f32: (d logistic_regres.weight (1, 2 D1^2 + D1), d bias (1,)) from the paired rows and g = dL/ds —
    weighted_moments + dplda_fold_grad (same bits) without the fp64 moment matrix in between."""
    lib = _lib.load()
    _require_dev_f32(paired, "paired")
    _require_dev_f32(g, "g")
    B, n = paired.shape
    if n != 2 * D1 or g.numel() != B:
        raise ValueError("paired must be (B, 2 D1) and g (B,)")
    if paired.stride(1) != 1 or paired.stride(0) % 4 != 0 or paired.data_ptr() % 16 != 0:
        paired = paired.contiguous()
    nbytes = lib.nplda_moments_workspace_bytes(B, n)
    if nbytes == 0:
        raise _lib.NpldaHipError(f"paired rows of length {n} are outside the moments kernel")
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=paired.device)
    out = torch.empty(2 * D1 * D1 + D1 + 1, dtype=torch.float32, device=paired.device)
    gg = g.detach().contiguous()
    with _lib.on_device(paired.device):
        code = lib.nplda_dplda_grad_f32(_lib.ptr(paired), B, paired.stride(0), D1, _lib.ptr(gg), _lib.ptr(out),
                                        out.data_ptr() + 4 * (out.numel() - 1), _lib.ptr(ws), nbytes, _lib.current_stream())
    _lib.check(code, "nplda_dplda_grad_f32")
    return out[:-1].view(1, -1), out[-1:]


def dplda_update(paired, g, wlr, blr, m, v, step, lr, beta1, beta2, eps, wd, thetas=(), dtheta=None, image=None, ws=None,
                 grad_out=None, loss=None, loss_sum=None):
    """nplda_dplda_update_f32: the tail of DPlda's recipe step in two launches — weighted moments of the paired rows, then per
    element of [logistic_regres.weight | bias] gradient fold + torch.optim.Adam's update + the new value stored into the
    parameter and into `image` (the (buf, D0, D1) of dplda_pack the next forward scores with).  m / v: flat moments
    [weight | bias | thresholds]; step: device [steps taken, scratch].  loss / loss_sum: optional 0-d float32 / (1,) float64
    device tensors, loss_sum += loss inside the update launch."""
    import ctypes
    lib = _lib.load()
    B, n = paired.shape
    if loss_sum is not None and (loss is None or loss.dtype != torch.float32 or loss_sum.dtype != torch.float64):
        raise ValueError("dplda_update: loss must be float32 and loss_sum float64")
    D1 = n // 2
    K = len(thetas)
    if wlr.numel() != 2 * D1 * D1 + D1 or blr.numel() != 1 or m.numel() < wlr.numel() + 1 + K or v.numel() != m.numel():
        raise ValueError("dplda_update: parameter / moment shapes do not belong to a DPlda of this D1")
    if paired.stride(1) != 1 or paired.stride(0) % 4 != 0 or paired.data_ptr() % 16 != 0:
        paired = paired.contiguous()
    nbytes = lib.nplda_moments_workspace_bytes(B, n)
    if nbytes == 0:
        raise _lib.NpldaHipError(f"paired rows of length {n} are outside the moments kernel")
    if ws is None or ws.numel() * 4 < nbytes:
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=paired.device)
    gg = g.detach().contiguous()
    tarr = (ctypes.c_void_p * max(K, 1))(*[t.data_ptr() for t in thetas]) if K else None
    buf, D0 = (image[0], image[1]) if image is not None else (None, 0)
    with _lib.on_device(paired.device):
        code = lib.nplda_dplda_update_f32(_lib.ptr(paired), B, paired.stride(0), D1, _lib.ptr(gg), _lib.ptr(wlr), _lib.ptr(blr),
                                          _lib.ptr(m), _lib.ptr(v), tarr, _lib.ptr(dtheta) if K else None, K, _lib.ptr(step),
                                          float(lr), float(beta1), float(beta2), float(eps), float(wd), _lib.ptr(buf), int(D0),
                                          _lib.ptr(grad_out), _lib.ptr(loss) if loss_sum is not None else None,
                                          _lib.ptr(loss_sum), _lib.ptr(ws), ws.numel() * 4, _lib.current_stream())
    _lib.check(code, "nplda_dplda_update_f32")
    return ws


def dplda_update_loss(paired, s, t, loss_thetas, betas, alpha, kind, wlr, blr, m, v, step, lr, beta1, beta2, eps, wd, thetas=(),
                      image=None, ws=None, grad_out=None, loss_sum=None):
    """nplda_dplda_update_loss_f32: dplda_update with the loss inside its first launch — the moments blocks form dL/ds_i from
    (s, t) themselves, one more block of that launch is the loss kernel (loss, dL/dtheta).  Returns (loss, dtheta, ws), or None
    when the batch is outside the one-block loss (B > 4096 / unaligned): the caller runs loss_fwd_bwd + dplda_update."""
    import ctypes
    lib = _lib.load()
    B, n = paired.shape
    D1 = n // 2
    K, LK = len(thetas), len(loss_thetas)
    if wlr.numel() != 2 * D1 * D1 + D1 or blr.numel() != 1 or m.numel() < wlr.numel() + 1 + K or v.numel() != m.numel():
        raise ValueError("dplda_update_loss: parameter / moment shapes do not belong to a DPlda of this D1")
    _require_dev_f32(s, "output")
    _require_dev_f32(t, "target")
    if s.shape != (B,) or t.shape != (B,):
        raise ValueError("output and target must be (B,) tensors")
    if B > 4096 or not s.is_contiguous() or not t.is_contiguous() or s.data_ptr() % 16 or t.data_ptr() % 16:
        return None
    if paired.stride(1) != 1 or paired.stride(0) % 4 != 0 or paired.data_ptr() % 16 != 0:
        paired = paired.contiguous()
    ns = lib.nplda_loss_nsums(LK, kind)
    if ns == 0 or kind == LOSS_HARD_CDET:
        raise _lib.NpldaHipError("dplda_update_loss: unsupported loss kind / number of thresholds")
    nbytes = lib.nplda_moments_workspace_bytes(B, n)
    if nbytes == 0:
        raise _lib.NpldaHipError(f"paired rows of length {n} are outside the moments kernel")
    if ws is None or ws.numel() * 4 < nbytes:
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=paired.device)
    dev = paired.device
    sums = torch.empty(ns, dtype=torch.float64, device=dev)
    loss = torch.empty((), dtype=torch.float32, device=dev)
    dth = torch.empty(max(LK, 1), dtype=torch.float32, device=dev)
    barr = (ctypes.c_float * max(LK, 1))(*[float(b) for b in betas]) if kind != LOSS_BCE else None
    tarr = (ctypes.c_void_p * max(K, 1))(*[x.data_ptr() for x in thetas]) if K else None
    buf, D0 = (image[0], image[1]) if image is not None else (None, 0)
    with _lib.on_device(dev):
        code = lib.nplda_dplda_update_loss_f32(_lib.ptr(paired), B, paired.stride(0), D1, _lib.ptr(s), _lib.ptr(t), kind,
                                               _theta_array(loss_thetas), barr, LK, float(alpha), _lib.ptr(sums), _lib.ptr(loss),
                                               _lib.ptr(dth), None, _lib.ptr(wlr), _lib.ptr(blr), _lib.ptr(m), _lib.ptr(v), tarr, K,
                                               _lib.ptr(step), float(lr), float(beta1), float(beta2), float(eps), float(wd),
                                               _lib.ptr(buf), int(D0), _lib.ptr(grad_out), _lib.ptr(loss_sum), _lib.ptr(ws),
                                               ws.numel() * 4, _lib.current_stream())
    _lib.check(code, "nplda_dplda_update_loss_f32")
    return loss, dth, ws


def weighted_moments(x, w0, w1=None, out=None):
    """nplda_weighted_moments_f32: x (B, n) fp32, w0/w1 (B,) fp32 -> (cnt (nc,), sum (nc, n), sq (nc, n, n)) doubles,
    nc = 2 if w1 is given else 1.  `out` = a previous result to accumulate into (streamed batches)."""
    lib = _lib.load()
    _require_dev_f32(x, "x")
    _require_dev_f32(w0, "w0")
    if w1 is not None:
        _require_dev_f32(w1, "w1")
    if x.dim() != 2:
        raise ValueError("x must be 2-D")
    B, n = x.shape
    if x.stride(1) != 1 or x.stride(0) % 4 != 0 or x.data_ptr() % 16 != 0:
        x = x.contiguous()
    if n % 4 != 0:
        raise ValueError(f"row length must be a multiple of 4 (got {n})")
    ws = [w0.contiguous()] + ([w1.contiguous()] if w1 is not None else [])
    if any(w.numel() != B for w in ws):
        raise ValueError("weights must have one entry per row")
    nc = len(ws)
    if out is None:
        cnt = torch.empty(nc, dtype=torch.float64, device=x.device)
        sm = torch.empty((nc, n), dtype=torch.float64, device=x.device)
        sq = torch.empty((nc, n, n), dtype=torch.float64, device=x.device)
        acc = 0
    else:
        cnt, sm, sq = out
        if cnt.shape != (nc,) or sm.shape != (nc, n) or sq.shape != (nc, n, n) or sq.dtype != torch.float64:
            raise ValueError("`out` does not match this call")
        acc = 1
    nbytes = lib.nplda_moments_workspace_bytes(B, n)
    if nbytes == 0:
        raise _lib.NpldaHipError(f"row length {n} is outside the compiled kernel set")
    wsb = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    with _lib.on_device(x.device):
        code = lib.nplda_weighted_moments_f32(_lib.ptr(x), B, x.stride(0) if B > 0 else n, n, _lib.ptr(ws[0]),
                                              _lib.ptr(ws[1]) if nc == 2 else None, _lib.ptr(cnt), _lib.ptr(sm),
                                              _lib.ptr(sq), acc, _lib.ptr(wsb), nbytes, _lib.current_stream())
    _lib.check(code, "nplda_weighted_moments_f32")
    return cnt, sm, sq


def dplda_fold_grad(cnt, sm, sq, D1, reduce=None):
    """(cnt, sum, sq) of paired rows weighted by g = dL/ds -> gradient of DPlda's linear unit (utils/models.py:484-490):
    d wlr = [G12 + G21 | G11 + G22 | s1 + s2] (row-major blocks of G = sum g x x^T), d bias = sum g.
    `reduce` (data parallel): sum-all-reduce of the folded fp64 vector [d wlr | d bias] before it is rounded to fp32,
    so that N ranks give the single-process gradient to fp64 rounding."""
    G = sq[0]
    G11, G12, G21, G22 = G[:D1, :D1], G[:D1, D1:], G[D1:, :D1], G[D1:, D1:]
    dw = torch.cat([(G12 + G21).reshape(-1), (G11 + G22).reshape(-1), sm[0, :D1] + sm[0, D1:], cnt[:1]])
    if reduce is not None:
        dw = reduce(dw)
    return dw[:-1].float().reshape(1, -1), dw[-1:].float()


def detcost_sweep(scores, target, betas, exact=False, want_eer=False):
    """nplda_detcost_sweep_f32 -> (minc (K,), thr (K,), minc_avg (1,), eer (1,) or None), float32 device tensors."""
    lib = _lib.load()
    _require_dev_f32(scores, "scores")
    _require_dev_f32(target, "target")
    s, t = scores.detach().reshape(-1).contiguous(), target.detach().reshape(-1).contiguous()
    if s.numel() != t.numel():
        raise ValueError("scores and targets must have the same length")
    K = len(betas)
    if K < 1:
        raise ValueError("at least one beta")
    N = s.numel()
    nbytes = lib.nplda_detcost_workspace_bytes(N)
    if nbytes == 0:
        raise _lib.NpldaHipError(f"{N} scores are outside the supported range")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=s.device)
    out = torch.empty(2 * K + 2, dtype=torch.float32, device=s.device)
    import ctypes
    barr = (ctypes.c_float * K)(*[float(b) for b in betas])
    base = out.data_ptr()
    with _lib.on_device(s.device):
        code = lib.nplda_detcost_sweep_f32(_lib.ptr(s), _lib.ptr(t), N, barr, K, 1 if exact else 0, base, base + 4 * K,
                                           base + 8 * K, (base + 8 * K + 4) if want_eer else None, _lib.ptr(ws), nbytes,
                                           _lib.current_stream())
    _lib.check(code, "nplda_detcost_sweep_f32")
    return out[:K], out[K:2 * K], out[2 * K:2 * K + 1], (out[2 * K + 1:] if want_eer else None)
