"""Thin tensor-level wrappers over the C ABI (include/nplda_hip.h).

Every function takes torch tensors that live on a HIP device, enqueues the kernel on torch's
current stream and returns torch tensors.  No CPU path exists: a CPU tensor is an error here
(neuralplda_amd.models stages CPU inputs through the device, it never computes on the host).
"""
import torch

from . import _lib


class PackedParams:
    """MFMA-fragment-ordered image of one parameter set (see csrc/nplda_common.h)."""

    __slots__ = ("buf", "D0", "D1", "D2", "ldz")

    def __init__(self, buf, D0, D1, D2, ldz):
        self.buf, self.D0, self.D1, self.D2, self.ldz = buf, D0, D1, D2, ldz

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


def pack_params(W1, b1, W2, b2, P_sqrt, Q):
    """nplda_pack_params_f32: nn.Linear-layout parameters -> PackedParams on the same device."""
    lib = _lib.load()
    for n, t in (("W1", W1), ("b1", b1), ("W2", W2), ("b2", b2), ("P_sqrt", P_sqrt), ("Q", Q)):
        _require_dev_f32(t, n)
    D1, D0 = W1.shape
    D2, D1b = W2.shape
    if D1b != D1 or b1.numel() != D1 or b2.numel() != D2 or P_sqrt.numel() != D2 or Q.numel() != D2:
        raise ValueError("inconsistent parameter shapes")
    if D0 % 4 != 0:
        raise ValueError(f"xvector_dim must be a multiple of 4 (got {D0}); pad the x-vectors")
    nbytes = lib.nplda_packed_bytes(D0, D1, D2)
    if nbytes == 0:
        raise _lib.NpldaHipError(
            f"model {D0}->{D1}->{D2} is outside the compiled kernel set (max dim {lib.nplda_max_dim()})")
    buf = torch.empty(nbytes // 4, dtype=torch.float32, device=W1.device)
    ts = [t.detach().contiguous() for t in (W1, b1, W2, b2, P_sqrt, Q)]
    with torch.cuda.device(W1.device):
        code = lib.nplda_pack_params_f32(*[_lib.ptr(t) for t in ts], D0, D1, D2, _lib.ptr(buf), nbytes,
                                         _lib.current_stream())
    _lib.check(code, "nplda_pack_params_f32")
    return PackedParams(buf, D0, D1, D2, lib.nplda_padded_dim(D1, D2))


def score_pairs(x1, x2, packed):
    """nplda_score_pairs_f32: (B, D0), (B, D0) -> (B,) scores."""
    lib = _lib.load()
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
    with torch.cuda.device(x1.device):
        code = lib.nplda_score_pairs_f32(_lib.ptr(x1), _lib.ptr(x2), B, ld1, _lib.ptr(packed.buf), packed.D0,
                                         packed.D1, packed.D2, _lib.ptr(s), _lib.current_stream())
    _lib.check(code, "nplda_score_pairs_f32")
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
    with torch.cuda.device(x.device):
        code = lib.nplda_embed_f32(_lib.ptr(x), N, ld, _lib.ptr(packed.buf), packed.D0, packed.D1, packed.D2,
                                   _lib.ptr(z), packed.ldz, _lib.ptr(q), _lib.current_stream())
    _lib.check(code, "nplda_embed_f32")
    return z, q
