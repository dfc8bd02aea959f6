"""Native readers for the Kaldi objects the NPLDA initialisation consumes.

The reference shells out to Kaldi binaries (`copy-matrix`, `copy-vector`, `ivector-copy-plda`:
utils/models.py:442-448, utils/Kaldi2NumpyUtils/kaldiPlda2numpydict.py:17) and parses their text
output.  This module reads the same files directly — binary (`\\0B` + `FM `/`DM `/`FV `/`DV `
tokens, `\\x04`+int32 sizes, little-endian payload) or Kaldi text (`[ ... ]`) — so no Kaldi
installation is needed.  It is host-side file I/O only.
"""
import io
import os
import struct

import numpy as np

__all__ = ["read_vector", "read_matrix", "read_plda", "plda_psi_to_pq", "read_vector_ark", "read_vector_scp",
           "read_scp", "load_vector_ark", "load_vector_scp", "write_vector_ark", "fold_init",
           "write_matrix_binary", "write_vector_binary", "write_plda_binary"]


class KaldiFormatError(ValueError):
    pass


def _open(f):
    if isinstance(f, (bytes, bytearray)):
        return io.BytesIO(bytes(f)), True
    if hasattr(f, "read"):
        return f, False
    return open(f, "rb"), True


def _peek(fh, n):
    pos = fh.tell()
    b = fh.read(n)
    fh.seek(pos)
    return b


def _is_binary(fh):
    if _peek(fh, 2) == b"\0B":
        fh.read(2)
        return True
    return False


def _read_token(fh):
    """Whitespace-delimited token (binary mode tokens are followed by one space)."""
    tok = b""
    while True:
        c = fh.read(1)
        if not c:
            break
        if c in b" \t\r\n":
            if tok:
                break
            continue
        tok += c
    return tok.decode("ascii")


def _read_int32(fh):
    sz = fh.read(1)
    if sz != b"\x04":
        raise KaldiFormatError(f"expected int32 size marker \\x04, got {sz!r}")
    return struct.unpack("<i", fh.read(4))[0]


def _read_binary_vector_body(fh, tok):
    if tok not in ("FV", "DV"):
        raise KaldiFormatError(f"expected FV/DV vector token, got {tok!r}")
    dt = np.dtype("<f4") if tok == "FV" else np.dtype("<f8")
    n = _read_int32(fh)
    buf = fh.read(n * dt.itemsize)
    if len(buf) != n * dt.itemsize:
        raise KaldiFormatError("truncated vector payload")
    return np.frombuffer(buf, dtype=dt).astype(np.float64)


def _read_binary_matrix_body(fh, tok):
    if tok == "CM" or tok.startswith("CM"):
        raise KaldiFormatError("compressed Kaldi matrices (CM) are not supported")
    if tok not in ("FM", "DM"):
        raise KaldiFormatError(f"expected FM/DM matrix token, got {tok!r}")
    dt = np.dtype("<f4") if tok == "FM" else np.dtype("<f8")
    r = _read_int32(fh)
    c = _read_int32(fh)
    buf = fh.read(r * c * dt.itemsize)
    if len(buf) != r * c * dt.itemsize:
        raise KaldiFormatError("truncated matrix payload")
    return np.frombuffer(buf, dtype=dt).reshape(r, c).astype(np.float64)


def _read_text_vector_body(fh):
    """Text vector: ` [ v v v ]` (the opening bracket may already have been consumed)."""
    vals = []
    while True:
        tok = _read_token(fh)
        if tok == "":
            raise KaldiFormatError("unterminated text vector")
        if tok == "[":
            continue
        if tok == "]":
            break
        if tok.endswith("]"):
            vals.append(float(tok[:-1]))
            break
        vals.append(float(tok))
    return np.asarray(vals, dtype=np.float64)


def _read_text_matrix_body(fh):
    """Text matrix: ` [\\n r r r\\n r r r ]`."""
    text = b""
    while True:
        c = fh.read(1)
        if not c:
            raise KaldiFormatError("unterminated text matrix")
        text += c
        if c == b"]":
            break
    body = text.decode("ascii").replace("[", " ").replace("]", " ")
    rows = [ln.split() for ln in body.strip().split("\n") if ln.strip()]
    if not rows:
        return np.zeros((0, 0))
    return np.asarray(rows, dtype=np.float64)


def read_vector(f):
    """Kaldi Vector<float|double>, binary or text (e.g. Kaldi_Models/mean.vec) -> float64 (n,)."""
    fh, close = _open(f)
    try:
        if _is_binary(fh):
            return _read_binary_vector_body(fh, _read_token(fh))
        return _read_text_vector_body(fh)
    finally:
        if close:
            fh.close()


def read_matrix(f):
    """Kaldi Matrix<float|double>, binary or text (e.g. Kaldi_Models/transform.mat) -> float64 (r, c)."""
    fh, close = _open(f)
    try:
        if _is_binary(fh):
            return _read_binary_matrix_body(fh, _read_token(fh))
        return _read_text_matrix_body(fh)
    finally:
        if close:
            fh.close()


def plda_psi_to_pq(psi):
    """Diagonal P/Q of the PLDA log-likelihood ratio from the between-class variances Psi
    (utils/Kaldi2NumpyUtils/kaldiPlda2numpydict.py:34-38)."""
    ac = np.asarray(psi, dtype=np.float64)
    tot = 1.0 + ac
    diagP = ac / (tot * (tot - ac * ac / tot))
    diagQ = (1.0 / tot) - 1.0 / (tot - ac * ac / tot)
    return diagP, diagQ


def read_plda(f):
    """Kaldi `Plda` object (ivector-compute-plda output), binary or text.  Returns the dict the
    reference builds (utils/Kaldi2NumpyUtils/kaldiPlda2numpydict.py:15-43): plda_mean,
    diagonalizing_transform, Psi_across_covar_diag, diagP, diagQ."""
    fh, close = _open(f)
    try:
        binary = _is_binary(fh)
        tok = _read_token(fh)
        if tok != "<Plda>":
            raise KaldiFormatError(f"expected <Plda>, got {tok!r}")
        if binary:
            mean = _read_binary_vector_body(fh, _read_token(fh))
            trans = _read_binary_matrix_body(fh, _read_token(fh))
            psi = _read_binary_vector_body(fh, _read_token(fh))
        else:
            mean = _read_text_vector_body(fh)
            trans = _read_text_matrix_body(fh)
            psi = _read_text_vector_body(fh)
        end = _read_token(fh)
        if end != "</Plda>":
            raise KaldiFormatError(f"expected </Plda>, got {end!r}")
    finally:
        if close:
            fh.close()
    diagP, diagQ = plda_psi_to_pq(psi)
    return {"plda_mean": mean, "diagonalizing_transform": trans, "Psi_across_covar_diag": psi,
            "diagP": diagP, "diagQ": diagQ}


def read_vector_ark(f):
    """Iterate (key, float32 vector) over a Kaldi vector archive (binary or text `key [ v ]` lines) —
    the x-vector ark format dataprep_*.py read through kaldi_io.read_vec_flt (dataprep_sre.py:152-167)."""
    fh, close = _open(f)
    try:
        while True:
            key = _read_token(fh)
            if key == "":
                return
            if _is_binary(fh):
                vec = _read_binary_vector_body(fh, _read_token(fh))
            else:
                vec = _read_text_vector_body(fh)
            yield key, vec.astype(np.float32)
    finally:
        if close:
            fh.close()


# ---- bulk x-vector readers: ark / scp -> ONE (N, D) float32 matrix ----------------------------
#
# dataprep_sre.py:152-167 builds the "mega dict" with one kaldi_io.read_vec_flt(rxfilename) call and one small numpy
# array per utterance, then pickles the dict (2-4 GB at VoxCeleb scale).  Here the archive is memory-mapped and the
# binary records are walked with a fixed-stride check: every `FV ` record of an x-vector ark has the same length, so
# after the keys are located all payloads are gathered with one strided numpy copy straight into a caller-provided
# (e.g. pinned) buffer — no per-utterance Python objects, no dict.

def read_scp(path):
    """Kaldi script file -> [(key, rxfilename)], rxfilename = 'file' or 'file:offset' (as dataprep_sre.py:158-160
    splits its lines: first blank separates key and rxfilename)."""
    out = []
    with open(path, "r") as fh:
        for ln in fh:
            ln = ln.rstrip("\n")
            if not ln.strip():
                continue
            key, _, rx = ln.partition(" ")
            out.append((key, rx.strip()))
    return out


def _split_rx(rx):
    """'path:offset' -> (path, offset) (offset None for a plain file; a ':' inside the path is kept)."""
    head, sep, tail = rx.rpartition(":")
    if sep and tail.isdigit():
        return head, int(tail)
    return rx, None


def _scp_file(f, scp_path):
    """An archive named by an scp entry: as written (Kaldi resolves against the working directory), else — for a relative
    name that is not there — next to the scp itself (an scp shipped in the same directory as its archive)."""
    if os.path.isabs(f) or os.path.exists(f):
        return f
    alt = os.path.join(os.path.dirname(os.path.abspath(scp_path)), f)
    return alt if os.path.exists(alt) else f


def _vec_at(buf, off):
    """float32 view of the (binary or text) vector whose object starts at byte `off` of `buf`."""
    if bytes(buf[off:off + 2]) == b"\0B":
        tok = bytes(buf[off + 2:off + 5])
        if tok not in (b"FV ", b"DV ") or buf[off + 5] != 4:
            raise KaldiFormatError(f"unexpected vector header {bytes(buf[off + 2:off + 10])!r}")
        n = int(np.frombuffer(buf, dtype="<i4", count=1, offset=off + 6)[0])
        dt = "<f4" if tok == b"FV " else "<f8"
        return np.frombuffer(buf, dtype=dt, count=n, offset=off + 10)
    end = bytes(buf[off:off + (1 << 16)]).find(b"]")
    return _read_text_vector_body(io.BytesIO(bytes(buf[off:off + end + 1])))


def read_vector_scp(path):
    """Iterate (key, float32 vector) over a Kaldi scp whose entries point at vectors ('ark:offset' or one-vector files):
    the exact replacement of `{key: kaldi_io.read_vec_flt(rx)}` at dataprep_sre.py:160."""
    maps = {}
    for key, rx in read_scp(path):
        f, off = _split_rx(rx)
        if f not in maps:
            maps[f] = np.memmap(_scp_file(f, path), dtype=np.uint8, mode="r")
        yield key, np.asarray(_vec_at(maps[f], off or 0), dtype=np.float32)


def _out_buffer(out, n, dim):
    if out is None:
        return np.empty((n, dim), dtype=np.float32)
    arr = out.numpy() if hasattr(out, "numpy") else out
    if arr.dtype != np.float32 or arr.ndim != 2 or arr.shape[0] < n or arr.shape[1] != dim:
        raise ValueError(f"out must be a float32 ({n}+, {dim}) buffer")
    return arr[:n]


def load_vector_ark(path, out=None):
    """Whole binary vector ark -> (keys, (N, D) float32 matrix).  `out`: optional pre-allocated float32 array or CPU
    torch tensor with >= N rows (pinned memory makes the following host-to-device copy asynchronous).  Falls back to
    the record-by-record reader for text archives or ragged records."""
    buf = np.memmap(path, dtype=np.uint8, mode="r")
    total = buf.shape[0]
    keys, offs = [], []
    pos = 0
    dim = None
    fast = True
    view = memoryview(buf)
    while pos < total:
        sp = bytes(view[pos:pos + 4096]).find(b" ")
        if sp < 0:
            break
        key = bytes(view[pos:pos + sp]).decode("ascii").strip()
        p = pos + sp + 1
        if bytes(view[p:p + 2]) != b"\0B" or bytes(view[p + 2:p + 5]) != b"FV " or view[p + 5] != 4:
            fast = False
            break
        n = int(np.frombuffer(buf, dtype="<i4", count=1, offset=p + 6)[0])
        if dim is None:
            dim = n
        elif n != dim:
            fast = False
            break
        keys.append(key)
        offs.append(p + 10)
        pos = p + 10 + 4 * n
    if not fast or dim is None:
        pairs = list(read_vector_ark(path))
        if not pairs:
            return [], np.zeros((0, 0), dtype=np.float32)
        mat = _out_buffer(out, len(pairs), pairs[0][1].shape[0])
        for i, (_, v) in enumerate(pairs):
            mat[i] = v
        return [k for k, _ in pairs], mat
    mat = _out_buffer(out, len(keys), dim)
    offs = np.asarray(offs, dtype=np.int64)
    idx = offs[:, None] + np.arange(4 * dim, dtype=np.int64)[None, :]
    # gather in slabs so the index matrix stays small (64k rows x 2 KB)
    step = 1 << 16
    m8 = mat.view(np.uint8).reshape(len(keys), 4 * dim)
    for lo in range(0, len(keys), step):
        m8[lo:lo + step] = buf[idx[lo:lo + step]]
    return keys, mat


def load_vector_scp(path, out=None):
    """Kaldi scp of x-vectors -> (keys, (N, D) float32 matrix), each archive memory-mapped once.  Entries of the form
    'ark:offset' that point at binary float vectors of one common length are gathered with one strided copy per
    archive; anything else goes through read_vector_scp."""
    entries = read_scp(path)
    if not entries:
        return [], np.zeros((0, 0), dtype=np.float32)
    by_file = {}
    for i, (_, rx) in enumerate(entries):
        f, off = _split_rx(rx)
        by_file.setdefault(f, []).append((i, off or 0))
    first_f, first_off = _split_rx(entries[0][1])
    probe = np.memmap(_scp_file(first_f, path), dtype=np.uint8, mode="r")
    dim = int(_vec_at(probe, first_off or 0).shape[0])
    mat = _out_buffer(out, len(entries), dim)
    m8 = mat.view(np.uint8).reshape(len(entries), 4 * dim)
    for f, lst in by_file.items():
        buf = np.memmap(_scp_file(f, path), dtype=np.uint8, mode="r")
        rows = np.asarray([i for i, _ in lst], dtype=np.int64)
        offs = np.asarray([o for _, o in lst], dtype=np.int64)
        hdr = buf[offs[:, None] + np.arange(10, dtype=np.int64)[None, :]]
        ok = (hdr[:, :6] == np.frombuffer(b"\0BFV \x04", dtype=np.uint8)).all() and \
            (hdr[:, 6:10].copy().view("<i4")[:, 0] == dim).all()
        if ok:
            step = 1 << 16
            for lo in range(0, len(rows), step):
                idx = offs[lo:lo + step, None] + 10 + np.arange(4 * dim, dtype=np.int64)[None, :]
                m8[rows[lo:lo + step]] = buf[idx]
        else:
            for i, o in lst:
                v = np.asarray(_vec_at(buf, o), dtype=np.float32)
                if v.shape[0] != dim:
                    raise KaldiFormatError(f"{entries[i][0]}: vector of length {v.shape[0]}, expected {dim}")
                mat[i] = v
    return [k for k, _ in entries], mat


def write_vector_ark(ark_path, keys, mat, scp_path=None):
    """Binary float-vector archive 'key \\0BFV \\x04<n>payload' per row (what copy-vector writes), optionally with the
    matching scp ('key ark_path:offset', offset = position of the \\0B marker).  One buffer, one write."""
    mat = np.ascontiguousarray(mat, dtype="<f4")
    n, dim = mat.shape
    if len(keys) != n:
        raise ValueError("one key per row")
    hdr = b"\0BFV \x04" + struct.pack("<i", dim)
    chunks, offsets, pos = [], [], 0
    for i, k in enumerate(keys):
        kb = k.encode("ascii") + b" "
        offsets.append(pos + len(kb))
        chunks.append(kb + hdr)
        chunks.append(mat[i].tobytes())
        pos += len(kb) + len(hdr) + 4 * dim
    with open(ark_path, "wb") as fh:
        fh.write(b"".join(chunks))
    if scp_path is not None:
        with open(scp_path, "w") as fh:
            fh.write("".join(f"{k} {ark_path}:{o}\n" for k, o in zip(keys, offsets)))
    return offsets


# ---- Kaldi initialisation of the model classes --------------------------------------------------------------------

def fold_init(model, mean_vec_file, transform_mat_file, plda_file=None):
    """Write a Kaldi LDA (+ PLDA) into a model's parameters the way the reference's loaders do
    (NeuralPlda.LoadPldaParamsFromKaldi utils/models.py:441-457, DPlda.LoadParamsFromKaldi :551-564,
    GaussianBackend.LoadPldaParamsFromKaldi :653-658), reading the files natively:
      centering_and_LDA            <- T[:, :-1],  T[:, -1] - T[:, :-1] mean           (mean subtraction folded into the bias)
      centering_and_wccn_plda      <- D,  -D plda_mean                                 (only with a PLDA file)
      P_sqrt, Q                    <- sqrt(diagP), diagQ of plda_psi_to_pq(Psi)
    The parameters are written through `.data` like the reference does and their version counters are bumped, so cached
    parameter images are rebuilt."""
    import torch
    T = read_matrix(transform_mat_file)
    mean = read_vector(mean_vec_file)
    new = {"centering_and_LDA.weight": T[:, :-1], "centering_and_LDA.bias": T[:, -1] - T[:, :-1].dot(mean)}
    if plda_file is not None:
        plda = read_plda(plda_file)
        D = plda["diagonalizing_transform"]
        new.update({"centering_and_wccn_plda.weight": D, "centering_and_wccn_plda.bias": -D.dot(plda["plda_mean"]),
                    "P_sqrt": np.sqrt(plda["diagP"]), "Q": plda["diagQ"]})
    sd = model.state_dict()
    for name, val in new.items():
        sd[name].data.copy_(torch.from_numpy(np.ascontiguousarray(val)).float())
    params = dict(model.named_parameters())
    for name in new:
        if name in params:
            torch.autograd.graph.increment_version(params[name])


# ---- writers (used by tests and by tools that synthesise Kaldi-format fixtures) -----------------

def _bin_vec(v, double):
    v = np.asarray(v)
    dt, tok = ("<f8", b"DV ") if double else ("<f4", b"FV ")
    return tok + b"\x04" + struct.pack("<i", v.shape[0]) + v.astype(dt).tobytes()


def _bin_mat(m, double):
    m = np.asarray(m)
    dt, tok = ("<f8", b"DM ") if double else ("<f4", b"FM ")
    return tok + b"\x04" + struct.pack("<i", m.shape[0]) + b"\x04" + struct.pack("<i", m.shape[1]) + \
        np.ascontiguousarray(m).astype(dt).tobytes()


def write_vector_binary(path, v, double=False):
    with open(path, "wb") as fh:
        fh.write(b"\0B" + _bin_vec(v, double))


def write_matrix_binary(path, m, double=False):
    with open(path, "wb") as fh:
        fh.write(b"\0B" + _bin_mat(m, double))


def write_plda_binary(path, mean, transform, psi):
    with open(path, "wb") as fh:
        fh.write(b"\0B<Plda> " + _bin_vec(mean, True) + _bin_mat(transform, True) + _bin_vec(psi, True) +
                 b"</Plda> ")
