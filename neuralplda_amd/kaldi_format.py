"""Native readers for the Kaldi objects the NPLDA initialisation consumes.

The reference shells out to Kaldi binaries (`copy-matrix`, `copy-vector`, `ivector-copy-plda`:
utils/models.py:442-448, utils/Kaldi2NumpyUtils/kaldiPlda2numpydict.py:17) and parses their text
output.  This module reads the same files directly — binary (`\\0B` + `FM `/`DM `/`FV `/`DV `
tokens, `\\x04`+int32 sizes, little-endian payload) or Kaldi text (`[ ... ]`) — so no Kaldi
installation is needed.  It is host-side file I/O only.
"""
import io
import struct

import numpy as np

__all__ = ["read_vector", "read_matrix", "read_plda", "plda_psi_to_pq", "read_vector_ark",
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
