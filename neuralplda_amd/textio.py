"""Native text I/O of the trial-list path: thin numpy wrappers over the host-side entry points of libnplda_hip.so
(nplda_text_scan / nplda_text_lookup / nplda_scores_write, csrc/nplda_textio.cpp).

They replace np.genfromtxt + per-trial Python loops + np.savetxt of utils/sv_trials_loaders.py:377-383, :400-406,
:429-437 and utils/scorefile_generator.py:26-38, :45-55 — byte-compatible, ~10^3 x faster (a 500 k-trial score file:
~7 s of numpy/Python text handling in the reference's way, tens of ms here)."""
import ctypes

import numpy as np

from . import _lib

__all__ = ["IdBlob", "scan", "lookup", "write_scores", "format_f32", "format_f64", "row_tokens", "column_f64",
           "count_unique", "column_tokens"]

RAW, SPLITEXT, BASENAME_SPLITEXT, STRIP_SPH = 0, 1, 2, 3


class IdBlob:
    """An id -> number table in the layout nplda_text_lookup reads: ids joined by '\\n', start offsets, numbers."""

    def __init__(self, ids, nums=None):
        ids = list(ids)
        self.n = len(ids)
        self.blob = "\n".join(ids).encode("utf-8")
        raw = np.frombuffer(self.blob, dtype=np.uint8)
        nl = np.flatnonzero(raw == 10).astype(np.int64)
        if self.n and len(nl) != self.n - 1:
            raise ValueError("ids must not contain newlines")
        off = np.empty(self.n + 1, dtype=np.int64)
        off[0] = 0
        if self.n:
            off[1:self.n] = nl + 1
            off[self.n] = len(self.blob) + 1
        self.off = off
        self.nums = None if nums is None else np.ascontiguousarray(nums, dtype=np.int64)
        if self.nums is not None and len(self.nums) != self.n:
            raise ValueError("one number per id")

    @classmethod
    def from_dict(cls, id_to_num):
        return cls(id_to_num.keys(), np.fromiter(id_to_num.values(), dtype=np.int64, count=len(id_to_num)))


def _text(text):
    if isinstance(text, str):
        text = text.encode("utf-8")
    return text


def scan(text):
    """(rows, ncols) of a whitespace-separated table; ValueError on ragged rows (as np.genfromtxt)."""
    text = _text(text)
    ncols = ctypes.c_int(0)
    rows = _lib.load().nplda_text_scan(text, len(text), ctypes.byref(ncols))
    if rows < 0:
        raise ValueError("rows of the trials file have different numbers of columns")
    return int(rows), int(ncols.value)


def lookup(text, idblob, skip_rows=0, mode1=RAW, mode2=RAW, label_col=-1, rows=None):
    """-> (i1, i2, label or None, row_of, first_bad_row).  Rows that do not resolve are skipped (see the header)."""
    text = _text(text)
    if rows is None:
        rows, _ = scan(text)
    cap = max(rows - skip_rows, 0)
    i1 = np.empty(cap, dtype=np.int64)
    i2 = np.empty(cap, dtype=np.int64)
    lab = np.empty(cap, dtype=np.float32) if label_col >= 0 else None
    row_of = np.empty(cap, dtype=np.int64)
    kept, bad = ctypes.c_int64(0), ctypes.c_int64(-1)
    code = _lib.load().nplda_text_lookup(
        text, len(text), skip_rows, mode1, mode2, label_col, idblob.blob, idblob.off.ctypes.data,
        None if idblob.nums is None else idblob.nums.ctypes.data, idblob.n, i1.ctypes.data, i2.ctypes.data,
        None if lab is None else lab.ctypes.data, row_of.ctypes.data, ctypes.byref(kept), ctypes.byref(bad))
    _lib.check(code, "nplda_text_lookup")
    k = int(kept.value)
    return i1[:k], i2[:k], (None if lab is None else lab[:k]), row_of[:k], int(bad.value)


def write_scores(path, text, scores, skip_rows=0, keep_cols=2, header=None):
    """float32 scores are written as str(np.float32), float64 scores as str(np.float64)."""
    text = _text(text)
    scores = np.asarray(scores)
    f64 = scores.dtype == np.float64
    scores = np.ascontiguousarray(scores, dtype=np.float64 if f64 else np.float32)
    code = _lib.load().nplda_scores_write(str(path).encode(), text, len(text), skip_rows, keep_cols,
                                          None if header is None else header.encode("utf-8"),
                                          scores.ctypes.data, int(f64), len(scores))
    _lib.check(code, "nplda_scores_write")


def format_f32(v):
    buf = ctypes.create_string_buffer(32)
    n = _lib.load().nplda_format_f32(float(np.float32(v)), buf)
    return buf.raw[:n].decode()


def format_f64(v):
    buf = ctypes.create_string_buffer(32)
    n = _lib.load().nplda_format_f64(float(v), buf)
    return buf.raw[:n].decode()


def column_f64(text, col, n, skip_rows=0):
    """Column `col` (negative: from the end) of the n data rows after skip_rows as float64; ValueError if a token is
    not a number (as ndarray.astype(float) raises)."""
    text = _text(text)
    out = np.empty(n, dtype=np.float64)
    code = _lib.load().nplda_text_column_f64(text, len(text), skip_rows, col, out.ctypes.data, n)
    if code != 0:
        raise ValueError("could not convert a score column to float (or the table changed size)")
    return out


def count_unique(text, col, skip_rows=0):
    text = _text(text)
    k = ctypes.c_int64(0)
    _lib.check(_lib.load().nplda_text_count_unique(text, len(text), skip_rows, col, ctypes.byref(k)),
               "nplda_text_count_unique")
    return int(k.value)


def column_tokens(text, col, n, skip_rows=0, stride=1):
    """Tokens of column `col` in rows skip_rows + k * stride, k < n, as a list of str."""
    text = _text(text)
    st = np.empty(n, dtype=np.int64)
    ln = np.empty(n, dtype=np.int64)
    _lib.check(_lib.load().nplda_text_column_spans(text, len(text), skip_rows, col, stride, st.ctypes.data,
                                                   ln.ctypes.data, n), "nplda_text_column_spans")
    return [text[a:a + b].decode("utf-8") for a, b in zip(st.tolist(), ln.tolist())]


def row_tokens(text, row):
    """Tokens of data row `row` (0-based over non-blank, comment-stripped lines) — slow path for error messages."""
    k = -1
    for line in _text(text).split(b"\n"):
        toks = line.split(b"#", 1)[0].split()
        if toks:
            k += 1
            if k == row:
                return [t.decode("utf-8", "replace") for t in toks]
    raise IndexError(row)
