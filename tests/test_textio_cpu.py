"""Host-side text I/O (csrc/nplda_textio.cpp) against the reference's way of doing the same thing:
np.genfromtxt(dtype=str) + per-trial dict look-ups (utils/sv_trials_loaders.py:377-383, :400-406, :429-437) and
astype(str) + np.savetxt (utils/scorefile_generator.py:36-38, :53-55).  Byte-exact: no tolerance."""
import os

import numpy as np
import pytest

from neuralplda_amd import textio


def _ref_rows(path):
    t = np.genfromtxt(path, dtype="str")
    return t.reshape(1, -1) if t.ndim == 1 else t


def _make_file(tmp_path, rng, n, ids, with_noise=True):
    lines = []
    for k in range(n):
        a, b = ids[rng.integers(len(ids))], ids[rng.integers(len(ids))]
        lab = ["1", "0", "1.0", "0.0", "+1", "1e0", "nan"][rng.integers(7)] if with_noise else str(rng.integers(2))
        style = rng.integers(5) if with_noise else 0
        if style == 0:
            lines.append(f"{a}\t{b}\t{lab}")
        elif style == 1:
            lines.append(f"  {a}   /some/dir/{b}.wav \t {lab}  ")
        elif style == 2:
            lines.append(f"{a} {b}.sph {lab}\r")
        elif style == 3:
            lines.append(f"UNKNOWN-{k}\t{b}\t{lab}   # a comment")
        else:
            lines.append(f"{a}\t{b}\tnot_a_number")
        if with_noise and rng.random() < 0.05:
            lines.append("")
        if with_noise and rng.random() < 0.03:
            lines.append("# only a comment")
    p = tmp_path / "trials.tsv"
    p.write_text("\n".join(lines) + ("\n" if rng.random() < 0.5 else ""))
    return str(p)


def _ref_lookup(rows, id_to_num, key1, key2, label=True):
    x1, x2, l, src = [], [], [], []
    for r, tr in enumerate(rows):
        try:
            a, b = id_to_num[key1(tr[0])], id_to_num[key2(tr[1])]
            c = float(tr[2]) if label else 0.0
            x1.append(a); x2.append(b); l.append(c); src.append(r)
        except Exception:
            pass
    return np.asarray(x1, np.int64), np.asarray(x2, np.int64), np.asarray(l, np.float32), np.asarray(src, np.int64)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_lookup_matches_genfromtxt_and_dict_loop(tmp_path, seed):
    rng = np.random.default_rng(seed)
    ids = [f"spk{u // 3:03d}-utt{u:04d}" for u in range(300)] + ["a.b", ".hidden", "x/y.z", "dup", "dup"]
    id_to_num = {u: 7 * i + 3 for i, u in enumerate(ids)}  # arbitrary numbers; "dup": last wins
    path = _make_file(tmp_path, rng, 2000, ids)
    rows = _ref_rows(path)
    text = open(path, "rb").read()
    assert textio.scan(text) == (rows.shape[0], rows.shape[1])
    blob = textio.IdBlob.from_dict(id_to_num)
    ident = lambda d: d  # noqa: E731
    ext = lambda d: os.path.splitext(d)[0]  # noqa: E731
    base = lambda d: os.path.splitext(os.path.basename(d))[0]  # noqa: E731
    for m1, m2, k1, k2 in ((textio.RAW, textio.RAW, ident, ident), (textio.RAW, textio.SPLITEXT, ident, ext),
                           (textio.BASENAME_SPLITEXT, textio.BASENAME_SPLITEXT, base, base)):
        r1, r2, rl, rsrc = _ref_lookup(rows, id_to_num, k1, k2)
        i1, i2, lab, src, bad = textio.lookup(text, blob, 0, m1, m2, label_col=2)
        assert np.array_equal(i1, r1) and np.array_equal(i2, r2) and np.array_equal(src, rsrc)
        assert np.array_equal(lab, rl, equal_nan=True)
        expect_bad = next((r for r in range(len(rows)) if r not in set(rsrc.tolist())), -1)
        assert bad == expect_bad
    # skip_rows and no label column
    r1, r2, _, rsrc = _ref_lookup(rows[5:], id_to_num, base, base, label=False)
    i1, i2, lab, src, _ = textio.lookup(text, blob, 5, textio.BASENAME_SPLITEXT, textio.BASENAME_SPLITEXT)
    assert lab is None and np.array_equal(i1, r1) and np.array_equal(i2, r2) and np.array_equal(src, rsrc)


def test_id_normalisation_edge_cases():
    cases = ["a.wav", "/d/a.wav", "d.e/f", "d.e/f.g.h", ".bashrc", "/x/.bashrc", "..x", "...", "a.", "a..b", "/", "a/",
             "no_ext", "x/.y.z", "trailing.dot."]
    keys = sorted({os.path.splitext(c)[0] for c in cases} | {os.path.splitext(os.path.basename(c))[0] for c in cases}
                  | set(cases))
    keys = [k for k in keys if k and not k.isspace()]
    blob = textio.IdBlob(keys)
    pos = {k: i for i, k in enumerate(keys)}
    text = "\n".join(f"{c} {c} 1" for c in cases if c.strip())
    rows = [c for c in cases if c.strip()]
    for mode, fn in ((textio.RAW, lambda d: d), (textio.SPLITEXT, lambda d: os.path.splitext(d)[0]),
                     (textio.BASENAME_SPLITEXT, lambda d: os.path.splitext(os.path.basename(d))[0])):
        i1, i2, _, src, _ = textio.lookup(text, blob, 0, mode, mode, label_col=2)
        want = [(r, pos[fn(c)]) for r, c in enumerate(rows) if fn(c) in pos]
        assert list(zip(src.tolist(), i1.tolist())) == want and np.array_equal(i1, i2)


def test_scan_rejects_ragged_and_handles_empty(tmp_path):
    with pytest.raises(ValueError):
        textio.scan("a b c\nd e\n")
    assert textio.scan("") == (0, 0)
    assert textio.scan("\n\n# c\n") == (0, 0)
    assert textio.scan("a b") == (1, 2)
    i1, i2, lab, src, bad = textio.lookup("", textio.IdBlob(["a"]), label_col=2)
    assert len(i1) == 0 and bad == -1
    i1, _, _, _, bad = textio.lookup("a a 1\n", textio.IdBlob([]), label_col=2)
    assert len(i1) == 0 and bad == 0


def test_write_scores_matches_savetxt(tmp_path):
    rng = np.random.default_rng(5)
    ids = [f"id{u:05d}" for u in range(50)]
    path = _make_file(tmp_path, rng, 500, ids, with_noise=False)
    rows = _ref_rows(path)
    text = open(path, "rb").read()
    scores = np.concatenate([rng.standard_normal(490).astype(np.float32) * 3,
                             np.asarray([0.0, -0.0, 1e-5, 123456.789, 1e17, -1e-4, np.inf, np.nan, 1.0, 2.5e-8], np.float32)])
    # sre layout: first row is the header, every column kept, LLR appended
    ref = tmp_path / "ref_sre.tsv"
    np.savetxt(ref, np.c_[rows[1:], scores[:len(rows) - 1].astype(str)], header="\t".join(rows[0]) + "\tLLR", fmt="%s",
               delimiter="\t", comments="")
    out = tmp_path / "out_sre.tsv"
    textio.write_scores(out, text, scores[:len(rows) - 1], skip_rows=1, keep_cols=rows.shape[1],
                        header="\t".join(rows[0]) + "\tLLR")
    assert out.read_bytes() == ref.read_bytes()
    # voices layout: two columns + score, no header
    ref = tmp_path / "ref_vo.tsv"
    np.savetxt(ref, np.c_[rows[:, :2], scores[:len(rows)].astype(str)], fmt="%s", delimiter="\t", comments="")
    out = tmp_path / "out_vo.tsv"
    textio.write_scores(out, text, scores[:len(rows)], skip_rows=0, keep_cols=2)
    assert out.read_bytes() == ref.read_bytes()
    from neuralplda_amd import _lib
    with pytest.raises(_lib.NpldaHipError):  # more scores than data rows
        textio.write_scores(out, text, np.zeros(len(rows) + 1, np.float32))
    with pytest.raises(_lib.NpldaHipError):  # unwritable path
        textio.write_scores(tmp_path / "no_such_dir" / "x.tsv", text, scores[:3])


def test_format_f32_matches_numpy_str():
    rng = np.random.default_rng(1)
    vals = np.concatenate([rng.standard_normal(5000).astype(np.float32) * s for s in (1, 1e-4, 1e-6, 1e5, 1e15, 1e20)]
                          + [rng.integers(0, 2 ** 32, 20000, dtype=np.uint64).astype(np.uint32).view(np.float32),
                             np.asarray([0.0, -0.0, 1e-4, 9.9999e-5, 1e16, 9.99e15, 1.0, 100000.0, 1e-45, 3.4028235e38,
                                         np.inf, -np.inf, np.nan], np.float32)])
    ref = vals.astype(str)
    for v, r in zip(vals.tolist(), ref.tolist()):
        assert textio.format_f32(v) == r


def test_loaders_drop_and_count(tmp_path):
    """_read_trials through the native path == the reference loop on its own G8 fixture text."""
    from neuralplda_amd import sv_trials_loaders as svl
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g8_loaders.npz"), allow_pickle=True)
    ids = [str(u) for u in g["utt_ids"]]
    id_to_num = {u: i for i, u in enumerate(ids)}
    p = tmp_path / "tr.tsv"
    p.write_text(str(g["train_trials_text"]))
    rows = _ref_rows(str(p))
    r1, r2, rl, _ = _ref_lookup(rows, id_to_num, lambda d: d, lambda d: d)
    x1, x2, l, dropped = svl._read_trials(str(p), id_to_num, strip_ext_col2=False)
    assert np.array_equal(x1.numpy(), r1) and np.array_equal(x2.numpy(), r2) and np.array_equal(l.numpy(), rl)
    assert dropped == len(rows) - len(r1) == 1


def test_f64_columns_unique_spans_and_sph_mode(tmp_path):
    rng = np.random.default_rng(9)
    R, M = 7, 13
    rid = [f"enr{r}" if r % 2 else f"tst{r}.sph" for r in range(R)]
    vals = rng.standard_normal(R * M) * np.repeat(10.0 ** rng.integers(-8, 8, R), M)
    lines = ["modelid\tsegment\tside\tLLR"] + [f"{rid[k // M]}\tcoh{k % M:03d}\ta\t{float(vals[k])!r}" for k in range(R * M)]
    p = tmp_path / "coh.tsv"
    p.write_text("\n".join(lines) + "\n")
    ref = np.genfromtxt(p, dtype="str", skip_header=1)
    text = "\n".join(lines[1:]) + "\n"
    assert textio.scan(text) == ref.shape
    got = textio.column_f64(text, -1, R * M)
    assert np.array_equal(got, ref[:, -1].astype(float)) and np.array_equal(got, vals)
    assert np.array_equal(textio.column_f64(text, 3, R * M), got)
    assert textio.count_unique(text, 1) == len(np.unique(ref[:, 1])) == M
    assert textio.count_unique(text, 0) == R
    assert textio.column_tokens(text, 0, R, stride=M) == list(ref[:, 0].reshape(-1, M)[:, 0])
    with pytest.raises(ValueError):
        textio.column_f64(text, 1, R * M)  # not numbers
    with pytest.raises(ValueError):
        textio.column_f64(text, -1, R * M + 1)  # fewer rows than asked
    # '.sph' removal anywhere in the id (str.replace), both columns
    blob = textio.IdBlob([w.replace(".sph", "") for w in rid] + ["ab"])
    trial_text = "\n".join(f"{rid[a]} {rid[b]} x" for a, b in [(0, 1), (2, 3), (6, 0)]) + "\na.sphb a.sph.sphb x\n"
    i1, i2, _, _, bad = textio.lookup(trial_text, blob, 0, textio.STRIP_SPH, textio.STRIP_SPH)
    assert bad == -1 and i1.tolist() == [0, 2, 6, R] and i2.tolist() == [1, 3, 0, R]


def test_format_f64_and_f64_writer(tmp_path):
    rng = np.random.default_rng(2)
    vals = np.concatenate([rng.standard_normal(5000) * s for s in (1, 1e-4, 1e-6, 1e5, 1e15, 1e20, 1e-300, 1e300)]
                          + [rng.integers(0, 2 ** 63, 20000, dtype=np.uint64).view(np.float64),
                             np.asarray([0.0, -0.0, 1e-4, 9.9999e-5, 1e16, 9.99e15, 1.0, 1e5, 5e-324, 1.7976931348623157e308,
                                         np.inf, -np.inf, np.nan, 0.1, 1 / 3])])
    ref = vals.astype(str)
    for v, r in zip(vals.tolist(), ref.tolist()):
        assert textio.format_f64(v) == r
    rows = np.asarray([[f"e{k}", f"t{k}.sph", "a", "0.5"] for k in range(300)])
    text = "modelid\tsegmentid\tside\tLLR\n" + "\n".join("\t".join(r) for r in rows) + "\n"
    sc = vals[:300].copy()
    ref_p, out_p = tmp_path / "ref.tsv", tmp_path / "out.tsv"
    np.savetxt(ref_p, np.c_[rows[:, :-1], sc.astype(str)], header="\t".join(["modelid", "segmentid", "side", "LLR"]),
               fmt="%s", delimiter="\t")
    textio.write_scores(out_p, text, sc, skip_rows=1, keep_cols=3, header="# " + "\t".join(["modelid", "segmentid", "side", "LLR"]))
    assert out_p.read_bytes() == ref_p.read_bytes()


@pytest.mark.parametrize("threads", [2, 5, 13])
def test_threaded_chunks_give_the_single_thread_results(tmp_path, monkeypatch, threads):
    """The file is cut at line boundaries into one chunk per host thread; results must not depend on the thread count
    (forced here on a small file with blank lines, comments and dropped rows so that chunks start mid-structure)."""
    rng = np.random.default_rng(threads)
    ids = [f"spk{u // 3:03d}-utt{u:04d}" for u in range(200)]
    path = _make_file(tmp_path, rng, 3000, ids)
    text = open(path, "rb").read()
    blob = textio.IdBlob(ids)
    monkeypatch.setenv("NPLDA_TEXT_THREADS", "1")
    base_scan = textio.scan(text)
    base = textio.lookup(text, blob, 2, textio.BASENAME_SPLITEXT, textio.BASENAME_SPLITEXT, label_col=2)
    sc = rng.standard_normal(base_scan[0] - 2).astype(np.float32)
    textio.write_scores(tmp_path / "one.tsv", text, sc, skip_rows=2, keep_cols=2, header="a\tb\tLLR")
    nuniq = textio.count_unique(text, 1)
    monkeypatch.setenv("NPLDA_TEXT_THREADS", str(threads))
    assert textio.scan(text) == base_scan
    got = textio.lookup(text, blob, 2, textio.BASENAME_SPLITEXT, textio.BASENAME_SPLITEXT, label_col=2)
    for a, b in zip(base[:4], got[:4]):
        assert np.array_equal(a, b, equal_nan=True)
    assert got[4] == base[4]
    textio.write_scores(tmp_path / "many.tsv", text, sc, skip_rows=2, keep_cols=2, header="a\tb\tLLR")
    assert (tmp_path / "many.tsv").read_bytes() == (tmp_path / "one.tsv").read_bytes()
    assert textio.count_unique(text, 1) == nuniq
    with pytest.raises(ValueError):
        textio.scan(text + b"one two three four five\n")  # a ragged row in the last chunk
    # numeric column, several threads
    vals = rng.standard_normal(5000)
    t2 = "\n".join(f"e{k} c{k % 50} {float(v)!r}" for k, v in enumerate(vals)) + "\n"
    assert np.array_equal(textio.column_f64(t2, -1, 5000), vals)
    assert textio.count_unique(t2, 1) == 50
