"""SURVEY.md f2 remainder: Kaldi scp / ark readers and the ark -> resident x-vector table path that replaces
`{utt: kaldi_io.read_vec_flt(rx)}` + pickle (dataprep_sre.py:152-167).  Host-side only (CPU)."""
import os
import time

import numpy as np
import pytest
import torch

from neuralplda_amd import kaldi_format as kf
from neuralplda_amd import sv_trials_loaders as svl


def test_scp_and_ark_readers_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    M = rng.standard_normal((257, 512)).astype(np.float32)
    keys = [f"id{i:04d}-utt_{i % 7}" for i in range(257)]
    ark, scp = str(tmp_path / "xvector.1.ark"), str(tmp_path / "xvector.1.scp")
    offs = kf.write_vector_ark(ark, keys, M, scp)
    assert kf.read_scp(scp)[3] == (keys[3], f"{ark}:{offs[3]}")
    pairs = list(kf.read_vector_scp(scp))
    assert [k for k, _ in pairs] == keys and all(np.array_equal(v, M[i]) for i, (_, v) in enumerate(pairs))
    k2, m2 = kf.load_vector_scp(scp)
    assert k2 == keys and np.array_equal(m2, M)
    k3, m3 = kf.load_vector_ark(ark)
    assert k3 == keys and np.array_equal(m3, M)
    # a shuffled sub-list of the scp (what a trial-specific scp looks like) and a text-format archive
    sub = [5, 200, 17, 0]
    with open(tmp_path / "sub.scp", "w") as fh:
        fh.write("".join(f"{keys[i]} {ark}:{offs[i]}\n" for i in sub))
    k4, m4 = kf.load_vector_scp(str(tmp_path / "sub.scp"))
    assert k4 == [keys[i] for i in sub] and np.array_equal(m4, M[sub])
    with open(tmp_path / "text.ark", "w") as fh:
        for i in range(3):
            fh.write(f"{keys[i]}  [ " + " ".join(repr(float(v)) for v in M[i]) + " ]\n")
    k5, m5 = kf.load_vector_ark(str(tmp_path / "text.ark"))
    assert k5 == keys[:3] and np.allclose(m5, M[:3], rtol=1e-7)
    # double-precision records and ragged archives fall back to the record reader / fail loudly
    with open(tmp_path / "dv.ark", "wb") as fh:
        fh.write(b"a " + b"\0B" + kf._bin_vec(M[0], True) + b"b " + b"\0B" + kf._bin_vec(M[1], True))
    k6, m6 = kf.load_vector_ark(str(tmp_path / "dv.ark"))
    assert k6 == ["a", "b"] and np.array_equal(m6, M[:2])


def test_table_from_a_100k_vector_ark_replaces_the_mega_dict(tmp_path):
    """>= 100 000 x-vectors: archive -> XvectorTable without a dict, accepted by the loaders in place of mega_dict."""
    rng = np.random.default_rng(1)
    n, D = 120000, 512
    M = rng.standard_normal((n, D), dtype=np.float32)
    keys = [f"spk{i // 40:05d}-utt{i:07d}" for i in range(n)]
    half = n // 2
    a1, s1 = str(tmp_path / "x.1.ark"), str(tmp_path / "x.1.scp")
    a2, s2 = str(tmp_path / "x.2.ark"), str(tmp_path / "x.2.scp")
    kf.write_vector_ark(a1, keys[:half], M[:half], s1)
    kf.write_vector_ark(a2, keys[half:], M[half:], s2)
    t0 = time.perf_counter()
    tab = svl.XvectorTable.from_scp(s1, s2)
    dt = time.perf_counter() - t0
    assert len(tab) == n and tab.dim == D and np.array_equal(tab.host, M)
    assert dt < 30.0  # one strided gather per archive, no per-utterance Python objects (kaldi_io: ~1e4 vectors/s)
    tab2 = svl.XvectorTable.from_ark(a1, a2)
    assert tab2.ids == keys and np.array_equal(tab2.host, M)
    # dict protocol of the reference's scripts: list(mega) / mega[utt] / len / in
    assert list(tab)[:3] == keys[:3] and keys[77] in tab and np.array_equal(tab[keys[77]], M[77])
    num_to_id = {i: j for i, j in enumerate(list(tab))}  # xvector_NeuralPlda_pytorch.py:120
    assert svl.xvector_table(tab) is tab
    d1, d2 = torch.tensor([5, 100000, 7]), torch.tensor([119999, 0, 7])
    X1, X2 = svl.load_xvec_trials_from_numbatch(tab, num_to_id, d1, d2, torch.device("cpu"))
    assert np.array_equal(X1.numpy(), M[[5, 100000, 7]]) and np.array_equal(X2.numpy(), M[[119999, 0, 7]])
    trials = np.asarray([[f"/a/b/{keys[9]}.wav", f"{keys[60001]}.sph"]])
    I1, I2 = svl.load_xvec_trials_from_idbatch(tab, trials, torch.device("cpu"))
    assert np.array_equal(I1.numpy()[0], M[9]) and np.array_equal(I2.numpy()[0], M[60001])
    with pytest.raises(KeyError):
        svl.load_xvec_trials_from_numbatch(tab, {0: "nope"}, torch.tensor([0]), torch.tensor([0]), torch.device("cpu"))
    # a key that occurs twice: the later archive wins, as mega_xvec_dict.update() does (dataprep_sre.py:162)
    kf.write_vector_ark(str(tmp_path / "dup.ark"), [keys[3]], M[10:11])
    tab3 = svl.XvectorTable.from_ark(a1, str(tmp_path / "dup.ark"))
    assert len(tab3) == half and np.array_equal(tab3[keys[3]], M[10])
