"""CPU-only tests: the C-ABI library loads and exports every declared symbol, the host-side mirror of
the reference interface (module layout, pickling, loaders, config, Kaldi readers, metrics) behaves like
the reference (golden vectors G1/G5/G8), and the product path refuses to compute without a HIP device."""
import io
import os
import pickle
import re

import numpy as np
import pytest
import torch

from oracle import nplda_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


class NC:
    def __init__(self, D0=512, D1=170, D2=170, beta=(99.0, 199.0), alpha=15.0, loss="SoftCdet"):
        self.xvector_dim, self.layer1_LDA_dim, self.layer2_PLDA_spkfactor_dim = D0, D1, D2
        self.beta, self.alpha, self.device, self.loss = list(beta), alpha, "cpu", loss


def test_library_exports_every_declared_symbol(hip_lib):
    hdr = open(os.path.join(ROOT, "include", "nplda_hip.h")).read()
    declared = set(re.findall(r"\b((?:nplda|gb)_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 15
    from neuralplda_amd import _lib
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(hip_lib, name), name
    # argument-free entry points work without a GPU
    assert hip_lib.nplda_abi_version() == 4
    assert hip_lib.nplda_max_dim() == 192
    assert hip_lib.nplda_padded_dim(150, 150) == 160 and hip_lib.nplda_padded_dim(170, 170) == 176
    assert hip_lib.nplda_packed_bytes(512, 150, 150) > 0 and hip_lib.nplda_packed_bytes(512, 500, 500) == 0
    assert hip_lib.nplda_loss_nsums(2, 0) == 10 and hip_lib.nplda_loss_nsums(1, 1) == 4 and hip_lib.nplda_loss_nsums(9, 0) == 0
    assert hip_lib.nplda_grad_floats(512, 170, 170) == 170 * 512 + 170 + 170 * 170 + 3 * 170
    assert b"invalid argument" in hip_lib.nplda_strerror(-22)


def test_module_layout_matches_reference():
    from neuralplda_amd import models
    g1 = np.load(os.path.join(G, "g1_kaldi_params.npz"))
    torch.manual_seed(0)
    m = models.NeuralPlda(NC())
    assert list(m.state_dict().keys()) == [str(k) for k in g1["state_dict_keys"]]
    assert m.threshold[99.0] is m.Th99 and m.threshold[199.0] is m.Th199      # dict aliases registered params
    assert m.centering_and_LDA.weight.shape == (170, 512) and m.P_sqrt.shape == (170,)
    assert float(m.alpha) == 15.0 and m.beta == [99.0, 199.0] and m.lossfn == "SoftCdet"
    assert len(list(m.parameters())) == 9                                         # what optim.Adam receives
    m2 = m.to(torch.device("cpu"))
    assert m2.threshold[99.0] is m2.Th99
    # sdsvc-style beta 9.9 registers as Th9 (utils/models.py:355-358)
    assert "Th9" in models.NeuralPlda(NC(beta=(9.9,))).state_dict()


def test_pickle_roundtrip_and_reference_class_path(tmp_path):
    from neuralplda_amd import compat, models
    m = models.NeuralPlda(NC(64, 24, 20))
    with torch.no_grad():
        m.Th99.fill_(-0.7)
    f = tmp_path / "NPLDA_1_0.pt"
    m.SaveModel(str(f))
    m2 = pickle.load(open(f, "rb"))
    assert isinstance(m2, models.NeuralPlda) and m2.threshold[99.0] is m2.Th99 and m2.Th99.item() == pytest.approx(-0.7)
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    # a pickle that names the reference's class path (utils.models.NeuralPlda) binds to this implementation
    compat.install()
    try:
        old = models.NeuralPlda.__module__
        models.NeuralPlda.__module__ = "utils.models"
        try:
            blob = pickle.dumps(m)
        finally:
            models.NeuralPlda.__module__ = old
        assert b"utils.models" in blob
        m3 = pickle.loads(blob)
        assert type(m3) is models.NeuralPlda and m3._reduce_sums is None
        import utils.models as um
        assert um.NeuralPlda is models.NeuralPlda and um.GaussianBackend is models.GaussianBackend
        from utils.sv_trials_loaders import load_xvec_trials_from_numbatch  # noqa: F401
        from utils.NpldaConf import NpldaConf  # noqa: F401
    finally:
        compat.uninstall()


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful on a box without a HIP device")
def test_no_cpu_fallback():
    from neuralplda_amd import _lib, models
    m = models.NeuralPlda(NC(64, 24, 20))
    x = torch.randn(4, 64)
    with pytest.raises(_lib.NpldaHipError):
        m(x, x)
    with pytest.raises(_lib.NpldaHipError):
        m.softcdet(torch.randn(4), torch.ones(4))
    with pytest.raises(_lib.NpldaHipError):
        m.extract_plda_embeddings(x)
    with pytest.raises(_lib.NpldaHipError):
        models.GaussianBackend(NC(32, 16, 16))(torch.randn(2, 32), torch.randn(2, 32))


def test_kaldi_readers_and_init(tmp_path):
    from neuralplda_amd import kaldi_format as kf, models
    rng = np.random.default_rng(5)
    T = rng.standard_normal((24, 65))
    mean = rng.standard_normal(64)
    pm, Dt, psi = rng.standard_normal(24), rng.standard_normal((24, 24)), rng.uniform(0.5, 30, 24)
    kf.write_matrix_binary(tmp_path / "transform.mat", T, double=False)
    open(tmp_path / "mean.vec", "w").write(" [ " + " ".join(repr(float(v)) for v in mean) + " ]\n")
    kf.write_plda_binary(tmp_path / "plda", pm, Dt, psi)
    np.testing.assert_allclose(kf.read_matrix(tmp_path / "transform.mat"), T.astype(np.float32))
    np.testing.assert_allclose(kf.read_vector(tmp_path / "mean.vec"), mean)
    pl = kf.read_plda(tmp_path / "plda")
    np.testing.assert_array_equal(pl["diagonalizing_transform"], Dt)
    # text forms of the same objects
    txt = " [\n" + " \n".join("  " + " ".join(repr(float(v)) for v in r) for r in T) + " ]\n"
    np.testing.assert_allclose(kf.read_matrix(txt.encode()), T)
    ptxt = ("<Plda>  [ " + " ".join(map(repr, pm.tolist())) + " ]\n" + " [\n" +
            " \n".join("  " + " ".join(map(repr, r)) for r in Dt.tolist()) + " ]\n [ " + " ".join(map(repr, psi.tolist())) +
            " ]\n</Plda> \n")
    np.testing.assert_allclose(kf.read_plda(ptxt.encode())["Psi_across_covar_diag"], psi)
    # vector archive (binary)
    ark = io.BytesIO()
    for k in ("utt1", "utt2"):
        ark.write(k.encode() + b" \0B" + b"FV \x04" + np.int32(4).tobytes() + np.arange(4, dtype="<f4").tobytes())
    got = list(kf.read_vector_ark(ark.getvalue()))
    assert [k for k, _ in got] == ["utt1", "utt2"] and got[1][1].tolist() == [0, 1, 2, 3]
    with pytest.raises(kf.KaldiFormatError):
        kf.read_matrix(b"\0BCM \x04")
    # LoadPldaParamsFromKaldi == the reference formulas (oracle restates utils/models.py:450-457)
    m = models.NeuralPlda(NC(64, 24, 24))
    m.LoadPldaParamsFromKaldi(str(tmp_path / "mean.vec"), str(tmp_path / "transform.mat"), str(tmp_path / "plda"))
    ref = orc.kaldi_init_params(T.astype(np.float32).astype(np.float64), mean, pm, Dt, psi)
    sd = m.state_dict()
    for key, arr in (("centering_and_LDA.weight", ref.W1), ("centering_and_LDA.bias", ref.b1),
                     ("centering_and_wccn_plda.weight", ref.W2), ("centering_and_wccn_plda.bias", ref.b2),
                     ("P_sqrt", ref.P_sqrt), ("Q", ref.Q)):
        np.testing.assert_allclose(sd[key].numpy(), arr, rtol=1e-6, atol=1e-7, err_msg=key)
    gb = models.GaussianBackend(NC(64, 24, 24))
    gb.LoadPldaParamsFromKaldi(str(tmp_path / "mean.vec"), str(tmp_path / "transform.mat"))
    np.testing.assert_allclose(gb.centering_and_LDA.bias.numpy(), ref.b1, rtol=1e-6, atol=1e-7)
    assert not gb.centering_and_LDA.weight.requires_grad and gb.paired_cov_inv_target.shape == (48, 48)


def test_loaders_match_reference_golden(tmp_path):
    """G8: same kept set, same first batch (identical RNG consumption), same gathered x-vectors."""
    from neuralplda_amd import sv_trials_loaders as L
    g = np.load(os.path.join(G, "g8_loaders.npz"))
    utt = [str(u) for u in g["utt_ids"]]
    id_to_num = {u: i for i, u in enumerate(utt)}
    num_to_id = {i: u for i, u in enumerate(utt)}
    mega = {u: g["xvec"][i] for i, u in enumerate(utt)}
    trf, vaf = tmp_path / "train_trials.tsv", tmp_path / "val_trials.tsv"
    trf.write_text(str(g["train_trials_text"]))
    vaf.write_text(str(g["val_trials_text"]))
    np.random.seed(1)
    torch.manual_seed(1)
    loader = L.combine_trials_and_get_loader([str(trf)], id_to_num, subsample_factors=[0.5], batch_size=32)
    assert len(loader.dataset) == int(g["n_train_dataset"]) and loader.dataset.dropped == 1
    d1, d2, t = next(iter(loader))
    assert d1.dtype == torch.int64 and t.dtype == torch.float32
    np.testing.assert_array_equal(d1.numpy(), g["batch_d1"])
    np.testing.assert_array_equal(d2.numpy(), g["batch_d2"])
    np.testing.assert_array_equal(t.numpy(), g["batch_t"])
    np.random.seed(2)
    torch.manual_seed(2)
    vd = L.get_trials_loaders_dict([str(vaf)], id_to_num, subsample_factors=[1.01], batch_size=50)
    assert list(vd.keys()) == [str(k) for k in g["val_key"]]
    v1, v2, vt = next(iter(vd[str(g["val_key"][0])]))
    assert len(vd[str(g["val_key"][0])].dataset) == int(g["n_val_dataset"])
    np.testing.assert_array_equal(v1.numpy(), g["val_d1"])
    np.testing.assert_array_equal(v2.numpy(), g["val_d2"])
    np.testing.assert_array_equal(vt.numpy(), g["val_t"])
    # gathers (CPU destination = host index-select, exactly the reference's values)
    X1, X2 = L.load_xvec_trials_from_numbatch(mega, num_to_id, d1, d2, torch.device("cpu"))
    np.testing.assert_array_equal(X1.numpy(), g["X1"])
    np.testing.assert_array_equal(X2.numpy(), g["X2"])
    I1, I2 = L.load_xvec_trials_from_idbatch(mega, g["idtrials"], torch.device("cpu"))
    np.testing.assert_array_equal(I1.numpy(), g["I1"])
    np.testing.assert_array_equal(I2.numpy(), g["I2"])
    e1, e2 = L.load_xvec_trials_from_idbatch(mega, g["idtrials"][:0], torch.device("cpu"))
    assert e1.shape == (0, 512) and e2.shape == (0, 512)
    with pytest.raises(KeyError):
        L.load_xvec_trials_from_idbatch(mega, np.asarray([["nope.wav", utt[0]]]), torch.device("cpu"))


def test_npldaconf(tmp_path):
    from neuralplda_amd.NpldaConf import NpldaConf
    from neuralplda_amd import scorefile_generator as sg
    cfg = tmp_path / "c.cfg"
    cfg.write_text("""[Paths]
training_data_trials_list = a.tsv,b.tsv
validation_trials_list = v.tsv
test_trials_list = t.tsv
mega_xvector_scp = x.scp
mega_xvector_pkl = x.pkl
meanvec = mean.vec
transformmat = transform.mat
kaldiplda = plda
[NPLDA]
xvector_dim = 512
layer1_LDA_dim = 170
layer2_PLDA_spkfactor_dim = 170
initialization = kaldi
device = cuda
seed = 1
alpha = 15
[Training]
train_subsample_factors=1.01,0.5
valid_subsample_factors=None
loss = softCdet
cmiss = 10
cfa = 1
target_probs = 0.01,0.005
batch_size = 2048
n_epochs = 20
lr = 0.0001
heldout_set_for_th_init = v
heldout_set_for_lr_decay = v
[Scoring]
scorefile_format = sre
[Logging]
log_interval = 1000
""")
    nc = NpldaConf(str(cfg))
    assert nc.beta == pytest.approx([9.9, 19.9]) and nc.training_data_trials_list == ["a.tsv", "b.tsv"]
    assert nc.train_subsample_factors == [1.01, 0.5] and nc.valid_subsample_factors is None
    assert nc.generate_scorefile is sg.generate_sre_scores and nc.loss == "softCdet" and nc.batch_size == 2048
    with pytest.raises(IOError):
        NpldaConf(str(tmp_path / "missing.cfg"))


def test_metrics_have_no_cpu_implementation():
    """minc / eer run on the HIP device only (nplda_detcost_sweep_f32); without one they fail loudly."""
    from neuralplda_amd import _lib, metrics
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    g = np.load(os.path.join(G, "g5_metrics.npz"))
    with pytest.raises(_lib.NpldaHipError):
        metrics.minc(torch.from_numpy(g["s"]), torch.from_numpy(g["t"]), [99.0, 199.0])


def test_abi_argument_validation_needs_no_gpu(hip_lib):
    """Every entry point validates its arguments before touching the device: bad arguments give NPLDA_E* codes and
    empty batches are successful no-ops — checked here without a GPU (nothing is launched)."""
    import ctypes
    EINVAL, EUNSUP = -22, -95
    dummy = ctypes.create_string_buffer(4096)
    p = ctypes.addressof(dummy)
    p16 = (p + 15) // 16 * 16
    L = hip_lib
    # empty batches: OK without any pointer
    assert L.nplda_score_pairs_f32(None, None, 0, 512, None, 512, 150, 150, None, None) == 0
    assert L.nplda_embed_f32(None, 0, 512, None, 512, 150, 150, None, 160, None, None) == 0
    assert L.nplda_score_indexed_f32(None, 160, None, 0, None, None, 0, None, 512, 150, 150, None, None) == 0
    assert L.nplda_gather_rows_f32(None, 512, 10, None, 0, 512, None, 512, None) == 0
    assert L.nplda_forward_train_f32(None, None, 0, 512, None, 512, 150, 150, None, None, None, None, 160, None) == 0
    assert L.gb_score_pairs_f32(None, None, 0, 512, None, 512, 170, None, None, None) == 0
    assert L.nplda_asnorm_apply_f64(None, None, None, 0, None, 5, None, None) == 0
    assert L.nplda_score_pairs_bf16x3(None, None, 0, 512, None, 512, 150, 150, None, None) == 0
    # negative sizes, null pointers, misaligned rows, unsupported dimensions
    assert L.nplda_score_pairs_f32(p16, p16, -1, 512, p16, 512, 150, 150, p16, None) == EINVAL
    assert L.nplda_score_pairs_f32(None, p16, 4, 512, p16, 512, 150, 150, p16, None) == EINVAL
    assert L.nplda_score_pairs_f32(p16 + 4, p16, 4, 512, p16, 512, 150, 150, p16, None) == EINVAL   # 16-byte alignment
    assert L.nplda_score_pairs_f32(p16, p16, 4, 510, p16, 512, 150, 150, p16, None) == EINVAL       # ldx < D0
    assert L.nplda_score_pairs_f32(p16, p16, 4, 512, p16, 510, 150, 150, p16, None) == EINVAL       # D0 % 4
    assert L.nplda_score_pairs_f32(p16, p16, 4, 512, p16, 512, 500, 150, p16, None) == EUNSUP
    assert L.nplda_embed_f32(p16, 4, 512, p16, 512, 150, 150, p16, 150, None, None) == EINVAL       # ldz < padded dim
    assert L.nplda_pack_params_f32(p16, p16, p16, p16, p16, p16, 512, 150, 150, p16, 16, None) == -28  # ENOSPC
    assert L.nplda_pack_params_f32(p16, p16, None, p16, p16, p16, 512, 150, 150, p16, 1 << 30, None) == EINVAL
    assert L.nplda_backward_f32(p16, p16, 4, 512, p16, 512, 150, 150, p16, p16, p16, p16, 160, p16, p16, 16, p16,
                                None) == -28
    assert L.nplda_loss_sums_f32(p16, p16, 4, None, 2, 15.0, 0, p16, None) == EINVAL
    th = (ctypes.c_void_p * 2)(p16, p16)
    assert L.nplda_loss_sums_f32(p16, p16, 4, th, 9, 15.0, 0, p16, None) == EUNSUP                  # K > 4
    assert L.nplda_loss_sums_f32(p16, p16, 4, th, 2, 15.0, 7, p16, None) == EINVAL                  # unknown loss kind
    assert L.nplda_cohort_stats_f32(p16, p16, 4, p16, p16, 0, 160, p16, 512, 150, 150, 500, 1, p16, p16, 4096, None) == EINVAL
    assert L.nplda_row_stats_f32(p16, 4, 2, 8, 500, 1, p16, None) == EINVAL                          # lds < M
    assert L.nplda_gather_rows_f32(p16, 512, 10, p16, 4, 510, p16, 512, None) == EINVAL
    assert L.gb_pack_params_f32(p16, p16, p16, p16, p16, p16, 512, 500, p16, 1 << 30, None) == EUNSUP
    assert L.nplda_adam_step_f32(None, None, None, None, None, 3, p16, 1e-3, 0.9, 0.999, 1e-8, 0.0, None) == EINVAL
    # entry points added later: moments, quadratic-form images, detection-cost sweep
    assert L.nplda_weighted_moments_f32(p16, -1, 340, 340, p16, None, p16, p16, p16, 0, p16, 1 << 30, None) == EINVAL
    assert L.nplda_weighted_moments_f32(p16, 8, 340, 342, p16, None, p16, p16, p16, 0, p16, 1 << 30, None) == EUNSUP   # n % 4
    assert L.nplda_weighted_moments_f32(p16, 8, 340, 388, p16, None, p16, p16, p16, 0, p16, 1 << 30, None) == EUNSUP   # n > 384
    assert L.nplda_weighted_moments_f32(p16, 8, 338, 340, p16, None, p16, p16, p16, 0, p16, 1 << 30, None) == EINVAL   # ldx < n
    assert L.nplda_weighted_moments_f32(p16, 8, 340, 340, p16, None, p16, p16, p16, 0, p16, 16, None) == -28
    assert L.nplda_moments_workspace_bytes(2048, 340) > 0 and L.nplda_moments_workspace_bytes(2048, 342) == 0
    assert L.gb_pack_quadform_f32(p16, p16, None, None, 0.0, 512, 170, p16, 1 << 30, None) == EINVAL
    assert L.gb_pack_dplda_f32(p16, p16, p16, None, 512, 170, p16, 1 << 30, None) == EINVAL
    assert L.gb_pack_dplda_f32(p16, p16, p16, p16, 512, 170, p16, 16, None) == -28
    assert L.gb_score_rows_f32(None, None, 0, 172, None, 172, 170, None, None) == 0
    assert L.gb_score_rows_f32(p16, p16, 4, 170, p16, 172, 170, p16, None) == EINVAL                 # ldy < D0
    betas = (ctypes.c_float * 9)(*([99.0] * 9))
    assert L.nplda_detcost_sweep_f32(p16, p16, -1, betas, 2, 0, p16, p16, p16, None, p16, 1 << 30, None) == EINVAL
    assert L.nplda_detcost_sweep_f32(p16, p16, 8, betas, 9, 0, p16, p16, p16, None, p16, 1 << 30, None) == EUNSUP      # K > 8
    assert L.nplda_detcost_workspace_bytes(-1) == 0  # (sizes of real inputs come from rocPRIM and need a device)
    # sizes
    assert L.nplda_backward_workspace_bytes(4096, 512, 150, 150) > 2 * 8192 * 160 * 4
    assert L.nplda_cohort_workspace_bytes(22000, 10000) == 22000 * 10000 * 4 + 256  # tile counters + score rows
    assert L.nplda_cohort_workspace_bytes(10 ** 7, 10 ** 4) <= (4 << 30) + 256
    assert L.gb_packed_bytes(512, 170) > 4 * 176 * 176 * 4 and L.nplda_bf16x3_packed_bytes(512, 150, 150) > 0


def test_trial_loader_fast_iterator_equals_torch_dataloader():
    """TrialLoader's vectorised iterator yields exactly the batches torch's DataLoader(shuffle=True) machinery yields
    under the same RNG state, for several epochs, including the short last batch."""
    from torch.utils.data import DataLoader
    from neuralplda_amd import sv_trials_loaders as svl
    n = 1000
    ds = svl.TrialIndexDataset(torch.arange(n), torch.arange(n) * 2, (torch.arange(n) % 3 == 0).float())
    for bs in (64, 1000, 1024, 7):
        torch.manual_seed(123)
        ref = DataLoader(ds, batch_size=bs, shuffle=True, collate_fn=ds.collate)
        ref_batches = [b for _ in range(3) for b in ref]
        ref_next = torch.rand(1)
        torch.manual_seed(123)
        fast = svl.TrialLoader(ds, batch_size=bs, shuffle=True, collate_fn=ds.collate)
        fast_batches = [b for _ in range(3) for b in fast]
        assert torch.equal(torch.rand(1), ref_next)  # the same amount of global RNG state was consumed
        assert len(fast) == len(ref) and len(fast_batches) == len(ref_batches)
        for a, b in zip(fast_batches, ref_batches):
            assert all(torch.equal(x, y) for x, y in zip(a, b))


def test_device_batches_packed_records():
    """TrialLoader.device_batches(pack=True): the same batches as the plain iterator under the same RNG state, plus one
    contiguous record [rows1 (int64) | rows2 (int64) | labels (float32)] per FULL batch (None for a short last batch) — what
    FusedTrainStep.step_rows stages with a single device copy.  The mapping num -> row is applied to both index columns."""
    from neuralplda_amd import sv_trials_loaders as svl
    n, bs = 1000, 64
    ds = svl.TrialIndexDataset(torch.arange(n), torch.arange(n) * 2 % n, (torch.arange(n) % 3 == 0).float())
    row_map = torch.arange(n, dtype=torch.int64) * 3 + 1
    torch.manual_seed(5)
    plain = list(svl.TrialLoader(ds, batch_size=bs, shuffle=True, collate_fn=ds.collate))
    torch.manual_seed(5)
    packed = list(svl.TrialLoader(ds, batch_size=bs, shuffle=True, collate_fn=ds.collate).device_batches("cpu", row_map, pack=True))
    assert len(packed) == len(plain) == (n + bs - 1) // bs
    for (d1, d2, t), (r1, r2, tl, rec) in zip(plain, packed):
        assert torch.equal(r1, row_map[d1.long()]) and torch.equal(r2, row_map[d2.long()]) and torch.equal(tl, t)
        if len(r1) == bs:
            assert rec.dtype == torch.uint8 and rec.numel() == 20 * bs and rec.is_contiguous()
            assert torch.equal(rec[:8 * bs].view(torch.int64), r1) and torch.equal(rec[8 * bs:16 * bs].view(torch.int64), r2)
            assert torch.equal(rec[16 * bs:].view(torch.float32), tl.float())
        else:
            assert rec is None
    assert packed[-1][3] is None and len(packed[-1][0]) == n % bs
    with pytest.raises(KeyError):
        bad = row_map.clone()
        bad[7] = -1
        list(svl.TrialLoader(ds, batch_size=bs, shuffle=True, collate_fn=ds.collate).device_batches("cpu", bad, pack=True))
    # device_epoch: the same epoch as (records of the full batches, the short last batch)
    torch.manual_seed(5)
    records, tail = svl.TrialLoader(ds, batch_size=bs, shuffle=True, collate_fn=ds.collate).device_epoch("cpu", row_map)
    assert records.shape == (n // bs, 20 * bs) and records.dtype == torch.uint8 and records.is_contiguous()
    for k in range(n // bs):
        assert torch.equal(records[k], packed[k][3])
    assert all(torch.equal(a, b) for a, b in zip(tail, packed[-1][:3]))
    torch.manual_seed(5)
    full, none = svl.TrialLoader(ds, batch_size=100, shuffle=True, collate_fn=ds.collate).device_epoch("cpu", row_map)
    assert full.shape == (10, 2000) and none is None


def test_odd_batch_size_has_no_packed_records():
    """Record k of an epoch starts at 20 * bs * k bytes: an odd batch size cannot hold aligned int64 fields.
    device_batches(pack=True) then yields the same batches with record None (the consumer copies the three views);
    device_epoch refuses with a clear error instead of a stride RuntimeError from deep inside torch."""
    from neuralplda_amd import sv_trials_loaders as svl
    n, bs = 50, 3
    ds = svl.TrialIndexDataset(torch.arange(n), torch.arange(n) * 2 % n, (torch.arange(n) % 3 == 0).float())
    torch.manual_seed(9)
    plain = list(svl.TrialLoader(ds, batch_size=bs, shuffle=True, collate_fn=ds.collate))
    torch.manual_seed(9)
    packed = list(svl.TrialLoader(ds, batch_size=bs, shuffle=True, collate_fn=ds.collate).device_batches("cpu", None, pack=True))
    assert len(packed) == len(plain)
    for (d1, d2, t), (r1, r2, tl, rec) in zip(plain, packed):
        assert torch.equal(r1, d1) and torch.equal(r2, d2) and torch.equal(tl, t) and rec is None
    with pytest.raises(ValueError, match="even batch_size"):
        svl.TrialLoader(ds, batch_size=bs, shuffle=True, collate_fn=ds.collate).device_epoch("cpu", None)


def test_mode_switch_drops_the_packed_image_cache():
    """NeuralPlda.invalidate_packed / train() / eval(): a `.data` write does not bump a parameter's version counter, so the
    mode switches drop the cached image (models._packed_for) unconditionally."""
    from neuralplda_amd import models

    class NC:
        xvector_dim, layer1_LDA_dim, layer2_PLDA_spkfactor_dim = 32, 16, 16
        beta, alpha, device, loss = [99.0], 15.0, "cpu", "SoftCdet"

    m = models.NeuralPlda(NC())
    v0 = m.Q._version
    m.Q.data.mul_(2.0)
    assert m.Q._version == v0  # the hazard: invisible to a version-keyed cache
    for switch in (m.eval, m.train, m.invalidate_packed):
        m._pack_cache["key"] = "stale"
        switch()
        assert m._pack_cache == {}


def test_order_free_device_epoch_keeps_the_rng_protocol():
    """device_batches(permute=False) — validate()'s form: the trials in file order, and the global generator left exactly
    where an ordinary (permuted) iteration leaves it, so everything drawn afterwards (the training shuffles) is unchanged."""
    from neuralplda_amd import sv_trials_loaders as svl
    n, bs = 1000, 128
    ds = svl.TrialIndexDataset(torch.arange(n), torch.arange(n) * 7 % n, (torch.arange(n) % 4 == 0).float())
    row_map = torch.arange(n, dtype=torch.int64) + 5
    torch.manual_seed(11)
    list(svl.TrialLoader(ds, batch_size=bs, shuffle=True, collate_fn=ds.collate))
    state_iter = torch.get_rng_state()
    torch.manual_seed(11)
    b_perm = list(svl.TrialLoader(ds, batch_size=bs, shuffle=True, collate_fn=ds.collate).device_batches("cpu", row_map))
    state_perm = torch.get_rng_state()
    torch.manual_seed(11)
    b_free = list(svl.TrialLoader(ds, batch_size=bs, shuffle=True, collate_fn=ds.collate).device_batches("cpu", row_map, permute=False))
    state_free = torch.get_rng_state()
    assert torch.equal(state_iter, state_perm) and torch.equal(state_iter, state_free)
    r1 = torch.cat([b[0] for b in b_free]); r2 = torch.cat([b[1] for b in b_free]); t = torch.cat([b[2] for b in b_free])
    assert torch.equal(r1, row_map[ds.x1.long()]) and torch.equal(r2, row_map[ds.x2.long()]) and torch.equal(t, ds.l)
    p1 = torch.cat([b[0] for b in b_perm])
    assert not torch.equal(p1, r1) and torch.equal(torch.sort(p1).values, torch.sort(r1).values)


def test_device_batches_sharded_for_data_parallel():
    """device_batches(shard=(rank, world)): every rank draws the same epoch and gets its dist.shard_bounds slice of each
    global batch plus the GLOBAL batch's [N_t, N_n] — what FusedTrainStep's one-collective step takes as global_counts."""
    from neuralplda_amd import dist as nd
    from neuralplda_amd import sv_trials_loaders as svl
    n, bs, world = 1003, 64, 3
    ds = svl.TrialIndexDataset(torch.arange(n), torch.arange(n) * 2 % n, (torch.arange(n) % 5 == 0).float())
    torch.manual_seed(11)
    plain = list(svl.TrialLoader(ds, batch_size=bs, shuffle=True, collate_fn=ds.collate).device_batches("cpu", None))
    per_rank = []
    for r in range(world):
        torch.manual_seed(11)
        per_rank.append(list(svl.TrialLoader(ds, batch_size=bs, shuffle=True, collate_fn=ds.collate)
                             .device_batches("cpu", None, shard=(r, world))))
    assert all(len(pr) == len(plain) for pr in per_rank)
    for k, (r1, r2, t) in enumerate(plain):
        B = len(r1)
        for r in range(world):
            lo, hi = nd.shard_bounds(B, world, r)
            s1, s2, st, gc = per_rank[r][k]
            assert torch.equal(s1, r1[lo:hi]) and torch.equal(s2, r2[lo:hi]) and torch.equal(st, t[lo:hi])
            assert gc.dtype == torch.float64 and gc.tolist() == [float(t.sum()), float(B - t.sum())]
        assert sum(len(per_rank[r][k][0]) for r in range(world)) == B
    with pytest.raises(ValueError):
        list(svl.TrialLoader(ds, batch_size=bs, shuffle=True, collate_fn=ds.collate).device_batches("cpu", None, pack=True, shard=(0, 2)))


def test_savemodel_after_make_data_parallel_with_an_explicit_group(tmp_path):
    """dist.make_data_parallel(model, group=g) keeps the ProcessGroup in model.__dict__['_dp_group']; a ProcessGroup
    does not pickle, so SaveModel / copy.deepcopy must drop it (as they drop the reduction closures)."""
    import copy
    import threading
    from neuralplda_amd import models

    class NC:
        xvector_dim, layer1_LDA_dim, layer2_PLDA_spkfactor_dim = 32, 16, 16
        beta, alpha, device, loss = [99.0], 15.0, "cpu", "SoftCdet"

    m = models.NeuralPlda(NC())
    m.__dict__["_dp_group"] = threading.Lock()  # stands in for a ProcessGroup: equally unpicklable
    m._reduce_sums = lambda v: v
    m._reduce_flat = lambda v: v
    m.SaveModel(str(tmp_path / "m.pt"))
    with open(tmp_path / "m.pt", "rb") as fh:
        m2 = pickle.load(fh)
    assert m2.__dict__["_dp_group"] is None and m2._reduce_flat is None
    m3 = copy.deepcopy(m)
    assert m3.__dict__["_dp_group"] is None
    assert m.__dict__["_dp_group"] is not None  # the live model keeps its group


def test_compat_fused_adam_leaves_cpu_parameters_to_torch():
    import neuralplda_amd.compat as compat
    real = torch.optim.Adam
    compat.install(fused_adam=True)
    try:
        o = torch.optim.Adam([torch.nn.Parameter(torch.zeros(4))], lr=1e-3, weight_decay=1e-5)
        assert type(o) is real
        o2 = torch.optim.Adam([{"params": [torch.nn.Parameter(torch.zeros(4))], "lr": 1e-2}], lr=1e-3)
        assert type(o2) is real and o2.param_groups[0]["lr"] == 1e-2
    finally:
        compat.uninstall()
    assert torch.optim.Adam is real
