"""Register-budget guards for two kernels whose speed depends on it (training step): the weight-gradient kernel
must keep TWO blocks resident per CU (its grid is sized for that: at 330 registers it silently ran as two rounds) and
the small-batch forward must not split its registers between VGPRs and AGPRs (146 cross-file moves per 96 MFMAs).
hipcc cross-compiles for gfx950 without a GPU, so this runs in the CPU suite."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "neuralplda_amd", "csrc")


def _resources(src):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    err = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                          "-I" + CSRC, "-c", os.path.join(CSRC, src), "-o", os.devnull,
                          "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=900).stderr
    out, cur = {}, None
    for line in err.splitlines():
        m = re.search(r"remark: +(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): +(\S+)", line)
        if not m:
            continue
        if m.group(1) == "Function Name":
            cur = out.setdefault(m.group(2), {})
        elif cur is not None:
            cur[m.group(1).split(" ")[0]] = int(m.group(2))
    return out


def test_register_budgets_of_the_training_kernels():
    res = _resources("nplda_backward.hip")
    wgrad = [v for k, v in res.items() if "wgrad_kernel" in k]
    assert len(wgrad) == 1
    assert wgrad[0]["Occupancy"] >= 2 and wgrad[0]["ScratchSize"] == 0 and wgrad[0]["VGPRs"] + wgrad[0]["AGPRs"] <= 256
    small = {k: v for k, v in res.items() if "nplda_fwd_small_kernel" in k}
    assert small, "the training-mode small forward is instantiated in this translation unit"
    for k, v in small.items():
        assert v["AGPRs"] == 0 and v["ScratchSize"] == 0, (k, v)
    # round 6: the one-kernel training steps at D = 150 and the streaming data gradient keep NOTHING in scratch.  (Until then
    # every training kernel had a 20-byte scratch object — the loss constants, written at a run-time offset by merged branch
    # tails — and the streaming data gradient spilled five 64-bit row pointers it reloaded, with an s_waitcnt vmcnt(0) each,
    # eight times per tile.)
    fb = {k: v for k, v in res.items() if "train_fb_small_kernelILi10E" in k or "train_fb_half_kernelILi10E" in k}
    assert len(fb) >= 10, sorted(res)
    for k, v in fb.items():
        assert v["ScratchSize"] == 0, (k, v)
    stream = {k: v for k, v in res.items() if "bwd_data_stream_kernelILi10E" in k}
    assert len(stream) == 2
    for k, v in stream.items():
        assert v["ScratchSize"] == 0 and v["Occupancy"] >= 2, (k, v)


def test_register_budget_of_the_cohort_gemm():
    """Three blocks per CU: <= 168 registers, no scratch (the stage loop has no room for spills)."""
    res = _resources("nplda_cohort.hip")
    gemm = [v for k, v in res.items() if "cohort_gemm_kernel" in k]
    assert len(gemm) == 1 and gemm[0]["Occupancy"] >= 3 and gemm[0]["ScratchSize"] == 0


def test_register_budget_of_the_fused_cohort_kernel():
    """cohort_fused2_kernel, the forms the library launches by default (8 waves per block, RGW = 2 for full row tiles, RGW = 1
    for calls with few rows): one 512-thread block per CU = two waves per SIMD = at most 256 registers, and NO scratch — a spill
    inside the tile loop is paid in matrix-pipe time.  The 16-wave forms (NPLDA_COHORT_NW=16, 128 registers) are opt-in and may
    spill a few bytes.  Round 6: the SPLIT forms (three bf16 pieces, six passes; the default at NB = 10 / 11) hold 3 x the
    operand registers per k: NB = 10 fits; NB = 11 with two row groups sits at 256 and keeps seventeen per-item values (68 bytes)
    in scratch (outside the MFMA loop: read in the assembly) — bounded here so that it does not grow unnoticed."""
    res = _resources("nplda_cohort_fused.hip")
    fused = {k: v for k, v in res.items() if "cohort_fused2_kernel" in k and "ELi8ELb0EEEvNS_9FusedArgsE" in k}
    assert len(fused) >= 8, sorted(res)
    for k, v in fused.items():
        assert v["ScratchSize"] == 0 and v["VGPRs"] + v["AGPRs"] <= 256 and v["Occupancy"] >= 2, (k, v)
    split = {k: v for k, v in res.items() if "cohort_fused2_kernel" in k and "ELi8ELb1EEEvNS_9FusedArgsE" in k}
    assert len(split) == 8, sorted(res)   # LOWEST x NB in {10, 11} x RGW in {1, 2}
    for k, v in split.items():
        tight = "ELi11ELi2E" in k
        assert v["ScratchSize"] <= (96 if tight else 0) and v["VGPRs"] + v["AGPRs"] <= 256 and v["Occupancy"] >= 2, (k, v)


def test_balanced_tile_kernels_keep_their_row_pointers_out_of_scratch():
    """nplda_fwd_mid_kernel at NB = 10 (D = 150), all eight forms (pair / embed, half tiles or not, rows by index or not): no
    scratch.  Round 6: the embedding form chose a row's table with `second ? a.xb : a.xa` on two pointer FIELDS of the by-value
    argument struct; hipcc compiled that as an indexed read of a scratch copy of the struct — sixteen scratch loads, each with
    an s_waitcnt vmcnt(0), at the head of every group (csrc/nplda_fwd_mid.h, mid_addr)."""
    res = _resources("nplda_forward.hip")
    mid = {k: v for k, v in res.items() if "nplda_fwd_mid_kernelILi10E" in k}
    assert len(mid) == 8, sorted(k for k in res if "mid_kernel" in k)
    for k, v in mid.items():
        assert v["ScratchSize"] == 0, (k, v)
