"""bench.py's output contract: ONE JSON line with the driver's keys, the tier's `roofline` and `cpu_baseline` objects and
consistent numbers — run as the driver runs it (plain python at N = 1, torch.distributed.run for the multi-rank form)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check(line, steps, warmup, with_cpu):
    d = json.loads(line)
    assert d["metric"] == "scored trial-pairs/sec (512-d xvec)" and d["unit"] == "pairs/s"
    assert d["n_gpus"] == 1 and d["steps"] == steps and d["warmup"] == warmup
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    B = d["config"]["pairs_per_gpu_per_step"]
    assert abs(d["value"] - B / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.3 < r["frac"] < 1.0
    assert abs(r["achieved"] - B * r["flop_per_pair_algorithmic"] / (r["kernel_ms"] * 1e-3) / 1e12) < 1e-6 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] > 0
    assert 1000.0 < r["sclk_mhz_under_kernel"] < 2600.0 and r["frac_at_measured_clock"] >= r["frac"] * 0.9
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.02  # the kernel fits inside the step the wall clock saw
    if with_cpu:
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and c["unit"] == "pairs/s" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    return d


def test_bench_single_process_contract(hip_lib):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1",
                          "--cpu-seconds", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # exactly one JSON line
    d = _check(lines[0], 4, 1, True)
    assert "alt_bf16x3" in d and d["alt_bf16x3"]["max_abs_diff_vs_fp32_scores"] < 2e-5


def test_bench_torchrun_one_rank(hip_lib):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"),
                          "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-alt"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    _check(lines[0], 3, 1, False)
