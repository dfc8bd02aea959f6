"""bench.py's output contract: ONE JSON line with the driver's keys, the tier's `roofline` and `cpu_baseline` objects and
consistent numbers — run as the driver runs it (plain python at N = 1, torch.distributed.run for the multi-rank form)."""
import json

import numpy as np
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
    assert abs(d["value"] - B / (d["ms_per_step"] * 1e-3)) <= 1e-5 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 157.3
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6 and 0.3 < r["frac"] < 1.0
    assert abs(r["achieved"] - B * r["flop_per_pair_algorithmic"] / (r["kernel_ms"] * 1e-3) / 1e12) < 1e-5 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] > 0
    assert 1000.0 < r["sclk_mhz_under_kernel"] < 2600.0 and r["frac_at_measured_clock"] >= r["frac"] * 0.9
    assert r["kernel_ms"] <= d["ms_per_step"] * 1.02  # the kernel fits inside the step the wall clock saw
    if with_cpu:
        c = d["cpu_baseline"]
        assert c["kind"] == "port" and c["unit"] == "pairs/s" and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
        # torch CPU ops at all physical cores and one core, D = 150 and 170, gather-inclusive figure
        for k in ("d150_all_cores", "d150_one_core", "d150_best_width", "d170_all_cores", "d170_one_core",
                  "d170_best_width", "gather_inclusive"):
            assert c["detail"][k]["pairs_per_s"] > 0 and c["detail"][k]["reps"] >= 3, k
        assert c["detail"]["d150_all_cores"]["threads"] == c["physical_cores"] and c["detail"]["d150_one_core"]["threads"] == 1
        assert c["value"] == c["detail"]["d150_best_width"]["pairs_per_s"] >= c["detail"]["d150_all_cores"]["pairs_per_s"]
        assert c["cores"] == c["detail"]["d150_best_width"]["threads"] and c["cpu_model"]
    return d


def test_bench_single_process_contract(hip_lib):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1",
                          "--cpu-seconds", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # exactly one JSON line
    d = _check(lines[0], 4, 1, True)
    assert "alt_bf16x3" in d and d["alt_bf16x3"]["max_abs_diff_vs_fp32_scores"] < 2e-5
    a = d["alt_d170"]  # the reference's shipped shape rides on the same line
    assert a["bound"] == "mfma" and abs(a["frac"] - a["achieved"] / a["peak"]) < 1e-6 and 0.3 < a["frac"] < 1.0
    assert a["flop_per_pair_algorithmic"] == 465120
    assert d["config"]["ranks_in_group"] == 1
    # the kernel label comes from the dispatch actually taken, and the line says which build of the library ran
    assert d["roofline"]["kernel"].startswith("nplda_fwd_v6_kernel") and a["kernel"].startswith("nplda_fwd_v5_kernel")
    assert d["lib"]["abi_version"] >= 2 and len(d["lib"]["csrc_sha"]) == 16 and d["lib"]["stale"] is False
    # the other BASELINE configs ride on the default line: training (cfg2), AS-norm (cfg3), the head of the E2E fine-tune (cfg5)
    for k, unit in (("alt_cfg2", "pairs/s"), ("alt_cfg3", "trials/s"), ("alt_cfg5", "pairs/s")):
        o = d[k]
        assert "error" not in o, o
        assert o["unit"] == unit and o["value"] > 0 and o["ms_per_step"] > 0 and o["workload"].startswith(k[4:])
        assert o["roofline"]["bound"] == "mfma" and 0 < o["roofline"]["frac"] < 1
        assert o["d170"]["ms_per_step"] > 0 and 0 < o["d170"]["frac"] < 1  # the reference's shipped shape beside it
    assert d["alt_cfg2"]["ms_per_step"] < 0.2 and d["alt_cfg3"]["stats_ms"] <= 1.0 and d["alt_cfg5"]["ms_per_step"] < 0.3
    # the line fits the ~8 KB of stdout tail the driver keeps, and ends with the objects that are read first; the prose
    # and the full-precision figures are in the detail file it names
    assert len(lines[0]) <= 7000, len(lines[0])
    assert list(d)[-6:] == ["cpu_baseline", "alt_d170", "alt_cfg5", "alt_cfg3", "alt_cfg2", "roofline"]
    full = json.load(open(os.path.join(ROOT, d["detail"])))
    assert full["alt_cfg3"]["workload"].startswith("cfg3: cohort 10000 x rows 22000") and "sample" in full["cpu_baseline"]
    assert abs(full["value"] - d["value"]) <= 1e-6 * d["value"] and full["alt_d170"]["checksum_finite"]
    for k in ("alt_dropin", "alt_regimeB", "alt_gb", "alt_dplda", "alt_minc"):
        assert k in d and "error" not in d[k] and k in full, k


from tests.test_bench_line_cpu import test_line_fits_the_driver_tail  # noqa: E402,F401  (also run on the GPU box)


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


def _run(argv, env=None, timeout=900):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True,
                         timeout=timeout, cwd=ROOT, env=dict(os.environ, **(env or {})))
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    return out, lines


def test_bench_gpus_flag_launches_the_ranks_itself(hip_lib):
    """`python bench.py --gpus N` (no torchrun) must start N ranks or fail: with RCCL and one visible GPU, N = 2 is an
    error exit, never a silent one-rank number; under the gloo dry-run backend the two ranks really form a group."""
    import torch
    if torch.cuda.device_count() < 2:
        out, lines = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"])
        assert out.returncode != 0 and not lines
    for scaling in ("weak", "strong"):
        out, lines = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--pairs", "65536", "--scaling", scaling,
                           "--no-cpu-baseline", "--no-alt", "--no-clock-probe"], env={"NPLDA_BENCH_BACKEND": "gloo"})
        assert out.returncode == 0, out.stderr[-2000:]
        assert len(lines) == 1
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 and d["config"]["ranks_in_group"] == 2 and d["scaling"] == scaling
        assert d["config"]["parallelism"] == "trial-list shard x2" and d["config"]["backend"] == "gloo"
        per_gpu = d["config"]["pairs_per_gpu_per_step"]
        assert per_gpu == (65536 if scaling == "weak" else 32768)
        total = 2 * per_gpu if scaling == "weak" else 65536
        assert abs(d["value"] - total / (d["ms_per_step"] * 1e-3)) <= 1e-5 * d["value"]


def test_bench_cfg3_single_and_two_rank_dry_run(hip_lib):
    """--workload cfg3: embed -> row-sharded cohort statistics -> ONE all-gather of (R, 4) -> trial-sharded apply."""
    small = ["--workload", "cfg3", "--steps", "2", "--warmup", "1", "--cohort", "2000", "--enroll", "300", "--test",
             "1700", "--trials", "100000"]
    out, lines = _run(small)
    assert out.returncode == 0, out.stderr[-2000:]
    d1 = json.loads(lines[0])
    assert d1["n_gpus"] == 1 and d1["unit"] == "trials/s" and d1["config"]["rows_per_gpu"] == 2000
    assert d1["roofline"]["bound"] == "mfma" and d1["roofline"]["kernel_ms"] > 0
    assert abs(d1["value"] - 100000 / (d1["ms_per_step"] * 1e-3)) <= 1e-5 * d1["value"]
    assert d1["config"]["stats_ms"] + d1["config"]["allgather_ms"] + d1["config"]["apply_ms"] <= d1["ms_per_step"] * 1.5
    out, lines = _run(["--gpus", "2"] + small, env={"NPLDA_BENCH_BACKEND": "gloo"})
    assert out.returncode == 0, out.stderr[-2000:]
    d2 = json.loads(lines[0])
    assert d2["n_gpus"] == 2 and d2["config"]["ranks_in_group"] == 2 and d2["config"]["rows_per_gpu"] == 1000
    assert d2["config"]["trials_per_gpu"] == 50000 and d2["config"]["allgather_bytes"] == 2 * 1000 * 32
    assert d2["config"]["parallelism"] == "row shard + trial shard x2"


def test_bench_cfg3_full_size_line(hip_lib):
    """BASELINE configs[3] at full size on one GPU: the line the driver would get from `--workload cfg3`."""
    out, lines = _run(["--workload", "cfg3", "--steps", "10", "--warmup", "3"])  # (3 steps after 1 warm-up once read 6 ms: one slow step)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["config"]["cohort"] == 10000 and d["config"]["rows"] == 22000 and d["config"]["trials"] == 2000000
    # the statistics call (cohort score tiles + per-row mean / std / top-N, nothing spilled) within its round-2 figure + 30 %
    assert d["config"]["stats_ms"] <= 1.0 and d["ms_per_step"] < 3.0 and d["config"]["cohort_scores_per_s"] > 1e11


def test_bench_cfg2_single_and_two_rank_dry_run(hip_lib):
    """--workload cfg2: 4096-pair training minibatches from a resident table; two gloo ranks shard the batch (eager: the dry
    run's collectives cannot be captured)."""
    small = ["--workload", "cfg2", "--steps", "5", "--warmup", "2", "--table", "50000"]
    out, lines = _run(small)
    assert out.returncode == 0, out.stderr[-2000:]
    d1 = json.loads(lines[0])
    assert d1["n_gpus"] == 1 and d1["unit"] == "pairs/s" and d1["config"]["global_batch"] == 4096
    assert d1["config"]["graph_replay"] is True and d1["roofline"]["bound"] == "mfma"
    assert abs(d1["value"] - 4096 / (d1["ms_per_step"] * 1e-3)) <= 1e-5 * d1["value"]
    assert d1["ms_per_step"] < 0.5 and np.isfinite(d1["config"]["final_loss"])
    out, lines = _run(["--gpus", "2", "--scaling", "strong"] + small, env={"NPLDA_BENCH_BACKEND": "gloo"})
    assert out.returncode == 0, out.stderr[-2000:]
    d2 = json.loads(lines[0])
    assert d2["n_gpus"] == 2 and d2["config"]["ranks_in_group"] == 2 and d2["config"]["pairs_per_gpu_per_step"] == 2048
    assert d2["config"]["global_batch"] == 4096 and d2["config"]["parallelism"] == "data parallel x2"


def test_bench_emulate_rank_lines(hip_lib):
    """--emulate-rank r/N: one process runs rank r's share of an N-rank job (single-GPU shard timing, labelled as such)."""
    out, lines = _run(["--emulate-rank", "3/8", "--steps", "3", "--warmup", "1", "--no-clock-probe"])
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["emulated_rank"] == "3/8" and "NOT a scaling" in d["config"]["note"]
    assert d["scaling"] == "strong" and d["config"]["pairs_per_gpu_per_step"] == (1 << 20) // 8 and "cpu_baseline" not in d
    assert abs(d["value"] - 131072 / (d["ms_per_step"] * 1e-3)) <= 1e-5 * d["value"]
    out, lines = _run(["--workload", "cfg3", "--emulate-rank", "7/8", "--steps", "2", "--warmup", "1"])
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["config"]["rows_per_gpu"] == 2750 and d["config"]["trials_per_gpu"] == 250000 and d["config"]["emulated_rank"] == "7/8"
    assert d["config"]["allgather_bytes"] == 8 * 2750 * 32
    out, lines = _run(["--workload", "cfg2", "--emulate-rank", "0/8", "--scaling", "strong", "--steps", "20", "--warmup", "5",
                       "--table", "50000"])
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["config"]["pairs_per_gpu_per_step"] == 512 and d["config"]["global_batch"] == 4096
    cb = d["config"]["collective_bytes_per_step"]  # ONE all-reduce per step: flat gradient + 2 x 18 floats of loss sums
    assert list(cb) == ["one_allreduce_flat_gradient_and_loss_sums"] and 390000 < cb["one_allreduce_flat_gradient_and_loss_sums"] < 410000
    out, lines = _run(["--workload", "cfg5", "--emulate-rank", "3/8", "--steps", "20", "--warmup", "5"])
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(lines[0])  # the head step under an extractor's data parallelism: ONE all-reduce, dL/dx local
    assert d["config"]["emulated_rank"] == "3/8" and d["config"]["global_batch"] == 8 * d["config"]["pairs_per_gpu_per_step"]
    assert list(d["config"]["collective_bytes_per_step"]) == ["one_allreduce_flat_gradient_and_loss_sums"]
    assert "nplda_train_step_grad_dx_f32" in d["config"]["step"] and 0 < d["ms_per_step"] < 1.0
    assert _run(["--emulate-rank", "8/8"])[0].returncode != 0 and _run(["--emulate-rank", "1/2", "--gpus", "2"])[0].returncode != 0


def test_bench_cfg5_line(hip_lib):
    """--workload cfg5: the head's step of the end-to-end fine-tune (bf16 x-vectors in, dL/dx out)."""
    out, lines = _run(["--workload", "cfg5", "--steps", "20", "--warmup", "5"])
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(lines[0])
    assert d["unit"] == "pairs/s" and d["config"]["global_batch"] == 4096 and d["dtype"].startswith("bf16")
    assert abs(d["value"] - 4096 / (d["ms_per_step"] * 1e-3)) <= 1e-5 * d["value"] and d["ms_per_step"] < 0.3
    assert d["roofline"]["flop_per_pair_algorithmic"] == 2 * 398400 + 2 * 2 * 150 * 150 + 2 * 2 * 512 * 150


def test_driver_scale_command_carries_the_collective_workloads(hip_lib):
    """The driver's SCALE command, unchanged (`python -m torch.distributed.run ... bench.py --gpus N --steps K --warmup W`),
    with two gloo ranks sharing this box's GPU: the ONE line must carry, beside the weak cfg1 value (which has no data-path
    collective), the strong cfg1 figure, the row-sharded AS-norm step with its all-gather, the one-collective data-parallel
    training step and the head step — each with ranks_in_group, the collective's bytes and max-over-ranks times.  (With RCCL
    the training steps are also timed with the all-reduce captured inside the HIP graph; gloo cannot be captured, so this dry
    run reports the eager form — exactly what the RCCL run falls back to if the capture fails.)"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NPLDA_BENCH_BACKEND="gloo")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "3", "--warmup", "1"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["ranks_in_group"] == 2 and "alts_aborted" not in d
    assert d["config"]["pairs_per_gpu_per_step"] == 1 << 20 and "cpu_baseline" not in d
    s = d["alt_cfg1_strong"]
    assert s["scaling"] == "strong" and s["n_gpus"] == 2 and s["pairs_per_gpu_per_step"] == 1 << 19
    assert abs(s["value"] - (1 << 20) / (s["ms_per_step"] * 1e-3)) <= 1e-5 * s["value"]
    a3 = d["alt_cfg3"]
    assert a3["ranks_in_group"] == 2 and a3["rows_per_gpu"] == 11000 and a3["trials_per_gpu"] == 1000000
    assert a3["allgather_bytes"] == 2 * 11000 * 32 and "all_gather_into_tensor" in a3["collective"]
    assert a3["phase_times"] == "max over ranks" and a3["stats_ms"] > 0 and a3["allgather_ms"] > 0 and a3["apply_ms"] > 0
    a2 = d["alt_cfg2"]
    assert a2["ranks_in_group"] == 2 and a2["global_batch"] == 8192 and a2["pairs_per_gpu_per_step"] == 4096
    assert a2["scaling"] == "weak" and a2["mode"] == "eager collectives" and a2["graph_replay"] is False
    assert list(a2["collective_bytes_per_step"]) == ["one_allreduce_flat_gradient_and_loss_sums"]
    assert abs(a2["value"] - 8192 / (a2["ms_per_step"] * 1e-3)) <= 1e-5 * a2["value"]
    a5 = d["alt_cfg5"]
    assert a5["ranks_in_group"] == 2 and a5["global_batch"] == 8192 and "eager collectives" in a5["mode"]
    assert list(a5["collective_bytes_per_step"]) == ["one_allreduce_flat_gradient_and_loss_sums"]


def test_scale_command_watchdog_keeps_the_headline(hip_lib):
    """A collective of the alt workloads that never returns must not cost the driver its line: with a 0.2-second budget the
    watchdog fires inside the alts, rank 0 prints the line assembled so far (the weak cfg1 headline + `alts_aborted`) and
    every rank leaves with status 0."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NPLDA_BENCH_BACKEND="gloo", NPLDA_BENCH_ALT_SECONDS="0.2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29543", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "3", "--warmup", "1"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and "watchdog" in d["alts_aborted"] and d["config"]["ranks_in_group"] == 2
