#!/usr/bin/env python3
"""Single-GPU shard timings: `bench.py --emulate-rank r/N` for N = 1, 2, 4, 8 over the workloads that shard:
cfg1 strong (one 1 M-pair list split N ways), cfg2 weak (4096 pairs per rank: data-parallel training is weak-scaled only), cfg3 (rows and trials split N ways, the cohort replicated).  One process, one GPU, rank r's exact
share; collectives are NOT run (their payload is listed).  The table is the compute side of a scaling curve — the implied
efficiency is (time at N = 1) / (N x time of the slowest emulated rank) for strong scaling — and is labelled as such:
it is not a scaling measurement.    usage: emulate_ranks.py [out.txt]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(argv):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, cwd=ROOT)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    if out.returncode != 0 or not lines:
        raise SystemExit(f"bench.py {' '.join(argv)} failed:\n{out.stderr[-1500:]}")
    return json.loads(lines[0])


def main():
    rows = []
    jobs = [("cfg1 strong (1 048 576 pairs / N)", ["--workload", "cfg1", "--steps", "30", "--warmup", "5", "--no-clock-probe"], "pairs"),
            # (no strong scaling of the 4096-pair training step: 512 pairs per rank cost what 4096 cost — the step is four launch
            #  latencies — so data-parallel training is weak-scaled only: DESIGN.md section 6)
            ("cfg2 weak (4096 pairs per rank)", ["--workload", "cfg2", "--scaling", "weak", "--steps", "300", "--warmup", "30"], "pairs"),
            ("cfg3 (22 000 rows, 2 M trials / N; cohort 10 000 replicated)", ["--workload", "cfg3", "--steps", "20", "--warmup", "3"], "trials"),
            ("cfg3 with a prepared cohort (CohortState: the cohort embedded and pre-passed once, outside the step)",
             ["--workload", "cfg3", "--steps", "20", "--warmup", "3", "--prepared-cohort"], "trials"),
            ("cfg5 weak (head step with dL/dx, 4096 bf16 pairs per rank)", ["--workload", "cfg5", "--steps", "300", "--warmup", "30"], "pairs")]
    for title, argv, unit in jobs:
        base = None
        for n in (1, 2, 4, 8):
            ranks = [0] if n == 1 else sorted({0, n - 1})  # first and last shard (the last may be short)
            worst = None
            for r in ranks:
                d = run(argv + ["--emulate-rank", f"{r}/{n}"])
                if worst is None or d["ms_per_step"] > worst["ms_per_step"]:
                    worst = d
            ms = worst["ms_per_step"]
            if n == 1:
                base = ms
            c = worst["config"]
            extra = ""
            if "stats_ms" in c:
                extra = f"stats {c['stats_ms']:.3f} ms, apply {c['apply_ms']:.3f} ms, all-gather payload {c['allgather_bytes']} B"
            elif c.get("collective_bytes_per_step"):
                extra = "all-reduce payload " + " + ".join(f"{v} B" for v in c["collective_bytes_per_step"].values())
            per = c.get("pairs_per_gpu_per_step", c.get("trials_per_gpu"))
            weak = "weak" in title
            eff = base / ms if weak else base / (n * ms)
            rows.append((title, n, per, ms, worst["roofline"]["frac"], eff, extra))
            print(rows[-1], flush=True)
    lines = ["# bench.py --emulate-rank r/N on ONE MI355X: rank r's share of an N-rank job, collectives not run (payload listed).",
             "# SINGLE-GPU SHARD TIMING — the compute side only; NOT a scaling measurement (no 8-GPU node was available).",
             "# 'implied' = compute-side efficiency: t(N=1) / (N x t) for strong scaling, t(N=1) / t for weak scaling; slowest of the",
             "# first and last rank's shard.", ""]
    cur = None
    for title, n, per, ms, frac, eff, extra in rows:
        if title != cur:
            lines += ["", title, f"  {'N':>2s} {'units/rank':>11s} {'ms/step':>10s} {'roofline frac':>14s} {'implied':>8s}  notes"]
            cur = title
        lines.append(f"  {n:2d} {per:11d} {ms:10.4f} {frac:14.3f} {eff:8.3f}  {extra}")
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
