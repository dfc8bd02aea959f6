#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (.db) outputs: per-kernel time statistics and per-kernel PMC means.

usage: rocpd_summary.py [--by-grid] [--drop-first] <results.db> [...]   (prints a text table; redirect into profiles/)
--by-grid:    kernel-trace rows grouped by (kernel, grid size) — one kernel launched at several problem sizes.
--drop-first: leave each kernel's FIRST launch of the process out of its statistics (it carries code-object loading and a
              cold clock: 20 ms on a 3 ms kernel, which alone moves a 69-launch average by 8 %).
--series N:   after the table, the per-launch durations (launch order) of the first N launches of the kernel with the
              largest total time — the cold start is several launches long, not one (clock ramp from idle).
Every time row carries calls / total / avg AND min / median / max, so an outlier is visible next to the average it skews.
"""
import sqlite3
import statistics
import sys


def short(name, n=90):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name if len(name) <= n else name[: n - 3] + "..."


def kernel_rows(c, by_grid, drop_first):
    """[(name, grid, wg, [durations ns in launch order])] from the `kernels` view."""
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    order = "start" if "start" in cols else ("id" if "id" in cols else "rowid")
    groups = {}
    for name, gx, wx, dur in c.execute(f"select name, grid_x, workgroup_x, duration from kernels order by {order}"):
        key = (name, gx) if by_grid else (name,)
        g = groups.setdefault(key, {"name": name, "grid": gx, "wg": wx, "d": []})
        g["d"].append(dur)
    rows = []
    for g in groups.values():
        d = g["d"][1:] if drop_first and len(g["d"]) > 1 else g["d"]
        rows.append((g["name"], g["grid"], g["wg"], d, len(g["d"]) - len(d)))
    return rows


def main():
    flags = {"--by-grid", "--drop-first"}
    by_grid, drop_first = "--by-grid" in sys.argv, "--drop-first" in sys.argv
    argv = list(sys.argv[1:])
    series = 0
    if "--series" in argv:
        i = argv.index("--series")
        series = int(argv[i + 1])
        del argv[i:i + 2]
    for path in [a for a in argv if a not in flags]:
        c = sqlite3.connect(path)
        print(f"== {path}" + ("   (first launch of each kernel dropped)" if drop_first else ""))
        try:
            rows = kernel_rows(c, by_grid, drop_first)
        except sqlite3.Error as e:
            rows = []
            print("  (no kernels view:", e, ")")
        if rows:
            total_all = sum(sum(r[3]) for r in rows) or 1
            rows.sort(key=(lambda r: (r[0], r[1])) if by_grid else (lambda r: -sum(r[3])))
            w = 62 if by_grid else 74
            head = f"  {'kernel':{w}s}" + (f" {'grid':>9s} {'wg':>5s}" if by_grid else "")
            print(head + f" {'calls':>6s} {'total_us':>11s} {'avg_us':>9s} {'min_us':>9s} {'med_us':>9s} {'max_us':>9s} {'pct':>6s}")
            for name, gx, wx, d, _ in rows[: (400 if by_grid else 18)]:
                if not d:
                    continue
                line = f"  {short(name, w):{w}s}" + (f" {gx:9d} {wx:5d}" if by_grid else "")
                print(line + f" {len(d):6d} {sum(d) / 1e3:11.1f} {statistics.fmean(d) / 1e3:9.2f} {min(d) / 1e3:9.2f} "
                             f"{statistics.median(d) / 1e3:9.2f} {max(d) / 1e3:9.2f} {100.0 * sum(d) / total_all:6.2f}")
        if series and rows:
            # the dominant kernel's launches one by one, nothing dropped (kernel_rows(..., drop_first=False))
            allrows = kernel_rows(c, by_grid, False)
            name, gx, wx, d, _ = max(allrows, key=lambda r: sum(r[3]))
            print(f"  per-launch series of {short(name, 70)} (grid {gx}), first {min(series, len(d))} of {len(d)} launches, us:")
            print("    " + " ".join(f"{v / 1e3:.1f}" for v in d[:series]))
        try:
            q = ("select kernel_name, counter_name, count(*), avg(value), min(value), max(value), avg(duration), "
                 "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_block_size), max(grid_size), "
                 "max(workgroup_size) from counters_collection group by kernel_name, counter_name order by avg(duration) desc")
            rows = list(c.execute(q))
        except sqlite3.Error:
            rows = []
        if rows:
            print(f"  {'kernel':60s} {'counter':24s} {'n':>4s} {'mean':>16s} {'min':>16s} {'max':>16s} {'avg_ns':>10s}  vgpr/agpr/sgpr/lds grid/wg")
            for r in rows[:60]:
                print(f"  {short(r[0], 60):60s} {r[1]:24s} {r[2]:4d} {r[3]:16.2f} {r[4]:16.2f} {r[5]:16.2f} {r[6]:10.0f}  "
                      f"{r[7]}/{r[8]}/{r[9]}/{r[10]} {r[11]}/{r[12]}")


if __name__ == "__main__":
    main()
