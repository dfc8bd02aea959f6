#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (.db) outputs: per-kernel time stats and per-kernel PMC means.

usage: rocpd_summary.py [--by-grid] <results.db> [...]   (prints a text table; redirect into profiles/)
--by-grid: kernel-trace rows grouped by (kernel, grid size) — one kernel launched at several problem sizes.
"""
import sqlite3
import sys


def short(name, n=90):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name if len(name) <= n else name[: n - 3] + "..."


def main():
    by_grid = "--by-grid" in sys.argv
    for path in [a for a in sys.argv[1:] if a != "--by-grid"]:
        c = sqlite3.connect(path)
        print(f"== {path}")
        if by_grid:
            q = ("select name, grid_x, workgroup_x, count(*), avg(duration), min(duration) from kernels "
                 "group by name, grid_x order by name, grid_x")
            print(f"  {'kernel':70s} {'grid':>9s} {'wg':>5s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s}")
            for name, gx, wx, n, avg, mn in c.execute(q):
                print(f"  {short(name, 70):70s} {gx:9d} {wx:5d} {n:6d} {avg / 1e3:10.2f} {mn / 1e3:10.2f}")
            continue
        try:
            rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
        except sqlite3.Error as e:
            rows = []
            print("  (no top_kernels view:", e, ")")
        if rows:
            print(f"  {'kernel':90s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>7s}")
            for name, calls, tot, avg, pct in rows[:15]:
                print(f"  {short(name):90s} {calls:6d} {tot:12.1f} {avg:10.2f} {pct:7.2f}")
        try:
            q = ("select kernel_name, counter_name, count(*), avg(value), min(value), max(value), avg(duration), "
                 "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_block_size), max(grid_size), "
                 "max(workgroup_size) from counters_collection group by kernel_name, counter_name order by avg(duration) desc")
            rows = list(c.execute(q))
        except sqlite3.Error:
            rows = []
        if rows:
            print(f"  {'kernel':60s} {'counter':24s} {'n':>4s} {'mean':>16s} {'min':>16s} {'max':>16s} {'avg_ns':>10s}  vgpr/agpr/sgpr/lds grid/wg")
            for r in rows[:40]:
                print(f"  {short(r[0], 60):60s} {r[1]:24s} {r[2]:4d} {r[3]:16.2f} {r[4]:16.2f} {r[5]:16.2f} {r[6]:10.0f}  "
                      f"{r[7]}/{r[8]}/{r[9]}/{r[10]} {r[11]}/{r[12]}")


if __name__ == "__main__":
    main()
