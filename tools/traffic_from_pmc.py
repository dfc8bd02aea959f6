#!/usr/bin/env python3
"""profiles/traffic.json from PMC passes, not by hand.

usage: traffic_from_pmc.py <traffic.json> <key>=<pmc dir>:<kernel substring>[+<kernel substring>...] ...

<pmc dir> is what tools/pmc_cmd.sh wrote (fetch.txt / write.txt: tools/rocpd_summary.py tables of separate rocprofv3 --pmc
FETCH_SIZE and --pmc WRITE_SIZE runs).  Per launch and kernel: HBM bytes = 2 x FETCH_SIZE KB x 1024 (gfx950 tallies a 128-byte
read request as 64 B — MI355X_MICROARCH.md, HBM section; calibrated in round 1 against TCC_MISS and an x-load-only kernel) +
WRITE_SIZE KB x 1024.  Several kernel substrings joined by '+' are the launches of ONE step (their bytes add up).  The key's
entry and a `_src_<key>` note naming the files and the raw counter means are written; other entries are kept."""
import json
import os
import re
import sys


def counter_means(path, counter):
    """{kernel name: mean} from the counter table of a rocpd_summary file."""
    out = {}
    for ln in open(path):
        m = re.match(r"\s+(\S.*?)\s+" + counter + r"\s+(\d+)\s+([0-9.]+)\s", ln)
        if m:
            out[m.group(1)] = float(m.group(3))
    return out


def main():
    path = sys.argv[1]
    data = json.load(open(path)) if os.path.exists(path) else {}
    for spec in sys.argv[2:]:
        key, rest = spec.split("=", 1)
        pdir, kernels = rest.rsplit(":", 1)
        fetch = counter_means(os.path.join(pdir, "fetch.txt"), "FETCH_SIZE")
        write = counter_means(os.path.join(pdir, "write.txt"), "WRITE_SIZE")
        total, notes = 0.0, []
        for sub in kernels.split("+"):
            f = [v for k, v in fetch.items() if sub in k]
            w = [v for k, v in write.items() if sub in k]
            if len(f) != 1 or len(w) != 1:
                raise SystemExit(f"{key}: kernel substring {sub!r} matches {len(f)} fetch / {len(w)} write rows in {pdir}")
            total += 2.0 * f[0] * 1024.0 + w[0] * 1024.0
            notes.append(f"{sub}: 2 x FETCH_SIZE {f[0]:.0f} KB + WRITE_SIZE {w[0]:.0f} KB")
        data[key] = int(round(total))
        data[f"_src_{key}"] = f"{pdir}/fetch.txt, write.txt (tools/traffic_from_pmc.py): " + "; ".join(notes)
    with open(path, "w") as fh:
        json.dump(data, fh, indent=1)
        fh.write("\n")


if __name__ == "__main__":
    main()
