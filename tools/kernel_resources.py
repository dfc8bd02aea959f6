#!/usr/bin/env python3
"""Registers / LDS / occupancy of every kernel in neuralplda_amd/csrc (hipcc -Rpass-analysis=kernel-resource-usage),
to check that the occupancy a kernel was designed for is the one it gets.  usage: kernel_resources.py [substring]"""
import glob, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
want = sys.argv[1] if len(sys.argv) > 1 else ""
rows = []
for src in sorted(glob.glob(os.path.join(ROOT, "neuralplda_amd", "csrc", "*.hip"))):
    out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                          "-I" + os.path.dirname(src), "-c", src, "-o", "/tmp/_kr.o",
                          "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
    cur = {}
    for line in out.splitlines():
        m = re.search(r"remark: +(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|"
                      r"LDS Size \[bytes/block\]): +(\S+)", line)
        if not m:
            continue
        k, v = m.group(1).split(" ")[0], m.group(2)
        if k == "Function":
            cur = {"name": v, "file": os.path.basename(src)}
        else:
            cur[k] = v
        if k == "LDS":
            rows.append(cur)
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    if "rocprim" in name or want not in name:
        continue
    print(f'occ {r.get("Occupancy"):>2}  vgpr {r.get("VGPRs"):>3}  agpr {r.get("AGPRs"):>3}  scratch {r.get("ScratchSize"):>4}  '
          f'lds {r.get("LDS"):>6}  {r["file"]:22s} {name[:110]}')
