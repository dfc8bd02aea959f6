#!/usr/bin/env python3
"""cohort_fused2_kernel under the ABLATE build (tools/exp_cohort_build.sh): time of the cfg3 statistics call for a list of
NPLDA_COHORT_ABL settings (each in a child process: the setting is read once) and the per-tile cycle stamps of waves 0 and
NW / 2 of block 8 (phase boundaries: 0 tile start, 1 DMA issued, 2 loop done, 3 epilogue done, 4 bookkeeping done,
5 vmcnt/lgkmcnt drained, 6 barrier passed).   usage: exp_cohort_stamps.py [D=150] [abl values ...]"""
import ctypes, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if os.environ.get("_CHILD") != "1":
    D = sys.argv[1] if len(sys.argv) > 1 else "150"
    for abl in (sys.argv[2:] or ["0"]):
        env = dict(os.environ, _CHILD="1", NPLDA_COHORT_ABL=abl)
        r = subprocess.run([sys.executable, __file__, D], env=env, capture_output=True, text=True)
        print(f"---- NPLDA_COHORT_ABL={abl} NW={os.environ.get('NPLDA_COHORT_NW', '8')}\n" + r.stdout + r.stderr[-600:])
    sys.exit(0)

import numpy as np, torch
from neuralplda_amd import _lib, models, ops
D = int(sys.argv[1])


class NC:
    xvector_dim, layer1_LDA_dim, layer2_PLDA_spkfactor_dim = 512, D, D
    beta, alpha, device, loss = [99.0], 15.0, "cuda", "SoftCdet"


torch.manual_seed(3)
m = models.NeuralPlda(NC()).cuda()
packed = ops.pack_params(*[p.detach() for p in m._params()])
R, M = 22000, 10000
zr, qr = ops.embed(torch.randn(R, 512, device="cuda"), packed)
zc, qc = ops.embed(torch.randn(M, 512, device="cuda"), packed)
for _ in range(3):
    ops.cohort_stats(zr, qr, zc, qc, packed, topn=500)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        ops.cohort_stats(zr, qr, zc, qc, packed, topn=500)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 4)
print(f"cohort_stats whole call: median {sorted(ts)[2]:.4f} ms  (min {min(ts):.4f})")
lib = _lib.load()
buf = (ctypes.c_ulonglong * (2 * 64 * 8))()
lib.nplda_cohort_debug_stamps.restype = ctypes.c_int
rc = lib.nplda_cohort_debug_stamps(buf)
st = np.array(buf[:], dtype=np.int64).reshape(2, 64, 8)
if rc == 0 and st[0, 5, 0] > 0:
    names = ["dma", "loop", "epi", "book", "drain", "barrier"]
    for w in (0, 1):
        tiles = [k for k in range(4, 60) if st[w, k, 6] > 0 and st[w, k + 1, 0] > 0]
        d = np.array([[st[w, k, i + 1] - st[w, k, i] for i in range(6)] for k in tiles], dtype=np.float64)
        per = np.array([st[w, k + 1, 0] - st[w, k, 0] for k in tiles], dtype=np.float64)
        print(f"wave {'0' if w == 0 else 'NW/2'}: tile period median {np.median(per):8.0f} cycles; phases (median): " +
              ", ".join(f"{n} {np.median(d[:, i]):7.0f}" for i, n in enumerate(names)))
    # tiles that end an item / a list band stand out by their period: the longest periods of the stamped block
    for w in (0,):
        per = sorted(((int(st[w, k + 1, 0] - st[w, k, 0]), k) for k in range(2, 62) if st[w, k + 1, 0] > 0 and st[w, k, 0] > 0), reverse=True)
        print("longest tile periods (cycles, tile): " + ", ".join(f"{p_}@{k}" for p_, k in per[:6]) + f"; median {per[len(per) // 2][0]}")
    if os.environ.get("STAMP_RAW"):
        base = st[0, 8, 0]
        for k in range(8, 20):
            print(f"tile {k:2d}  w0: " + " ".join(f"{st[0, k, i] - base:7d}" for i in range(7)) +
                  "   wH: " + " ".join(f"{st[1, k, i] - base:7d}" for i in range(7)))
    # offsets between the two waves' phase boundaries (wave NW/2 minus wave 0), same tile
    tiles = [k for k in range(4, 60) if st[0, k, 6] > 0 and st[1, k, 6] > 0]
    off = np.array([[st[1, k, i] - st[0, k, i] for i in range(7)] for k in tiles], dtype=np.float64)
    print("wave NW/2 minus wave 0 at stamps 0..6 (median): " + ", ".join(f"{np.median(off[:, i]):7.0f}" for i in range(7)))
