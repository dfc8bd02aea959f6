"""Random streaming-size batches at D = 150 (the shape nplda_fwd_v6_kernel serves: nine feature blocks on 16x16x4 MFMAs, six
features on 4x4x1 MFMAs) against the fp64 oracle: x-vector dims 16 .. 640 (any chunk count, odd ones included), B such that
the dispatch picks the streaming kernel, fp32 and bf16 rows, weights at several scales.  usage: fuzz_stream_d150.py [cases]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import nplda_oracle as orc
from neuralplda_amd import ops
rng = np.random.default_rng(150)
D = 150
worst = 0.0
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
for it in range(ncases):
    D0 = int(rng.choice([16, 32, 72, 144, 200, 256, 400, 512, 640]))
    if D0 == 512:   # 512-d: whole rounds of the persistent grid (otherwise the balanced-tile kernel takes the batch)
        B = int(rng.choice([32768, 32768 - int(rng.integers(1, 128)), 65536 - int(rng.integers(0, 128)), 131072 - int(rng.integers(0, 64))]))
    else:
        B = int(rng.integers(16385, 90000))
    scale = float(rng.choice([0.3, 1.0, 3.0]))
    k1, k2 = scale / np.sqrt(D0), scale / np.sqrt(D)
    p = orc.Params(rng.uniform(-k1, k1, (D, D0)).astype(np.float32), rng.uniform(-k1, k1, D).astype(np.float32),
                   rng.uniform(-k2, k2, (D, D)).astype(np.float32), rng.uniform(-k2, k2, D).astype(np.float32),
                   rng.uniform(0, 1, D).astype(np.float32), rng.uniform(-1, 1, D).astype(np.float32))
    pk = ops.pack_params(*[torch.from_numpy(a).cuda() for a in (p.W1, p.b1, p.W2, p.b2, p.P_sqrt, p.Q)])
    x1 = rng.standard_normal((B, D0)).astype(np.float32); x2 = rng.standard_normal((B, D0)).astype(np.float32)
    X1, X2 = torch.from_numpy(x1).cuda(), torch.from_numpy(x2).cuda()
    s = ops.score_pairs(X1, X2, pk).cpu().numpy()
    # the oracle on the first, the last (ragged tile) and 2000 random rows
    idx = np.unique(np.concatenate([np.arange(300), np.arange(B - 300, B), rng.integers(0, B, 2000)]))
    ref = orc.forward(x1[idx], x2[idx], p, np.float64)
    e1 = np.max(np.abs(s[idx] - ref) / (2e-5 + 1e-5 * np.abs(ref)))
    # bf16 rows: the same kernel on widened values — compare with the fp32 kernel on the widened rows
    Xb1, Xb2 = X1.bfloat16(), X2.bfloat16()
    sb = ops.score_pairs(Xb1, Xb2, pk)
    e2 = float((sb - ops.score_pairs(Xb1.float(), Xb2.float(), pk)).abs().max())
    worst = max(worst, e1)
    print(f"D0={D0} B={B} scale={scale}: score tol-units {e1:.3f}  bf16 rows vs fp32 kernel on the widened rows max|d| {e2:.1e}", flush=True)
    assert e1 <= 1 and e2 == 0.0, (D0, B)
print("ok, worst", worst)
