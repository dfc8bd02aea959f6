"""Random shapes (D = 145 .. 176, B = 1 .. 4096) through the small-batch scoring / training-forward / backward kernels against
the fp64 oracle: scores within the forward tolerance, flat gradient to 2e-4 of its max.  usage: fuzz_small_batch.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import nplda_oracle as orc
from neuralplda_amd import ops
rng = np.random.default_rng(2024)
worst = 0.0
for it in range(60):
    D = int(rng.integers(145, 177)); B = int(rng.choice([1, 2, 15, 16, 17, 31, 33, 100, 1003, 2048, 4095, 4096, int(rng.integers(1, 4097))]))
    k1, k2 = 1 / np.sqrt(512), 1 / np.sqrt(D)
    p = orc.Params(rng.uniform(-k1, k1, (D, 512)).astype(np.float32), rng.uniform(-k1, k1, D).astype(np.float32),
                   rng.uniform(-k2, k2, (D, D)).astype(np.float32), rng.uniform(-k2, k2, D).astype(np.float32),
                   rng.uniform(0, 1, D).astype(np.float32), rng.uniform(0, 1, D).astype(np.float32))
    pk = ops.pack_params(*[torch.from_numpy(a).cuda() for a in (p.W1, p.b1, p.W2, p.b2, p.P_sqrt, p.Q)])
    x1 = rng.standard_normal((B, 512)).astype(np.float32); x2 = rng.standard_normal((B, 512)).astype(np.float32)
    X1, X2 = torch.from_numpy(x1).cuda(), torch.from_numpy(x2).cuda()
    s = ops.score_pairs(X1, X2, pk).cpu().numpy()
    st, saved = ops.forward_train(X1, X2, pk)
    ref = orc.forward(x1, x2, p, np.float64)
    e1 = np.max(np.abs(s - ref) / (2e-5 + 1e-5 * np.abs(ref))); e2 = np.max(np.abs(st.cpu().numpy() - ref) / (2e-5 + 1e-5 * np.abs(ref)))
    g = rng.standard_normal(B).astype(np.float32)
    flat = ops.backward(saved, torch.from_numpy(g).cuda(), pk, torch.from_numpy(p.P_sqrt).cuda()).cpu().numpy()
    gr = orc.backward(x1, x2, g.astype(np.float64), p)
    refflat = np.concatenate([gr[k].ravel() for k in ("W1", "b1", "W2", "b2", "P_sqrt", "Q")])
    e3 = np.abs(flat - refflat).max() / (np.abs(refflat).max() + 1e-30)
    worst = max(worst, e1, e2)
    print(f"D={D} B={B}: score tol-units {e1:.3f} train-fwd {e2:.3f} grad rel {e3:.2e}", flush=True)
    assert e1 <= 1 and e2 <= 1 and e3 <= 2e-4, (D, B)
print("ok, worst", worst)
