import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from neuralplda_amd import ops
rng = np.random.default_rng(0)
D = int(sys.argv[1]) if len(sys.argv) > 1 else 150
p = [rng.standard_normal(sh).astype(np.float32) * sc for sh, sc in (((D, 512), 0.05), ((D,), 0.1), ((D, D), 0.08), ((D,), 0.1), ((D,), 0.5), ((D,), 0.5))]
pk = ops.pack_params(*[torch.from_numpy(a).cuda() for a in p])
X1 = torch.randn(70000, 512, device="cuda"); X2 = torch.randn(70000, 512, device="cuda")
for B in (4500, 6000, 8192, 10240, 12288, 14336, 16384, 16385, 20480, 24577, 28672, 40000, 49152):
    x1, x2 = X1[:B], X2[:B]
    for _ in range(5): ops.score_pairs(x1, x2, pk)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for rep in range(3):
        e0.record()
        for _ in range(30): ops.score_pairs(x1, x2, pk)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 30 * 1000)
    print(f"D={D} B={B:6d}: {best:7.1f} us", flush=True)
