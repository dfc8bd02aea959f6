"""ops.score_pairs at 256 .. 4608 pairs: us per call (best of 3 x 50).  NPLDA_FWD_SMALL_MAX=0 sends every size to the balanced-tile kernel."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralplda_amd import ops
rng = np.random.default_rng(0)
D = int(sys.argv[1]) if len(sys.argv) > 1 else 150
p = [rng.standard_normal(sh).astype(np.float32) * sc for sh, sc in (((D, 512), 0.05), ((D,), 0.1), ((D, D), 0.08), ((D,), 0.1), ((D,), 0.5), ((D,), 0.5))]
pk = ops.pack_params(*[torch.from_numpy(a).cuda() for a in p])
X1 = torch.randn(5000, 512, device="cuda"); X2 = torch.randn(5000, 512, device="cuda")
out = []
for B in (256, 512, 1024, 1536, 2048, 2056, 2560, 3072, 3584, 4096, 4608):
    x1, x2 = X1[:B], X2[:B]
    for _ in range(10): ops.score_pairs(x1, x2, pk)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for rep in range(3):
        e0.record()
        for _ in range(50): ops.score_pairs(x1, x2, pk)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 50 * 1000)
    out.append(f"{B}: {best:.1f}")
print(f"D={D}  " + "  ".join(out), flush=True)
