"""Is the D = 170 scoring rate data dependent (DVFS)?  Same kernel, four input sets, interleaved rounds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from neuralplda_amd import ops

B, D0 = 1 << 20, 512
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
cases = {}
for D in (150, 170):
    pk, _ = bench.make_params(D, dev)
    W1 = (torch.rand(D, 512, device=dev, generator=g) - 0.5) * 0.08
    b1 = torch.rand(D, device=dev, generator=g) - 0.5
    W2 = (torch.rand(D, D, device=dev, generator=g) - 0.5) * 0.15
    b2 = torch.rand(D, device=dev, generator=g) - 0.5
    Ps, Q = torch.rand(D, device=dev, generator=g), torch.rand(D, device=dev, generator=g)
    cases[f"D{D} kaldi"] = (D, ops.pack_params(*pk))
    cases[f"D{D} random"] = (D, ops.pack_params(W1, b1, W2, b2, Ps, Q))
xs = {"randn": (torch.randn(B, D0, device=dev, generator=g), torch.randn(B, D0, device=dev, generator=g)),
      "randn*8": (8 * torch.randn(B, D0, device=dev, generator=g), 8 * torch.randn(B, D0, device=dev, generator=g))}
for rnd in range(3):
    for cn, (D, pk) in cases.items():
        for xn, (x1, x2) in xs.items():
            ms, _ = bench.kernel_ms_of(lambda: ops.score_pairs(x1, x2, pk), reps=10)
            f = bench.algorithmic_flops_per_pair(D0, D, D)
            print(f"round {rnd}  {cn:12s} x={xn:8s} {ms:.3f} ms  frac {B * f / (ms * 1e-3) / 1e12 / 157.3:.3f}", flush=True)
