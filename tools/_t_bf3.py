import sys, os, torch
sys.path.insert(0, "/root/repo")
import bench
from neuralplda_amd import ops
for D in (150, 170):
    prm, _ = bench.make_params(D, torch.device("cuda:0"))
    pk3 = ops.pack_params(*prm, precision="bf16x3")
    pk = ops.pack_params(*prm)
    for B in (65536, 262144, 1048576):
        x1 = torch.randn(B, 512, device="cuda"); x2 = torch.randn(B, 512, device="cuda")
        s3 = ops.score_pairs(x1, x2, pk3); s = ops.score_pairs(x1, x2, pk)
        d = (s3 - s).abs().max().item()
        ms, _ = bench.kernel_ms_of(lambda: ops.score_pairs(x1, x2, pk3), reps=10)
        NB = 10 if D == 150 else 11
        mf = B * 2 * 6 * (16 * NB * 512 + 16 * NB * 32 * ((NB + 1) // 2)) * 2 / 2 
        print(f"D={D} B={B}: {ms*1e3:8.1f} us  {B/ms/1e3:.3e} pairs/s  maxdiff vs fp32 {d:.2e}  issued {mf/(ms*1e-3)/1e12:.0f} TFLOP/s", flush=True)
    z3, q3 = ops.embed(x1, pk3); z, q = ops.embed(x1, pk)
    print("embed diff", (z3 - z).abs().max().item(), (q3 - q).abs().max().item())
