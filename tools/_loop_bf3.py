import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from neuralplda_amd import ops
prec = sys.argv[1]
prm, _ = bench.make_params(150, torch.device("cuda:0"))
pk = ops.pack_params(*prm, precision=prec)
B = 1048576
x1 = torch.randn(B, 512, device="cuda"); x2 = torch.randn(B, 512, device="cuda")
t0 = time.time()
while time.time() - t0 < 14:
    for _ in range(50):
        ops.score_pairs(x1, x2, pk)
    torch.cuda.synchronize()
