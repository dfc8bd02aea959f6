"""Throw-away style micro timing of the fused forward (used while iterating on the kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neuralplda_amd import ops

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    for D in (150, 170):
        g = torch.Generator(device="cuda").manual_seed(1)
        W1 = (torch.rand(D, 512, device="cuda", generator=g) - 0.5) * 0.08
        b1 = torch.rand(D, device="cuda", generator=g) - 0.5
        W2 = (torch.rand(D, D, device="cuda", generator=g) - 0.5) * 0.15
        b2 = torch.rand(D, device="cuda", generator=g) - 0.5
        Ps = torch.rand(D, device="cuda", generator=g)
        Q = torch.rand(D, device="cuda", generator=g)
        packed = ops.pack_params(W1, b1, W2, b2, Ps, Q)
        x1 = torch.randn(B, 512, device="cuda", generator=g)
        x2 = torch.randn(B, 512, device="cuda", generator=g)
        for _ in range(3):
            s = ops.score_pairs(x1, x2, packed)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            s = ops.score_pairs(x1, x2, packed)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        nb = 10 if D == 150 else 11
        flop_pad = 2 * 2 * (512 + 16 * nb) * 16 * nb
        flop_alg = 2 * (2 * 512 * D + 2 * D * D) + 8 * D
        print(f"D={D} B={B}: {ms:.3f} ms  {B/ms*1e3:.3e} pairs/s  alg {B*flop_alg/ms/1e9:.1f} TF  padded {B*flop_pad/ms/1e9:.1f} TF  HBM {B*4100/ms/1e9:.2f} TB/s")
        # torch reference on device for sanity
        u1 = torch.nn.functional.normalize(x1[:4096] @ W1.T + b1); z1 = u1 @ W2.T + b2
        u2 = torch.nn.functional.normalize(x2[:4096] @ W1.T + b1); z2 = u2 @ W2.T + b2
        ref = (z1*Q*z1).sum(1) + (z2*Q*z2).sum(1) + 2*(z1*Ps*Ps*z2).sum(1)
        print("   max |s - torch_dev_ref| =", (s[:4096]-ref).abs().max().item())

if __name__ == "__main__":
    main()
