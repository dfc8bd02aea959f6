import torch, numpy as np, sys
sys.path.insert(0, '/root/repo')
from neuralplda_amd import ops
torch.manual_seed(0)
D=150
W1=torch.randn(D,512,device='cuda')*0.05; b1=torch.randn(D,device='cuda')*0.1
W2=torch.randn(D,D,device='cuda')*0.1; b2=torch.randn(D,device='cuda')*0.1
P=torch.randn(D,device='cuda'); Q=torch.randn(D,device='cuda')
packed=ops.pack_params(W1,b1,W2,b2,P,Q)
for N in (16, 40, 200, 1000, 5000, 40000):
    x=torch.randn(N,512,device='cuda')
    z,q=ops.embed(x,packed)
    u=x.double()@W1.double().T+b1.double(); y=u/u.norm(dim=1,keepdim=True).clamp_min(1e-12); zr=y@W2.double().T+b2.double()
    d=(z[:, :D].double()-zr).abs()
    bad=(d>1e-4)
    print(N, 'max err', d.max().item(), 'bad rows', bad.any(1).sum().item(), 'bad cols', bad.any(0).sum().item(), 'pad max', z[:, D:].abs().max().item())
    if bad.any():
        r=bad.any(1).nonzero().flatten()[:20].tolist(); c=bad.any(0).nonzero().flatten()[:40].tolist()
        print('  rows', r); print('  cols', c)
