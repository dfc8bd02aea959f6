import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from neuralplda_amd import ops
which = sys.argv[1]
R = int(sys.argv[2]) if len(sys.argv) > 2 else 2750
dev = torch.device("cuda:0")
prm, _ = bench.make_params(150, dev)
pk = ops.pack_params(*prm)
g = torch.Generator(device=dev).manual_seed(1)
xr = torch.randn(R, 512, device=dev, generator=g); xc = torch.randn(10000, 512, device=dev, generator=g)
T = 250000
ie = torch.randint(0, R // 10, (T,), device=dev, generator=g); it = torch.randint(R // 10, R, (T,), device=dev, generator=g)
raw = torch.randn(T, device=dev, generator=g, dtype=torch.float64)
zc, qc = ops.embed(xc, pk)
prep = ops.cohort_prepare(zc, qc, pk, topn=500)
zr0, qr0 = ops.embed(xr, pk)
st0 = ops.cohort_stats(zr0, qr0, zc, qc, pk, topn=500, prepared=prep)
def step():
    if which == "embed": return ops.embed(xr, pk)[0]
    if which == "stats": return ops.cohort_stats(zr0, qr0, zc, qc, pk, topn=500, prepared=prep)
    if which == "stats_noprep": return ops.cohort_stats(zr0, qr0, zc, qc, pk, topn=500)
    if which == "apply": return ops.asnorm_apply(raw, ie, it, st0)
    if which == "embed_stats":
        zr, qr = ops.embed(xr, pk)
        return ops.cohort_stats(zr, qr, zc, qc, pk, topn=500, prepared=prep)
    if which == "stats_apply":
        return ops.asnorm_apply(raw, ie, it, ops.cohort_stats(zr0, qr0, zc, qc, pk, topn=500, prepared=prep))
    if which == "all":
        zr, qr = ops.embed(xr, pk)
        return ops.asnorm_apply(raw, ie, it, ops.cohort_stats(zr, qr, zc, qc, pk, topn=500, prepared=prep))
if os.environ.get('EAGER_FIRST'):
    for _ in range(1000): step()
    torch.cuda.synchronize()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    out = step()
NREP = int(os.environ.get('NREP', '300'))
for i in range(NREP):
    gr.replay()
torch.cuda.synchronize()
print(which, NREP, "replays ok", float(out.double().sum()))
