"""A/B of the host-side switches of the reference's literal loop body (xvector_NeuralPlda_pytorch.py:35-43) in ONE process,
the variants interleaved round by round so that they share the box's noise:

    base       compat.install(fused_adam=True, inline_backward=False)
    inline     ... inline_backward=True  (what fused_adam=True selects)
    deferred   ... + deferred_keyerror=True

Prints min / median ms per step of `literal` (with loss.item()) and `resident` (index batches on the device, no .item())."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    import neuralplda_amd.compat as compat
    from neuralplda_amd import ops
    compat.install(fused_adam=True, inline_backward=False)
    from utils.models import NeuralPlda
    from utils import sv_trials_loaders as svl
    dev = torch.device("cuda")
    D = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 12

    class NC:
        xvector_dim, layer1_LDA_dim, layer2_PLDA_spkfactor_dim = 512, D, D
        beta, alpha, device, loss = [99.0, 199.0], 15.0, "cuda", "SoftCdet"

    torch.manual_seed(0)
    model = NeuralPlda(NC()).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-5)
    rng = np.random.default_rng(0)
    n_utt = 200000
    ids = [f"utt{i:07d}" for i in range(n_utt)]
    mega = svl.XvectorTable.from_matrix(ids, rng.standard_normal((n_utt, 512), dtype=np.float32))
    num_to_id = dict(enumerate(ids))
    batches = [(torch.from_numpy(rng.integers(0, n_utt, B)), torch.from_numpy(rng.integers(0, n_utt, B)),
                torch.from_numpy((rng.random(B) < 0.1).astype(np.float32))) for _ in range(16)]
    dbatches = [(a.to(dev), b.to(dev), c.to(dev)) for a, b, c in batches]
    model.train()

    def literal(k):
        d1, d2, t = batches[k % 16]
        opt.zero_grad()
        d1, d2, t = d1.to(dev), d2.to(dev), t.to(dev)
        x1, x2 = svl.load_xvec_trials_from_numbatch(mega, num_to_id, d1, d2, dev)
        loss = model.loss(model(x1, x2), t)
        lv = loss.item()
        loss.backward()
        opt.step()
        return lv

    def resident(k):
        d1, d2, t = dbatches[k % 16]
        opt.zero_grad()
        x1, x2 = svl.load_xvec_trials_from_numbatch(mega, num_to_id, d1, d2, dev)
        loss = model.loss(model(x1, x2), t)
        loss.backward()
        opt.step()

    variants = {"base": (True, False), "inline": (False, False), "deferred": (False, True)}
    res = {v: {"literal": [], "resident": []} for v in variants}
    for r in range(rounds + 1):
        for v, (mt, deferred) in variants.items():
            torch.autograd.set_multithreading_enabled(mt)
            ops.KEYERROR_DEFERRED = deferred
            for name, fn, n in (("literal", literal, 60), ("resident", resident, 60)):
                for k in range(5):
                    fn(k)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for k in range(n):
                    fn(k)
                torch.cuda.synchronize()
                if r:  # (round 0 warms up)
                    res[v][name].append((time.perf_counter() - t0) / n * 1e3)
    out = {"D": D, "B": B, "rounds": rounds}
    for v in variants:
        for name in ("literal", "resident"):
            ts = res[v][name]
            out[f"{v}_{name}_ms"] = [round(min(ts), 4), round(float(np.median(ts)), 4), round(max(ts), 4)]
    print(json.dumps(out, indent=1))
    ops.KEYERROR_DEFERRED = False
    compat.uninstall()


if __name__ == "__main__":
    main()
