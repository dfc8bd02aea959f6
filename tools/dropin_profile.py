"""Where the time of the reference's literal training-loop body goes (xvector_NeuralPlda_pytorch.py:35-43) when the
unchanged loop runs on this build's modules under compat.install():

    optimizer.zero_grad(); data.to(device); load_xvec_trials_from_numbatch(...); output = model(x1, x2);
    loss = model.loss(output, target); loss.item(); loss.backward(); optimizer.step()

Prints wall time per step (median of batches of 50 steps) and the host-side time of each phase (perf_counter between the
phases: launch / Python cost, with the device free-running except at the syncs the loop itself makes)."""
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
    compat.install(fused_adam="--fused-adam" in sys.argv, inline_backward="--inline-backward" in sys.argv,
                   deferred_keyerror="--deferred" in sys.argv)
    from utils.models import NeuralPlda
    from utils.sv_trials_loaders import load_xvec_trials_from_numbatch
    dev = torch.device("cuda")
    pos = [v for v in sys.argv[1:] if not v.startswith("--")]
    D = int(pos[0]) if len(pos) > 0 else 150
    B = int(pos[1]) if len(pos) > 1 else 4096

    class NC:
        xvector_dim, layer1_LDA_dim, layer2_PLDA_spkfactor_dim = 512, D, D
        beta, alpha, device, loss = [99.0, 199.0], 15.0, "cuda", "SoftCdet"

    torch.manual_seed(0)
    model = NeuralPlda(NC()).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=1e-5)
    rng = np.random.default_rng(0)
    n_utt = 100000
    tab = rng.standard_normal((n_utt, 512)).astype(np.float32)
    ids = [f"utt{i:06d}" for i in range(n_utt)]
    mega = dict(zip(ids, tab))
    num_to_id = dict(enumerate(ids))
    batches = [(torch.from_numpy(rng.integers(0, n_utt, B)), torch.from_numpy(rng.integers(0, n_utt, B)),
                torch.from_numpy((rng.random(B) < 0.1).astype(np.float32))) for _ in range(16)]
    phases = ["zero_grad", "to_device", "gather", "forward", "loss", "item", "backward", "adam"]
    acc = {k: 0.0 for k in phases}
    model.train()

    def body(k, timed):
        d1, d2, t = batches[k % len(batches)]
        t0 = time.perf_counter()
        opt.zero_grad()
        t1 = time.perf_counter()
        d1, d2, t = d1.to(dev), d2.to(dev), t.to(dev)
        t2 = time.perf_counter()
        x1, x2 = load_xvec_trials_from_numbatch(mega, num_to_id, d1, d2, dev)
        t3 = time.perf_counter()
        out = model(x1, x2)
        t4 = time.perf_counter()
        loss = model.loss(out, t)
        t5 = time.perf_counter()
        lv = loss.item()
        t6 = time.perf_counter()
        loss.backward()
        t7 = time.perf_counter()
        opt.step()
        t8 = time.perf_counter()
        if timed:
            for name, a, b in zip(phases, (t0, t1, t2, t3, t4, t5, t6, t7), (t1, t2, t3, t4, t5, t6, t7, t8)):
                acc[name] += b - a
        return lv

    for k in range(30):
        body(k, False)
    torch.cuda.synchronize()
    walls = []
    n = 0
    for rep in range(6):
        t0 = time.perf_counter()
        for k in range(50):
            body(k, True)
            n += 1
        torch.cuda.synchronize()
        walls.append((time.perf_counter() - t0) / 50)
    res = {"D": D, "B": B, "ms_per_step_median": 1e3 * float(np.median(walls)), "ms_per_step_min": 1e3 * min(walls),
           "host_us_per_phase": {k: 1e6 * v / n for k, v in acc.items()}}

    # the same body without the loop's own sync (loss.item()) and with device-resident index batches: how far the host alone
    # can run ahead
    dbatches = [(a.to(dev), b.to(dev), c.to(dev)) for a, b, c in batches]

    def body2(k):
        d1, d2, t = dbatches[k % len(dbatches)]
        opt.zero_grad()
        x1, x2 = load_xvec_trials_from_numbatch(mega, num_to_id, d1, d2, dev)
        loss = model.loss(model(x1, x2), t)
        loss.backward()
        opt.step()

    for k in range(10):
        body2(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(200):
        body2(k)
    torch.cuda.synchronize()
    res["ms_per_step_no_item_resident_indices"] = 1e3 * (time.perf_counter() - t0) / 200
    x1, x2 = load_xvec_trials_from_numbatch(mega, num_to_id, *dbatches[0][:2], dev)
    tt = dbatches[0][2]

    def body3():
        opt.zero_grad()
        loss = model.loss(model(x1, x2), tt)
        loss.backward()
        opt.step()

    for k in range(10):
        body3()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(200):
        body3()
    torch.cuda.synchronize()
    res["ms_per_step_model_loss_backward_adam_only"] = 1e3 * (time.perf_counter() - t0) / 200
    # torch.optim.Adam alone on these eight tensors
    for k in range(10):
        opt.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(200):
        opt.step()
    torch.cuda.synchronize()
    res["ms_torch_adam_step_alone"] = 1e3 * (time.perf_counter() - t0) / 200
    print(json.dumps(res, indent=1))
    if "--torchprof" in sys.argv:
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            for k in range(100):
                body3()
            torch.cuda.synchronize()
        print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=45, max_name_column_width=60))
    if "--cprofile" in sys.argv:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for k in range(300):
            body(k, False)
        torch.cuda.synchronize()
        pr.disable()
        st = pstats.Stats(pr, stream=sys.stdout)
        st.sort_stats("cumulative").print_stats(70)


if __name__ == "__main__":
    main()
