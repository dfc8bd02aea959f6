#!/usr/bin/env python3
"""FusedTrainStep replays at B = 4096 (BASELINE cfg2) for rocprofv3 --kernel-trace: what the graph's launches cost.

Three ways of feeding the step, each timed over 200 replays:
  inplace : the minibatch is written into the step's own input buffers (step.x1 / .x2 / .t) — no staging copies;
  copy    : foreign tensors -> three device copies (2 x 8 MB + 16 KB) in front of every replay;
  rows    : step_rows(table, rows1, rows2, t) — three index / label copies, the step's first kernel gathers the rows;
  rows1   : the same with the batch as ONE packed record (TrialLoader.device_batches(pack=True));
  records : begin_epoch / step_record — the records stay where the loader put them, the step walks them through its device
            cursor: the training loop's form (train.train).
(Run `rows` before `copy`: the first ~200 replays of the rows graph that follow a run of 8 MB device-to-device staging copies
take 0.27 ms each, then drop back to 0.10 ms — a runtime effect of switching between the copy engine and blit kernels on
the stream, not of the step's kernels; a training loop only ever uses step_rows.)
usage: fused_step_profile.py [D=150] [graph|eager] [modes=inplace,rows,copy] [B=4096] [table rows=200000]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from neuralplda_amd import models, train


class NC:
    xvector_dim, layer1_LDA_dim, layer2_PLDA_spkfactor_dim = 512, 150, 150
    beta, alpha, device, loss = [99.0, 199.0], 15.0, "cuda", "SoftCdet"


D = int(sys.argv[1]) if len(sys.argv) > 1 else 150
graph = (sys.argv[2] != "eager") if len(sys.argv) > 2 else True
modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["inplace", "rows1", "rows", "copy"]
NC.layer1_LDA_dim = NC.layer2_PLDA_spkfactor_dim = D
torch.manual_seed(0)
m = models.NeuralPlda(NC()).cuda()
B = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
x1 = torch.randn(B, 512, device="cuda"); x2 = torch.randn(B, 512, device="cuda")
t = (torch.rand(B, device="cuda") < 0.1).float()
table = torch.randn(int(sys.argv[5]) if len(sys.argv) > 5 else 200000, 512, device="cuda")
r1 = torch.randint(0, table.shape[0], (B,), device="cuda"); r2 = torch.randint(0, table.shape[0], (B,), device="cuda")
step = train.FusedTrainStep(m, 1e-4, weight_decay=1e-5, batch_size=B, graph=graph)


def timed(fn, n=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for mode in modes:
    if mode == "inplace" and graph:
        step.x1.copy_(x1); step.x2.copy_(x2); step.t.copy_(t)
        ms = timed(lambda: step(step.x1, step.x2, step.t))
    elif mode == "rows":
        ms = timed(lambda: step.step_rows(table, r1, r2, t))
    elif mode == "records":
        # 256 different minibatches (fresh table rows every step, as in an epoch)
        recs = torch.stack([torch.cat([torch.randint(0, table.shape[0], (B,), device="cuda").view(torch.uint8),
                                       torch.randint(0, table.shape[0], (B,), device="cuda").view(torch.uint8),
                                       t.view(torch.uint8)]) for _ in range(256)])

        def rec_step():
            if step._records_left == 0:
                step.begin_epoch(table, recs)
            step.step_record()
        ms = timed(rec_step)
    elif mode == "rows1":
        rec = torch.cat([r1.view(torch.uint8), r2.view(torch.uint8), t.view(torch.uint8)])
        ms = timed(lambda: step.step_rows(table, r1, r2, t, record=rec))
    else:
        ms = timed(lambda: step(x1, x2, t))
    print(f"D={D} B={B} graph={graph} feed={mode}: {ms:.4f} ms/step (wall)")
