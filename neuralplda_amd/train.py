"""Training / validation driver — counterpart of xvector_NeuralPlda_pytorch.py:30-181.

Same flow as the reference script (`train()` :30-52, `validate()` :56-83, `main_kaldiplda()` :88-181:
Adam(lr, weight_decay=1e-5), threshold initialisation from a held-out set, per-epoch validation, whole-module
pickle, score files, LR halving when the held-out minC rises three times in a row with the optimiser
re-created), with the batch gather, forward, loss, backward all on the device.  Two additions:

* `GraphedTrainStep` captures one whole optimisation step (zero_grad, forward, loss, backward, Adam) into a
  HIP graph (torch.cuda.graph): the step is ~10 kernel launches on ~1.8 GFLOP of math, i.e. host-launch-bound
  in eager mode, and a graph replay removes the Python / launch overhead.
* the configuration path is an argument instead of a literal (:97).
"""
import logging
import os
import pickle
import random
from datetime import datetime

import numpy as np
import torch
import torch.optim as optim

from .models import NeuralPlda
from .NpldaConf import NpldaConf
from .sv_trials_loaders import (TrialIndexDataset, TrialLoader, combine_trials_and_get_loader,
                                get_trials_loaders_dict, load_xvec_trials_from_numbatch, xvector_table)

__all__ = ["train", "validate", "GraphedTrainStep", "FusedTrainStep", "FusedDPldaStep", "HeadStepWithInputGrads", "main_kaldiplda", "main_dplda",
           "train_gaussian_backend"]


def train(nc, model, device, train_loader, mega_xvec_dict, num_to_id_dict, optimizer, epoch, valid_loaders=None,
          step_fn=None):
    """xvector_NeuralPlda_pytorch.py:30-52.  `step_fn(x1, x2, target) -> loss tensor` may replace the eager
    step (e.g. a GraphedTrainStep); the loss is read back once per logging interval, not every step."""
    model.train()
    losses = []
    device = torch.device(device)
    # Device-resident epoch: with the vectorised loader and the fused step the whole epoch's index arrays go to the device
    # once (mapped to x-vector table rows there), every batch is a view, and the gather runs inside the captured step
    # (FusedTrainStep.step_rows).  Same batches, same arithmetic as the generic path below — the host just stops being
    # the bottleneck (13 launches + 3 host-to-device copies per step otherwise).
    fast = (isinstance(step_fn, FusedTrainStep) and isinstance(train_loader, TrialLoader) and device.type == "cuda"
            and isinstance(train_loader.dataset, TrialIndexDataset) and train_loader.num_workers == 0
            and not train_loader.drop_last)
    dp = None
    if fast and step_fn._dp_call:
        # data parallel: every rank walks the SAME epoch (same RNG state on every rank) and takes its slice of each global
        # batch; the loader hands over the global label counts with it, so the step needs ONE collective (its all-reduce)
        # (the loader is the GLOBAL, unsharded one — the slicing happens in device_batches; rank and size are those of the
        # group the model was made data parallel on, and the epoch's seed is rank 0's)
        from . import dist as ndist
        dp_group = getattr(model, "_dp_group", None)
        dp = ndist.world(dp_group)
    if fast:
        table, row_map = _device_table(mega_xvec_dict, num_to_id_dict, device)
        bs = train_loader.batch_size
        same = step_fn.batch_size == bs and dp is None
        # decide the path BEFORE building anything: packed records are only made when something will consume them
        if same and step_fn.cursor_ok(table):
            # the whole epoch as packed records on the device; the captured step walks them through a device-side cursor
            # (no copy and no host write per step), the ragged last batch takes the eager step
            records, tail = train_loader.device_epoch(device, row_map)
            if not step_fn.records_ok(table, records):
                raise RuntimeError("device_epoch records do not fit the step that asked for them")
            nb = records.shape[0]
            if nb:
                step_fn.begin_epoch(table, records)
            # the steps between two progress lines go out several to a graph launch (step_records); a line is printed after
            # the steps 0, L, 2 L, ... like the per-batch loop (xvector_NeuralPlda_pytorch.py:44-50)
            L, idx = max(int(nc.log_interval), 1), 0
            while idx < nb:
                stop = idx if idx % L == 0 else min(nb - 1, (idx // L + 1) * L)
                step_fn.step_records(stop - idx + 1)
                idx = stop + 1
                if stop % L == 0:
                    _log_train(nc, epoch, stop, bs, train_loader, step_fn.pop_loss_mean())
            if tail is not None:
                step_fn.step_rows(table, *tail)
                if nb % nc.log_interval == 0:
                    _log_train(nc, epoch, nb, len(tail[0]), train_loader, step_fn.pop_loss_mean())
            return
        # graph replay from static index buffers: one record copy per step instead of three (records need an even batch
        # size: int64 fields at 20 * bs * k bytes); the eager step gathers from the views and never looks at a record
        pack = same and step_fn.use_graph and bs % 2 == 0
        if dp is not None:
            batches = ((r1, r2, t, None, gc) for r1, r2, t, gc in train_loader.device_batches(device, row_map, shard=dp, group=dp_group))
        elif pack:
            batches = ((r1, r2, t, None, rec)
                       for r1, r2, t, rec in train_loader.device_batches(device, row_map, pack=True))
        else:
            batches = ((r1, r2, t, None, None) for r1, r2, t in train_loader.device_batches(device, row_map))
    else:
        batches = ((None, None, t, d1, d2) for d1, d2, t in train_loader)
    for batch_idx, (rows1, rows2, target, data1, data2) in enumerate(batches):
        if fast:
            rec, data1 = data2, rows1  # (data2 carries the packed record here; len(data1) below)
            if dp is not None:  # (data2 carries the global batch's counts)
                step_fn.step_rows(table, rows1, rows2, target, global_counts=rec)
            else:
                step_fn.step_rows(table, rows1, rows2, target, record=rec)
            if batch_idx % nc.log_interval == 0:  # (the step keeps the interval's loss sum on the device)
                _log_train(nc, epoch, batch_idx, len(data1), train_loader, step_fn.pop_loss_mean())
            continue
        data1, data2, target = data1.to(device), data2.to(device), target.to(device)
        data1_xvec, data2_xvec = load_xvec_trials_from_numbatch(mega_xvec_dict, num_to_id_dict, data1, data2, device)
        if step_fn is not None and (isinstance(step_fn, FusedTrainStep)
                                    or data1_xvec.shape[0] == getattr(step_fn, "batch_size", -1)):
            loss = step_fn(data1_xvec, data2_xvec, target)
        else:
            optimizer.zero_grad()
            output = model(data1_xvec, data2_xvec)
            loss = model.loss(output, target)
            loss.backward()
            optimizer.step()
        fused = isinstance(step_fn, FusedTrainStep)
        if not fused:
            losses.append(loss.detach())
        if batch_idx % nc.log_interval == 0:
            _log_train(nc, epoch, batch_idx, len(data1), train_loader, step_fn.pop_loss_mean() if fused else losses)
            losses = []
    if device.type == "cuda":
        from . import ops as _ops
        if _ops.KEYERROR_DEFERRED:
            # compat.install(deferred_keyerror=True): the epoch's last batches have no "next loader call" to surface a bad
            # trial number — the end of the epoch is that call (ops.check_trial_indices: one synchronise per epoch)
            _ops.check_trial_indices(device)


def _device_table(mega_xvec_dict, num_to_id_dict, device):
    """(resident x-vector matrix, int64 device map trial number -> table row) of a mega dict / num_to_id dict pair."""
    tab = xvector_table(mega_xvec_dict)
    _, m, devmaps = tab.rows_from_nums(num_to_id_dict)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in devmaps:
        devmaps[key] = torch.from_numpy(m).to(device)
    return tab.on(device), devmaps[key]


def _log_train(nc, epoch, batch_idx, batch_len, train_loader, losses):
    """The progress line of xvector_NeuralPlda_pytorch.py:44-50 (mean of the losses since the previous line: a list of
    0-d tensors, each with storage of its own, or the mean itself)."""
    if isinstance(losses, float):
        mean_loss = losses
    else:  # sum(losses) / len(losses) of the reference, in double like its Python floats
        mean_loss = float(torch.stack([l.reshape(()).double() for l in losses]).sum().item()) / len(losses)
    msg = 'Train Epoch: {} [{}/{} ({:.0f}%)]\t {}: {:.6f}'.format(
        epoch, batch_idx * batch_len, len(train_loader.dataset), 100. * batch_idx / len(train_loader),
        nc.loss, mean_loss)
    print(msg)
    logging.info(msg)


_VALIDATE_CHUNK = {}


def _validate_chunk(model):
    """Pairs per scoring call of validate()'s device-resident pass: the largest list of c x 4096 pairs (c 16-pair tiles on
    each of 256 CUs) that nplda_score_pairs_rows_f32 scores with the gather folded in (the balanced-tile kernel: c = 8 k + 1
    is where it beats a part-filled streaming round), per model shape."""
    from . import _lib
    D1, D0 = model.centering_and_LDA.weight.shape
    D2 = model.centering_and_wccn_plda.weight.shape[0]
    key = (D0, D1, D2)
    if key not in _VALIDATE_CHUNK:
        lib, best = _lib.load(), 10240
        for c in (49, 41, 33, 25, 17, 9):
            name = lib.nplda_score_pairs_kernel_name(c * 4096, D0, D1, D2)
            if name and name.decode().startswith("nplda_fwd_mid_kernel"):
                best = c * 4096
                break
        _VALIDATE_CHUNK[key] = best
    return _VALIDATE_CHUNK[key]



def validate(nc, model, device, mega_xvec_dict, num_to_id_dict, data_loader, update_thresholds=False):
    """xvector_NeuralPlda_pytorch.py:56-83 (scores are collected in a list, not by repeated torch.cat)."""
    model.eval()
    if torch.device(device).type == "cuda":
        from . import ops as _ops
        if _ops.KEYERROR_DEFERRED:  # (compat.install(deferred_keyerror=True): the training batches' pending KeyError)
            _ops.check_trial_indices()
    with torch.no_grad():
        targets, scores = [], []
        device = torch.device(device)
        if (isinstance(data_loader, TrialLoader) and device.type == "cuda"
                and isinstance(data_loader.dataset, TrialIndexDataset) and data_loader.num_workers == 0
                and not data_loader.drop_last):
            # device-resident pass (see train()): same batches, same forward launches, no per-batch host copies
            from . import ops
            table, row_map = _device_table(mega_xvec_dict, num_to_id_dict, device)
            # NeuralPlda proper: the gather folded into the scoring kernel.  (DPlda inherits forward_rows but gathers whole
            # batches and has no PLDA layer: it keeps the loader-batched loop below.)
            fused = (getattr(type(model), "forward_rows", None) is NeuralPlda.forward_rows
                     and hasattr(model, "centering_and_wccn_plda"))
            # (file order: every metric below is a function of the set of (score, label) pairs, and the host-side
            # permutation of the epoch costs more than the whole pass; the global RNG moves on as an iteration moves it)
            if fused:
                # the gather is inside the kernel, nothing batch-sized is materialised: chunks sized for the KERNEL (the
                # largest list the balanced-tile kernel takes: 0.78 of the peak) instead of the loader's 5 x 2048 (0.55) —
                # 1 M trials in 6 launches instead of 103.  A pair's score does not depend on its neighbours.
                n, e1, e2, el, urows, j1, j2 = data_loader.device_columns_distinct(device, row_map)
                if (n > 2 * urows.numel() and getattr(model, "scoring_precision", "fp32") == "fp32"
                        and os.environ.get("NPLDA_VALIDATE_DENSE", "0") != "1"):
                    # a trial list over few utterances (a validation list names each one ~10 times): every distinct
                    # utterance embedded ONCE, the trials scored from the embedding table by index — what
                    # scorefile_generator does for score files (1 M trials over 200 k utterances: 3.2 ms -> 0.8 ms of
                    # GPU work).  NPLDA_VALIDATE_DENSE=1 keeps the dense pass (A/B measurements).
                    targets.append(el)
                    scores.append(model.forward_distinct(table, urows, j1, j2))
                    n = -1
                step = max(_validate_chunk(model), int(data_loader.batch_size or 1))
                for lo in range(0, max(n, 0), step):
                    targets.append(el[lo:lo + step])
                    scores.append(model.forward_rows(table, e1[lo:lo + step], e2[lo:lo + step]))
                if n == 0:
                    targets.append(el)
                    scores.append(model.forward_rows(table, e1, e2))
            else:
                for rows1, rows2, target in data_loader.device_batches(device, row_map, permute=False):
                    targets.append(target)
                    scores.append(model.forward(ops.gather_rows(table, rows1), ops.gather_rows(table, rows2)))
        else:
            for data1, data2, target in data_loader:
                data1, data2, target = data1.to(device), data2.to(device), target.to(device)
                x1, x2 = load_xvec_trials_from_numbatch(mega_xvec_dict, num_to_id_dict, data1, data2, device)
                targets.append(target)
                scores.append(model.forward(x1, x2))
        targets, scores = torch.cat(targets), torch.cat(scores)
        soft_cdet_loss = model.softcdet(scores, targets)
        cdet_mdl = model.cdet(scores, targets)
        minc, minc_threshold = model.minc(scores, targets, update_thresholds)
    lines = ['\n\nTest set: C_det (mdl): {:.4f}\n'.format(float(cdet_mdl)),
             'Test set: soft C_det (mdl): {:.4f}\n'.format(float(soft_cdet_loss)),
             'Test set: C_min: {:.4f}\n'.format(float(minc))]
    lines += ['Test set: argmin threshold [{}]: {:.4f}\n'.format(beta, float(minc_threshold[beta])) for beta in nc.beta]
    for ln in lines:
        logging.info(ln)
        print(ln)
    return minc, minc_threshold


class GraphedTrainStep:
    """One optimisation step (zero_grad -> forward -> loss -> backward -> optimizer.step) captured in a HIP graph.

    x1, x2: (B, D0) float32, target: (B,) float32 on the model's device, fixed B.  The optimiser must be
    capture-safe (torch.optim.Adam(..., capturable=True); `make_optimizer` builds one).  Returns the loss of the
    step as a 0-d device tensor of its own (a copy of the graph's output: callers collect losses in lists)."""

    def __init__(self, model, optimizer, batch_size, xvector_dim=None, warmup=3):
        p = next(model.parameters())
        if not p.is_cuda:
            raise ValueError("GraphedTrainStep needs the model on a HIP device")
        self.model, self.optimizer, self.batch_size = model, optimizer, int(batch_size)
        D0 = xvector_dim or model.centering_and_LDA.in_features
        dev = p.device
        self.x1 = torch.zeros(self.batch_size, D0, device=dev)
        self.x2 = torch.zeros(self.batch_size, D0, device=dev)
        self.t = torch.zeros(self.batch_size, device=dev)
        self.t[::2] = 1  # a valid label mix for the warm-up steps (both classes present)
        self._graph = None
        self._loss = None
        self._warmup = warmup

    def _step(self):
        self.optimizer.zero_grad(set_to_none=False)
        out = self.model(self.x1, self.x2)
        loss = self.model.loss(out, self.t)
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    def _capture(self):
        # warm-up on a side stream (allocator pools, lazy optimiser state), restoring the parameters afterwards
        state = [p.detach().clone() for p in self.model.parameters()]
        had_state = len(self.optimizer.state) > 0
        saved_opt = ({id(k): {n: (v.clone() if torch.is_tensor(v) else v) for n, v in st.items()}
                      for k, st in self.optimizer.state.items()} if had_state else None)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(self._warmup):
                self._step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        with torch.no_grad():
            for p, v in zip(self.model.parameters(), state):
                p.copy_(v)
        for k, st in self.optimizer.state.items():  # undo the warm-up's effect on the moments / step count
            for n, v in st.items():
                if torch.is_tensor(v):
                    if saved_opt is not None and id(k) in saved_opt:
                        v.copy_(saved_opt[id(k)][n])
                    else:
                        v.zero_()
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self._loss = self._step()

    def __call__(self, x1, x2, target):
        if self._graph is None:
            self._capture()
        self.x1.copy_(x1, non_blocking=True)
        self.x2.copy_(x2, non_blocking=True)
        self.t.copy_(target, non_blocking=True)
        self._graph.replay()
        # a replay rewrites the parameters on the device without touching any Python-side version counter, and the
        # model's packed-image cache (models._packed_for) is keyed on exactly those counters: bump them, or the next
        # forward() / validate() would score with the image of the weights from before the replay
        torch.autograd.graph.increment_version(list(self.model.parameters()))
        return self._loss.clone()


class FusedTrainStep:
    """The whole optimisation step as direct C-ABI launches, no autograd and no torch optimiser.  Up to 16384 pairs on
    one rank it is ONE call, nplda_train_step_f32: forward(train) -> data gradients with the loss folded in -> weight
    gradient slabs -> slab sums + Adam + refreshed parameter image + loss / dtheta (three launches: the first is forward, loss and data gradients in one kernel; the update rule is
    torch.optim.Adam(lr, weight_decay) of the reference, xvector_NeuralPlda_pytorch.py:139).  Larger batches and
    data-parallel models take the separate calls: pack -> forward(train) -> loss sums -> loss/g/dtheta -> backward (flat
    gradient) -> one-launch Adam (nplda_adam_step_f32).  Either form is optionally replayed from a HIP graph.
    With the `reduce_sums` / `reduce_flat` hooks of a data-parallel model (neuralplda_amd.dist.make_data_parallel) it is
    the data-parallel step: each rank feeds its shard of the global minibatch, the fp64 loss sums and the flat gradient
    are all-reduced (SUM) between the launches, and both collectives are captured INSIDE the HIP graph (RCCL collectives
    are stream operations), so the replayed step stays one graph launch per rank."""

    _one_call = False  # subclasses with their own kernels (FusedDPldaStep) keep the separate calls
    _dp_call = False
    _packed = _packed_key = None
    _loss_acc, _acc_n = None, 0  # fp64 device sum of the losses since pop_loss_mean(), number of steps in it
    _cursor = _stage = _graph_rec = _loss_rec = _graph_rec_table = _records_ref = None
    _graph_rec_multi = None
    records_per_replay = 8  # step_records: this many steps in ONE captured graph (a replay costs ~5 us of a 65 us step)
    _records_left = 0

    def __init__(self, model, lr, weight_decay=1e-5, betas=(0.9, 0.999), eps=1e-8, batch_size=None, graph=True):
        from . import _lib, ops
        from .models import _loss_kind
        self._lib, self._ops = _lib, ops
        p = next(model.parameters())
        if not p.is_cuda:
            raise ValueError("FusedTrainStep needs the model on a HIP device")
        self.model, self.dev = model, p.device
        self.lr, self.wd, self.betas, self.eps = float(lr), float(weight_decay), betas, float(eps)
        self.kind = _loss_kind(model.lossfn)
        self.thetas = ([model.threshold[b] for b in model.beta] if self.kind == ops.LOSS_SOFTCDET
                       else [model.threshold_Xent])
        self.betas_loss = [float(b) for b in model.beta] if self.kind == ops.LOSS_SOFTCDET else []
        self.alpha = model._alpha() if self.kind == ops.LOSS_SOFTCDET else 0.0
        self.params = list(model._params())
        D1, D0 = self.params[0].shape
        D2 = self.params[2].shape[0]
        self.dims = (D0, D1, D2)
        n = _lib.load().nplda_grad_floats(D0, D1, D2)
        K = len(self.thetas)
        self.m = torch.zeros(n + K, device=self.dev)
        self.v = torch.zeros(n + K, device=self.dev)
        self.step_count = torch.zeros(2, device=self.dev)  # [steps taken, launch scratch] (nplda_adam_step_f32)
        self.batch_size = batch_size
        self.use_graph = bool(graph) and batch_size is not None
        self._graph = None
        self._loss = None
        self._graph_rows = None
        self._loss_rows = None
        self._graph_table = None
        self.i1 = self.i2 = None
        self.reduce_sums = getattr(model, "_reduce_sums", None)
        self.reduce_flat = getattr(model, "_reduce_flat", None)
        # one-call step (nplda_train_step_f32): single rank, SoftCdet / BCE; its parameter image lives here and follows the
        # parameters from step to step (re-packed only when something else has touched them: version counters)
        self._one_call = (type(self) is FusedTrainStep and self.reduce_sums is None and self.reduce_flat is None
                          and self.kind in (ops.LOSS_SOFTCDET, ops.LOSS_BCE))
        # data-parallel one-collective step (nplda_train_step_grad_f32 -> ONE all-reduce -> nplda_train_step_apply_f32): the
        # same kernels as the one-call step; dL/ds needs only the GLOBAL batch's counts, which the caller passes (or a
        # 16-byte all-reduce in front finds)
        self._dp_call = (type(self) is FusedTrainStep and self.reduce_flat is not None
                         and self.kind in (ops.LOSS_SOFTCDET, ops.LOSS_BCE))
        self.gcount = torch.zeros(2, dtype=torch.float64, device=self.dev) if self._dp_call else None
        self._flat = None
        self._packed = None
        self._packed_key = None
        self._ws = {}
        self._loss_buf = torch.zeros((), device=self.dev)
        if self.use_graph:
            self.x1 = torch.zeros(batch_size, D0, device=self.dev)
            self.x2 = torch.zeros(batch_size, D0, device=self.dev)
            self.t = torch.zeros(batch_size, device=self.dev)
            self.t[::2] = 1

    def _acc(self):
        if self._loss_acc is None:
            self._loss_acc = torch.zeros(1, dtype=torch.float64, device=self.dev)
        return self._loss_acc

    def _account(self, loss, B):
        """Every step adds its loss to a device-side fp64 sum: the one-call step inside its last kernel (loss_sum of
        nplda_train_step_f32), the separate-launch forms here.  A replayed step hands out ONE loss tensor, rewritten by
        every replay, so the losses of a logging interval cannot be collected as a list of tensors."""
        self._acc_n += 1
        if isinstance(loss, tuple):  # (loss, dx1, dx2) of a step that also returns input gradients
            loss = loss[0]
        if self.__dict__.get("_acc_in_kernel", False):  # (FusedDPldaStep's recipe step: nplda_dplda_update_f32 added it)
            return
        if not ((self._one_call and 0 < B <= 16384) or (self._dp_call and B <= 16384)):
            self._acc().add_(loss.detach().reshape(1))

    def pop_loss_mean(self):
        """Mean of the losses of the steps since the previous call (the figure of the reference's progress line,
        xvector_NeuralPlda_pytorch.py:41-47: sum(losses) / len(losses)); one device read-back, then the sum starts over."""
        n, self._acc_n = self._acc_n, 0
        if n == 0:
            return float("nan")
        total = float(self._acc().item())
        self._loss_acc.zero_()
        return total / n

    def _sync_packed(self):
        """The step's parameter image, re-packed if anything but the step itself changed the parameters."""
        key = tuple(q._version for q in self.params)
        if self._packed is None:
            self._packed = self._ops.pack_params(*[q.detach() for q in self.params])
        elif key != self._packed_key:
            self._ops.pack_params_into(self._packed, *[q.detach() for q in self.params])
        self._packed_key = key

    def _pre_replay(self, force=False):
        """Hook in front of a graph replay / capture for steps that carry state derived from the parameters (FusedDPldaStep's
        quadratic-form image): bring it up to date if something else has touched the parameters."""

    def _touched(self):
        """The raw kernels have rewritten the parameters: bump their version counters (autograd's saved-tensor
        checks, the model's packed-image cache); the step's own image was refreshed by the same launch."""
        torch.autograd.graph.increment_version(self.params + self.thetas)
        self._packed_key = tuple(q._version for q in self.params)

    def _one_call_step(self, x1, x2, t):
        ops = self._ops
        B = x1.shape[0]
        ws = self._ws.get(B)
        if ws is None:
            ws = self._ws[B] = ops.train_step_workspace(B, self._packed)
        with torch.no_grad():
            ops.train_step(x1, x2, t, [q.detach() for q in self.params], [th.detach() for th in self.thetas],
                           self.betas_loss, self.alpha, self.kind, self.m, self.v, self.step_count, self.lr, self.betas[0],
                           self.betas[1], self.eps, self.wd, self._packed, ws, self._loss_buf, loss_sum=self._acc())
        # the captured step hands out its static output; an eager step a tensor of its own (callers keep losses around)
        return self._loss_buf if torch.cuda.is_current_stream_capturing() else self._loss_buf.clone()

    def _dp_step(self, x1, x2, t, table=None):
        """This rank's shard of the global minibatch: gradient phase -> ONE SUM all-reduce of [flat gradient | loss sums]
        (captured with the kernels when the step is a graph) -> update phase.  self.gcount holds the global [N_t, N_n].
        With `table`, x1 / x2 are row indices into it and the first kernel gathers the rows itself."""
        ops = self._ops
        B = x1.shape[0]
        key = ("rows", B) if table is not None else B
        ws = self._ws.get(key)
        if ws is None and B > 0:
            ws = self._ws[key] = ops.train_step_workspace(B, self._packed, rows=table is not None)
        if self._flat is None:
            self._flat = torch.zeros(ops.train_step_flat_floats(self._packed), device=self.dev)
        with torch.no_grad():
            prm, ths = [q.detach() for q in self.params], [th.detach() for th in self.thetas]
            if B == 0:
                # an empty shard (the ragged last batch cut over more ranks than it has rows): this rank contributes a zero
                # gradient and zero loss sums but issues the SAME single all-reduce as every other rank, and counts the
                # step the gradient kernel would have counted (Adam's bias correction must agree on every rank)
                self._flat.zero_()
                self.step_count[0] += 1
            elif table is not None:
                ops.train_step_grad_rows(table, x1, x2, t, prm, ths, self.betas_loss, self.alpha, self.kind, self.step_count,
                                         self._packed, ws, self._flat, self.gcount)
            else:
                ops.train_step_grad(x1, x2, t, prm, ths, self.betas_loss, self.alpha, self.kind, self.step_count,
                                    self._packed, ws, self._flat, self.gcount)
            self.reduce_flat(self._flat)
            ops.train_step_apply(self._flat, prm, ths, self.betas_loss, self.alpha, self.kind, self.m, self.v, self.step_count,
                                 self.lr, self.betas[0], self.betas[1], self.eps, self.wd, self._packed, self._loss_buf,
                                 loss_sum=self._acc())
        return self._loss_buf if torch.cuda.is_current_stream_capturing() else self._loss_buf.clone()

    def set_global_counts(self, target=None, counts=None):
        """Data parallel: the GLOBAL minibatch's [N_t, N_n] for the next step.  `counts` (two numbers or a tensor) when the
        caller knows them — a loader that slices its shard out of the global batch does; else they are found from this
        rank's `target` with a 16-byte all-reduce."""
        if not self._dp_call:
            return
        if counts is not None:
            c = counts if torch.is_tensor(counts) else torch.tensor([float(counts[0]), float(counts[1])], dtype=torch.float64)
            self.gcount.copy_(c.to(torch.float64).reshape(2), non_blocking=True)
        else:
            nt = target.detach().double().sum().reshape(1)
            c = torch.cat([nt, float(target.shape[0]) - nt])
            self.gcount.copy_(self.reduce_sums(c) if self.reduce_sums is not None else c)

    def _eager(self, x1, x2, t):
        ops = self._ops
        D0, D1, D2 = self.dims
        if self._one_call and 0 < x1.shape[0] <= 16384:
            if self._packed is None or not torch.cuda.is_current_stream_capturing():
                self._sync_packed()
            return self._one_call_step(x1, x2, t)
        if self._dp_call and x1.shape[0] <= 16384:  # (0 rows included: every rank makes the step's one collective)
            if self._packed is None or not torch.cuda.is_current_stream_capturing():
                self._sync_packed()
            return self._dp_step(x1, x2, t)
        with torch.no_grad():
            prm = [q.detach() for q in self.params]
            packed = ops.pack_params(*prm)
            s, saved = ops.forward_train(x1, x2, packed)
            ths = [th.detach() for th in self.thetas]
            if self.reduce_sums is None:
                loss, g, dth, _ = ops.loss_fwd_bwd(s, t, ths, self.betas_loss, self.alpha, self.kind)
            else:
                sums = self.reduce_sums(ops.loss_sums(s, t, ths, self.alpha, self.kind))
                loss, g, dth = ops.loss_finish(s, t, ths, self.betas_loss, self.alpha, self.kind, sums)
            flat = ops.backward(saved, g, packed, prm[4])
            if self.reduce_flat is not None:
                flat = self.reduce_flat(flat)
            # segments: the six tensors of the flat gradient, then one scalar per threshold
            grads = list(ops.split_flat_grad(flat, D0, D1, D2)) + [dth[k:k + 1] for k in range(len(ths))]
            self._adam(prm + ths, grads)
        return loss

    def _adam(self, tensors, grads):
        """One nplda_adam_step_f32 launch over `tensors` (moments in self.m / self.v, segment by segment)."""
        import ctypes
        lib = self._lib.load()
        nseg = len(tensors)
        arr = lambda ptrs: (ctypes.c_void_p * nseg)(*ptrs)  # noqa: E731
        offs, o = [], 0
        for q in tensors:
            offs.append(o)
            o += q.numel()
        code = lib.nplda_adam_step_f32(
            arr([q.data_ptr() for q in tensors]), arr([gq.data_ptr() for gq in grads]),
            arr([self.m.data_ptr() + 4 * of for of in offs]), arr([self.v.data_ptr() + 4 * of for of in offs]),
            (ctypes.c_int64 * nseg)(*[q.numel() for q in tensors]), nseg, self.step_count.data_ptr(), self.lr,
            self.betas[0], self.betas[1], self.eps, self.wd, self._lib.current_stream())
        self._lib.check(code, "nplda_adam_step_f32")

    def _capture_fn(self, fn):
        """Warm `fn` up on a side stream (optimiser state restored afterwards), then capture one call of it."""
        state = [q.detach().clone() for q in self.params + self.thetas]
        m0, v0, s0, a0 = self.m.clone(), self.v.clone(), self.step_count.clone(), self._acc().clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.no_grad():
            for q, val in zip(self.params + self.thetas, state):
                q.copy_(val)
            self.m.copy_(m0)
            self.v.copy_(v0)
            self.step_count.copy_(s0)
            self._loss_acc.copy_(a0)
        if self._one_call or self._dp_call:
            self._sync_packed()  # the warm-up moved the image along with the parameters: back to the restored values
        self._pre_replay(force=True)
        graph = torch.cuda.CUDAGraph()
        # with collectives in the step the RCCL watchdog thread polls events while we capture: only this thread's calls
        # may be checked against the capture
        dp = self.reduce_sums is not None or self.reduce_flat is not None
        with torch.cuda.graph(graph, **({"capture_error_mode": "thread_local"} if dp else {})):
            loss = fn()
        return graph, loss

    def _capture(self):
        self._graph, self._loss = self._capture_fn(lambda: self._eager(self.x1, self.x2, self.t))

    def step_rows(self, table, rows1, rows2, target, record=None, global_counts=None):
        """One step on the pairs (table[rows1], table[rows2]): `table` is the resident (N, D0) x-vector matrix, rows
        int64 device tensors.  With graph replay the gather is part of the captured step (its first kernel reads the rows
        through the indices) and the indices sit in static buffers, so a step costs three small device copies — ONE when
        `record` (TrialLoader.device_batches(pack=True): [rows1 | rows2 | labels] as one uint8 tensor) is given — and one
        graph launch on the host."""
        ops = self._ops
        B = rows1.shape[0]
        if self._dp_call and B <= 16384:
            self.set_global_counts(target, global_counts)
        if not self.use_graph or B != self.batch_size:
            loss = self._eager(ops.gather_rows(table, rows1), ops.gather_rows(table, rows2), target)
            self._touched()
            self._account(loss, B)
            return loss
        if self._graph_rows is None or self._graph_table != (table.data_ptr(), table.shape, table.stride(0)):
            self._capture_rows(table)
        if self._one_call or self._dp_call:
            self._sync_packed()
        if record is not None:
            self._rec.copy_(record, non_blocking=True)
        else:
            self.i1.copy_(rows1, non_blocking=True)
            self.i2.copy_(rows2, non_blocking=True)
            self._t_rows.copy_(target, non_blocking=True)
        self._graph_rows.replay()
        self._touched()
        self._account(self._loss_rows, B)
        return self._loss_rows

    # ---- a device-resident epoch: the batches as packed records, consumed through a device-side cursor ----------------
    def cursor_ok(self, table):
        """True if this step can walk an epoch of packed records on `table`'s device (begin_epoch / step_record): the
        one-call graph-replayed step, 4 | B <= 16384, 16 | D0.  Knowable before any record is built."""
        B = self.batch_size
        return (self._one_call and self.use_graph and B is not None and 0 < B <= 16384 and B % 4 == 0
                and self.dims[0] % 16 == 0 and table.device == self.dev)

    def records_ok(self, table, records):
        """True if step_record can run `records` ((nb, 20 B) uint8: TrialLoader.device_epoch) against `table`."""
        return (self.cursor_ok(table) and records is not None and records.dim() == 2
                and records.shape[1] == 20 * self.batch_size and records.dtype == torch.uint8
                and records.is_contiguous() and records.device == table.device)

    def begin_epoch(self, table, records):
        """Point the step at an epoch of packed batch records ([rows1 | rows2 | labels] each, back to back on the device).
        Every step_record() then trains on the next record: the captured graph reads its batch from the step's staging
        record, and the graph's own last kernel copies the epoch's next record there, counted by a three-word device cursor
        (nplda_train_step_records_f32) — no copy launch and no host write per step."""
        if not self.records_ok(table, records):
            raise ValueError("begin_epoch: records / table do not fit this step (see records_ok)")
        B = self.batch_size
        if self._cursor is None:
            self._cursor = torch.zeros(3, dtype=torch.int64, device=self.dev)
            self._stage = torch.zeros(20 * B, dtype=torch.uint8, device=self.dev)
        if self._graph_rec is None or self._graph_rec_table != (table.data_ptr(), table.shape, table.stride(0)):
            # warm-up and capture against stand-in records (row 0, both classes present)
            warm = torch.zeros((3, 20 * B), dtype=torch.uint8, device=self.dev)
            warm[:, 16 * B:].view(torch.float32)[:, ::2] = 1
            self._set_epoch(warm)
            self._sync_packed()
            self._graph_rec, self._loss_rec = self._capture_fn(lambda: self._eager_records(table))
            # the same step records_per_replay times in one graph: every step's last kernel stages the next record, so the
            # steps chain on the device exactly as single replays do (step_records)
            self._set_epoch(warm)
            self._sync_packed()
            k = int(self.records_per_replay)
            self._graph_rec_multi = (self._capture_fn(lambda: [self._eager_records(table) for _ in range(k)][-1])[0]
                                     if k > 1 else None)
            self._graph_rec_table = (table.data_ptr(), table.shape, table.stride(0))
            self._table_ref = table
        self._set_epoch(records)
        self._records_ref, self._records_left = records, records.shape[0]

    def _set_epoch(self, records):
        """Stage record 0; cursor = [address of record 0, next record to stage, record count]."""
        self._stage.copy_(records[0])
        self._cursor.copy_(torch.tensor([records.data_ptr(), 0, records.shape[0]], dtype=torch.int64))
        torch.cuda.current_stream().synchronize()  # (the cursor's source is a temporary host tensor)

    def _eager_records(self, table):
        ops = self._ops
        B = self.batch_size
        ws = self._ws.get(("rows", B))
        if ws is None:
            ws = self._ws[("rows", B)] = ops.train_step_workspace(B, self._packed, rows=True)
        with torch.no_grad():
            ops.train_step_records(table, self._cursor, self._stage, B, [q.detach() for q in self.params],
                                   [th.detach() for th in self.thetas], self.betas_loss, self.alpha, self.kind, self.m,
                                   self.v, self.step_count, self.lr, self.betas[0], self.betas[1], self.eps, self.wd,
                                   self._packed, ws, self._loss_buf, loss_sum=self._acc())
        return self._loss_buf

    def step_record(self):
        """One step on the epoch's next record (begin_epoch): one graph launch on the host."""
        if self._records_left <= 0:
            raise RuntimeError("step_record: the epoch's records are used up (begin_epoch)")
        self._records_left -= 1
        self._sync_packed()
        self._graph_rec.replay()
        self._touched()
        self._account(self._loss_rec, self.batch_size)
        return self._loss_rec

    def step_records(self, n):
        """n steps on the epoch's next n records: records_per_replay steps per graph launch while that many are asked for,
        single steps for the rest.  Same arithmetic, same parameters as n calls of step_record()."""
        if n < 0 or n > self._records_left:
            raise RuntimeError("step_records: the epoch has fewer records left (begin_epoch)")
        k = int(self.records_per_replay)
        while n > 0:
            if self._graph_rec_multi is not None and n >= k:
                self._records_left -= k
                self._sync_packed()
                self._graph_rec_multi.replay()
                self._touched()
                for _ in range(k):
                    self._account(self._loss_rec, self.batch_size)
                n -= k
            else:
                self.step_record()
                n -= 1
        return self._loss_rec

    def _eager_rows(self, table):
        ops = self._ops
        if self._one_call and 0 < self.batch_size <= 16384 and self.dims[0] % 16 == 0:
            # the step's first kernel gathers the rows itself (nplda_train_step_rows_f32): no gather launches
            if self._packed is None or not torch.cuda.is_current_stream_capturing():
                self._sync_packed()
            ws = self._ws.get(("rows", self.batch_size))
            if ws is None:
                ws = self._ws[("rows", self.batch_size)] = ops.train_step_workspace(self.batch_size, self._packed, rows=True)
            with torch.no_grad():
                ops.train_step_rows(table, self.i1, self.i2, self._t_rows, [q.detach() for q in self.params],
                                    [th.detach() for th in self.thetas], self.betas_loss, self.alpha, self.kind, self.m,
                                    self.v, self.step_count, self.lr, self.betas[0], self.betas[1], self.eps, self.wd,
                                    self._packed, ws, self._loss_buf, loss_sum=self._acc())
            return self._loss_buf if torch.cuda.is_current_stream_capturing() else self._loss_buf.clone()
        if self._dp_call and 0 < self.batch_size <= 16384 and self.dims[0] % 16 == 0:
            if self._packed is None or not torch.cuda.is_current_stream_capturing():
                self._sync_packed()
            return self._dp_step(self.i1, self.i2, self._t_rows, table=table)
        ops.gather_rows(table, self.i1, out=self.x1)
        ops.gather_rows(table, self.i2, out=self.x2)
        return self._eager(self.x1, self.x2, self._t_rows)

    def _capture_rows(self, table):
        if self.i1 is None:
            # one record [rows1 | rows2 | labels]: the three static inputs of the captured step are views of it
            bs = self.batch_size
            self._rec = torch.zeros(20 * bs, dtype=torch.uint8, device=self.dev)
            self.i1 = self._rec[:8 * bs].view(torch.int64)
            self.i2 = self._rec[8 * bs:16 * bs].view(torch.int64)
            self._t_rows = self._rec[16 * bs:].view(torch.float32)
        self._graph_rows, self._loss_rows = self._capture_fn(lambda: self._eager_rows(table))
        self._graph_table = (table.data_ptr(), table.shape, table.stride(0))
        self._table_ref = table  # keeps the captured pointer alive

    def __call__(self, x1, x2, target, global_counts=None):
        """global_counts (data parallel only): [N_t, N_n] of the GLOBAL minibatch this rank's shard was cut from; without
        it a 16-byte all-reduce of the shard's counts comes first (see set_global_counts)."""
        if self._dp_call and 0 < x1.shape[0] <= 16384:
            self.set_global_counts(target, global_counts)
        if not self.use_graph or x1.shape[0] != self.batch_size:
            loss = self._eager(x1, x2, target)
            self._touched()
            self._account(loss, x1.shape[0])
            return loss
        if self._graph is None:
            self._capture()
        if self._one_call or self._dp_call:
            self._sync_packed()
        self._pre_replay()
        # a caller that fills the step's own input buffers (step.x1 / .x2 / .t, e.g. gather_rows(..., out=step.x1)) skips
        # the staging copies: at 4096 x 512 they are 2 x 8 MB, 16 us of a 70 us step
        if x1 is not self.x1:
            self.x1.copy_(x1, non_blocking=True)
        if x2 is not self.x2:
            self.x2.copy_(x2, non_blocking=True)
        if target is not self.t:
            self.t.copy_(target, non_blocking=True)
        self._graph.replay()
        self._touched()
        self._account(self._loss, self.batch_size)
        return self._loss


class HeadStepWithInputGrads(FusedTrainStep):
    """The head's step of an end-to-end fine-tune (BASELINE configs[4]; the reference's Etdnn_Xvec_NeuralPlda chains an
    extractor into this head, utils/models.py:251-268): x-vectors that an extractor produced (any float dtype, typically
    bf16) go through NeuralPlda.forward and the loss, the head takes its Adam step, and dL/dx1, dL/dx2 come back in the
    inputs' dtype for the extractor's own backward.  `step(x1, x2, target) -> (loss, dx1, dx2)`; direct C-ABI launches
    replayed from a HIP graph (static input / output buffers), no autograd.  Same arithmetic as
    loss = model.loss(model(x1.float(), x2.float()), t); loss.backward(); optimizer.step() with x requiring grad."""

    def __init__(self, model, lr, weight_decay=1e-5, betas=(0.9, 0.999), eps=1e-8, batch_size=None, graph=True,
                 dtype=torch.bfloat16):
        super(HeadStepWithInputGrads, self).__init__(model, lr, weight_decay, betas, eps, batch_size=None, graph=False)
        self.batch_size, self.io_dtype = batch_size, dtype
        self.use_graph = bool(graph) and batch_size is not None
        self._g = None
        # data parallel (the head under an extractor's DDP): the one-collective form of the step — gradient phase with the
        # global label counts (dL/dx of this rank's rows is complete there), all-reduce of [flat gradient | loss sums], update
        self._dp_head = self.reduce_flat is not None and self.kind in (self._ops.LOSS_SOFTCDET, self._ops.LOSS_BCE)
        if self._dp_head:
            self.gcount = torch.zeros(2, dtype=torch.float64, device=self.dev)
        if self.use_graph:
            D0 = self.dims[0]
            self.x1 = torch.zeros(batch_size, D0, device=self.dev, dtype=dtype)
            self.x2 = torch.zeros(batch_size, D0, device=self.dev, dtype=dtype)
            self.t = torch.zeros(batch_size, device=self.dev)
            self.t[::2] = 1

    def _shape_ok(self, x1):
        D0, D1, D2 = self.dims
        return (self.kind in (self._ops.LOSS_SOFTCDET, self._ops.LOSS_BCE) and 0 < x1.shape[0] <= 16384
                and (x1.dtype == torch.float32 or (x1.dtype == torch.bfloat16 and D0 == 512 and max(D1, D2) > 144)))

    def _fused_ok(self, x1):
        return self.reduce_sums is None and self.reduce_flat is None and self._shape_ok(x1)

    def _dp_ok(self, x1):
        return self._dp_head and self._shape_ok(x1)

    def set_global_counts(self, target=None, counts=None):
        """As FusedTrainStep.set_global_counts: the GLOBAL minibatch's [N_t, N_n] for the next step."""
        if not self._dp_head:
            return
        if counts is not None:
            c = counts if torch.is_tensor(counts) else torch.tensor([float(counts[0]), float(counts[1])], dtype=torch.float64)
            self.gcount.copy_(c.to(torch.float64).reshape(2), non_blocking=True)
        else:
            nt = target.detach().double().sum().reshape(1)
            c = torch.cat([nt, float(target.shape[0]) - nt])
            self.gcount.copy_(self.reduce_sums(c) if self.reduce_sums is not None else c)

    def describe(self):
        if self._fused_ok(self.x1 if self.use_graph else torch.empty(1, self.dims[0], dtype=self.io_dtype)):
            return ("nplda_train_step_dx_f32: forward + loss + data gradients + dx = du . W1 in ONE kernel (bf16 rows widened in "
                    "registers, bf16 out) | weight-gradient slabs | slab sums + Adam + image" +
                    (", one HIP-graph replay" if self.use_graph else ""))
        return ("x.float() x2, pack, nplda_forward_train_f32, nplda_loss_fwd_bwd_f32, nplda_backward_ex_f32 (flat gradient + "
                "dx1, dx2), nplda_adam_step_f32, dx.to(dtype) x2" + (", one HIP-graph replay" if self.use_graph else ""))

    def _eager(self, x1, x2, t):
        ops = self._ops
        D0, D1, D2 = self.dims
        if self._fused_ok(x1):
            B = x1.shape[0]
            if self._packed is None or not torch.cuda.is_current_stream_capturing():
                self._sync_packed()
            key = ("dx", B, x1.dtype)
            st = self._ws.get(key)
            if st is None:
                st = self._ws[key] = (ops.train_step_dx_workspace(B, self._packed, x1.dtype == torch.bfloat16),
                                      torch.empty_like(x1), torch.empty_like(x2))
            ws, dx1, dx2 = st
            x1c = x1 if x1.is_contiguous() else x1.contiguous()
            x2c = x2 if x2.is_contiguous() else x2.contiguous()
            tc = t.float().contiguous()
            with torch.no_grad():
                ops.train_step_dx(x1c, x2c, tc, [q.detach() for q in self.params], [th.detach() for th in self.thetas],
                                  self.betas_loss, self.alpha, self.kind, self.m, self.v, self.step_count, self.lr,
                                  self.betas[0], self.betas[1], self.eps, self.wd, self._packed, ws, self._loss_buf,
                                  dx1, dx2, loss_sum=self._acc())
            if torch.cuda.is_current_stream_capturing():
                return self._loss_buf, dx1, dx2
            return self._loss_buf.clone(), dx1.clone(), dx2.clone()
        if self._dp_ok(x1):
            B = x1.shape[0]
            if self._packed is None or not torch.cuda.is_current_stream_capturing():
                self._sync_packed()
            key = ("dx", B, x1.dtype)
            st = self._ws.get(key)
            if st is None:
                st = self._ws[key] = (ops.train_step_dx_workspace(B, self._packed, x1.dtype == torch.bfloat16),
                                      torch.empty_like(x1), torch.empty_like(x2))
            ws, dx1, dx2 = st
            if self._flat is None:
                self._flat = torch.zeros(ops.train_step_flat_floats(self._packed), device=self.dev)
            x1c = x1 if x1.is_contiguous() else x1.contiguous()
            x2c = x2 if x2.is_contiguous() else x2.contiguous()
            tc = t.float().contiguous()
            with torch.no_grad():
                prm, ths = [q.detach() for q in self.params], [th.detach() for th in self.thetas]
                ops.train_step_grad_dx(x1c, x2c, tc, prm, ths, self.betas_loss, self.alpha, self.kind, self.step_count,
                                       self._packed, ws, self._flat, dx1, dx2, self.gcount)
                self.reduce_flat(self._flat)
                ops.train_step_apply(self._flat, prm, ths, self.betas_loss, self.alpha, self.kind, self.m, self.v,
                                     self.step_count, self.lr, self.betas[0], self.betas[1], self.eps, self.wd, self._packed,
                                     self._loss_buf, loss_sum=self._acc())
            if torch.cuda.is_current_stream_capturing():
                return self._loss_buf, dx1, dx2
            return self._loss_buf.clone(), dx1.clone(), dx2.clone()
        with torch.no_grad():
            prm = [q.detach() for q in self.params]
            packed = ops.pack_params(*prm)
            s, saved = ops.forward_train(x1.float(), x2.float(), packed)
            ths = [th.detach() for th in self.thetas]
            if self.reduce_sums is None:
                loss, g, dth, _ = ops.loss_fwd_bwd(s, t, ths, self.betas_loss, self.alpha, self.kind)
            else:  # data parallel, outside the fused forms: global loss sums, then the summed flat gradient
                sums = self.reduce_sums(ops.loss_sums(s, t, ths, self.alpha, self.kind))
                loss, g, dth = ops.loss_finish(s, t, ths, self.betas_loss, self.alpha, self.kind, sums)
            flat, dx1, dx2 = ops.backward(saved, g, packed, prm[4], want_dx=True)
            if self.reduce_flat is not None:
                flat = self.reduce_flat(flat)
            grads = list(ops.split_flat_grad(flat, D0, D1, D2)) + [dth[k:k + 1] for k in range(len(ths))]
            self._adam(prm + ths, grads)
        return loss, dx1.to(x1.dtype), dx2.to(x2.dtype)

    def _account_step(self, x1, loss):
        self._acc_n += 1
        if not (self._fused_ok(x1) or self._dp_ok(x1)):  # (the fused calls add their loss to the device-side sum themselves)
            self._acc().add_(loss.detach().reshape(1))

    def __call__(self, x1, x2, target, global_counts=None):
        """global_counts (data parallel): [N_t, N_n] of the GLOBAL minibatch (else a 16-byte all-reduce finds them)."""
        B = x1.shape[0]
        if self._dp_ok(x1):
            self.set_global_counts(target, global_counts)
        if not self.use_graph or B != self.batch_size or x1.dtype != self.io_dtype:
            out = self._eager(x1, x2, target)
            self._touched()
            self._account_step(x1, out[0])
            return out
        fused = self._fused_ok(self.x1) or self._dp_ok(self.x1)
        if self._g is None:
            if fused:
                self._sync_packed()
            self._g, self._out = self._capture_fn(lambda: self._eager(self.x1, self.x2, self.t))
        if fused:
            self._sync_packed()
        # a producer that writes into the step's own input buffers (step.x1 / .x2 / .t — e.g. the extractor's last layer
        # with out=step.x1) skips the staging copies: 3 x ~4 us on a 0.08 ms step
        if x1 is not self.x1:
            self.x1.copy_(x1, non_blocking=True)
        if x2 is not self.x2:
            self.x2.copy_(x2, non_blocking=True)
        if target is not self.t:
            self.t.copy_(target, non_blocking=True)
        self._g.replay()
        self._touched()
        self._account_step(self.x1, self._out[0])
        return self._out


class FusedDPldaStep(FusedTrainStep):
    """The recipe step of DPlda (xvector_DPlda_pytorch.py:140-152: LDA frozen, Adam on the linear unit and the
    thresholds) as direct launches: quadratic-form image -> fused LDA + normalise + score (saving the paired rows)
    -> loss sums / finish -> weighted moments sum_k g_k x x^T -> fold into d wlr, d bias -> one-launch Adam.
    Same interface as FusedTrainStep (call, step_rows, graph replay); follows the autograd + torch.optim.Adam
    trajectory of models.DPlda (tests/test_dplda_gpu.py)."""

    def __init__(self, model, lr, weight_decay=1e-5, betas=(0.9, 0.999), eps=1e-8, batch_size=None, graph=True,
                 train_lda=False, want_dx=False):
        """train_lda: the LDA layer trains too (a joint fine-tune; the recipe itself freezes it); want_dx: the step also
        returns dL/dx1, dL/dx2 — `step(x1, x2, t) -> (loss, dx1, dx2)` — for an extractor upstream.  Either adds the input
        side of the backward to the captured step (dL/d[y1; y2] = g ((M + M^T) x + v) as one resident-matrix GEMM on the
        paired rows, the F.normalize backward, the LDA wgrad / dgrad GEMMs: the launches autograd makes one by one)."""
        from . import _lib, ops
        from .models import _loss_kind
        self._lib, self._ops = _lib, ops
        p = model.logistic_regres.weight
        if not p.is_cuda:
            raise ValueError("FusedDPldaStep needs the model on a HIP device")
        self.train_lda, self.want_dx = bool(train_lda), bool(want_dx)
        if not self.train_lda and (model.centering_and_LDA.weight.requires_grad or model.centering_and_LDA.bias.requires_grad):
            raise ValueError("DPlda trains with centering_and_LDA frozen (xvector_DPlda_pytorch.py:140-147): "
                             "set requires_grad = False on its weight and bias, or pass train_lda=True")
        if (self.train_lda or self.want_dx) and model.centering_and_LDA.out_features % 2:
            raise ValueError("a backward through DPlda's LDA needs an even layer1_LDA_dim (16-byte paired rows)")
        self.model, self.dev = model, p.device
        self.lr, self.wd, self.betas, self.eps = float(lr), float(weight_decay), betas, float(eps)
        self.kind = _loss_kind(model.lossfn)
        # DPlda's BCE has no threshold (utils/models.py:503-506): the loss kernels get a constant zero, nothing to train
        self.thetas = [model.threshold[b] for b in model.beta] if self.kind == ops.LOSS_SOFTCDET else []
        self._zero = torch.zeros(1, device=p.device)
        self.betas_loss = [float(b) for b in model.beta] if self.kind == ops.LOSS_SOFTCDET else []
        self.alpha = float(model.alpha) if self.kind == ops.LOSS_SOFTCDET else 0.0
        self.params = [model.logistic_regres.weight, model.logistic_regres.bias]
        if self.train_lda:
            self.params += [model.centering_and_LDA.weight, model.centering_and_LDA.bias]
        D0, self.D1 = model.centering_and_LDA.in_features, model.centering_and_LDA.out_features
        n = sum(q.numel() for q in self.params) + len(self.thetas)
        self.m = torch.zeros(n, device=self.dev)
        self.v = torch.zeros(n, device=self.dev)
        self.step_count = torch.zeros(2, device=self.dev)
        self.batch_size = batch_size
        self.use_graph = bool(graph) and batch_size is not None
        self._graph = self._loss = self._graph_rows = self._loss_rows = self._graph_table = None
        self.i1 = self.i2 = None
        # data parallel (neuralplda_amd.dist.make_data_parallel): the fp64 loss sums, then the folded fp64 gradient
        self.reduce_sums = getattr(model, "_reduce_sums", None)
        self.reduce_flat = model.__dict__.get("_reduce_sums64")
        if self.use_graph:
            self.x1 = torch.zeros(batch_size, D0, device=self.dev)
            self.x2 = torch.zeros(batch_size, D0, device=self.dev)
            self.t = torch.zeros(batch_size, device=self.dev)
            self.t[::2] = 1

    launches_per_step = None

    def _image(self, force=False):
        """The quadratic-form image the recipe step scores with: built once (two launches), then kept current by the step's
        own update kernel; re-packed IN PLACE (its address is baked into the captured graph) when anything else has touched
        the parameters (version counters)."""
        mdl = self.model
        prm = [mdl.centering_and_LDA.weight, mdl.centering_and_LDA.bias] + self.params[:2]
        key = tuple((q.data_ptr(), q._version) for q in prm)
        img = getattr(self, "_img", None)
        if img is None or ((force or key != self._img_key) and not torch.cuda.is_current_stream_capturing()):
            self._img = self._ops.dplda_pack(*[q.detach() for q in prm], out=img)
            self._img_key = key
        return self._img

    def _pre_replay(self, force=False):
        if getattr(self, "_img", None) is not None:
            self._image(force=force)

    def _touched(self):
        super(FusedDPldaStep, self)._touched()
        if getattr(self, "_img", None) is not None:  # (the step's own launch has refreshed the image with the new values)
            mdl = self.model
            prm = [mdl.centering_and_LDA.weight, mdl.centering_and_LDA.bias] + self.params[:2]
            self._img_key = tuple((q.data_ptr(), q._version) for q in prm)

    def _recipe_step(self, x1, x2, t):
        """xvector_DPlda_pytorch.py:35-43 with the LDA frozen, on one rank: THREE launches up to 4096 pairs — LDA + normalise +
        quadratic-form score (paired rows kept) | weighted moments, their weights dL/ds_i formed inline, + the loss block (loss,
        dL/dtheta) | gradient fold + Adam + parameter and image stores (nplda_dplda_update_loss_f32) — four above that (the loss
        as a launch of its own).  Same arithmetic as the separate calls below (loss, fold and Adam are the same device
        functions)."""
        ops = self._ops
        wlr, blr = (q.detach() for q in self.params[:2])
        s, paired = ops._gb_call(x1, x2, self._image(), True, True)
        ths = [th.detach() for th in self.thetas]
        lths = ths if self.kind == ops.LOSS_SOFTCDET else [self._zero]
        B = x1.shape[0]
        ws0 = self.__dict__.get("_mws_by_B", {}).get(B)
        # THREE launches where the batch fits the one-block loss (<= 4096 pairs): the loss rides in the moments launch
        fused = None
        if os.environ.get("NPLDA_DPLDA_LOSS_LAUNCH", "0") != "1":
            fused = ops.dplda_update_loss(paired, s, t, lths, self.betas_loss, self.alpha, self.kind, wlr, blr, self.m, self.v,
                                          self.step_count, self.lr, self.betas[0], self.betas[1], self.eps, self.wd, thetas=ths,
                                          image=self._img, ws=ws0, loss_sum=self._acc())
        if fused is not None:
            loss, _, self._mws = fused
            self.launches_per_step = ("3 in one graph replay: LDA + normalise + quadratic-form score | weighted moments with dL/ds "
                                      "formed inline + the loss block | gradient fold + Adam + parameter and image stores")
        else:
            loss, g, dth, _ = ops.loss_fwd_bwd(s, t, lths, self.betas_loss, self.alpha, self.kind)
            self._mws = ops.dplda_update(paired, g, wlr, blr, self.m, self.v, self.step_count, self.lr, self.betas[0], self.betas[1],
                                         self.eps, self.wd, thetas=ths, dtheta=dth if ths else None, image=self._img, ws=ws0,
                                         loss=loss, loss_sum=self._acc())
            self.launches_per_step = ("4 in one graph replay: LDA + normalise + quadratic-form score | loss + dL/ds | weighted "
                                      "moments | gradient fold + Adam + parameter and image stores")
        self.__dict__.setdefault("_mws_by_B", {})[B] = self._mws
        self._acc_in_kernel = True  # (the update launch adds the step's loss to the running sum: _account leaves it alone)
        return loss

    def _eager(self, x1, x2, t):
        ops, mdl = self._ops, self.model
        with torch.no_grad():
            if (not (self.train_lda or self.want_dx) and self.reduce_sums is None and self.reduce_flat is None
                    and x1.shape[0] > 0 and os.environ.get("NPLDA_DPLDA_SEPARATE", "0") != "1"):
                return self._recipe_step(x1, x2, t)
            self._acc_in_kernel = False
            wlr, blr = (q.detach() for q in self.params[:2])
            W1, b1 = mdl.centering_and_LDA.weight.detach(), mdl.centering_and_LDA.bias.detach()
            packed = ops.dplda_pack(W1, b1, wlr, blr)
            input_side = self.train_lda or self.want_dx
            if input_side:
                s, paired, rn = ops._gb_call(x1, x2, packed, True, True, want_rn=True)
            else:
                s, paired = ops._gb_call(x1, x2, packed, True, True)
            ths = [th.detach() for th in self.thetas]
            lths = ths if self.kind == ops.LOSS_SOFTCDET else [self._zero]
            if self.reduce_sums is None:
                loss, g, dth, _ = ops.loss_fwd_bwd(s, t, lths, self.betas_loss, self.alpha, self.kind)
            else:
                sums = self.reduce_sums(ops.loss_sums(s, t, lths, self.alpha, self.kind))
                loss, g, dth = ops.loss_finish(s, t, lths, self.betas_loss, self.alpha, self.kind, sums)
            if self.reduce_flat is None:
                dw, db = ops.dplda_grad(paired, g, self.D1)  # moments + fold, one call (same bits)
            else:
                dw, db = ops.dplda_fold_grad(*ops.weighted_moments(paired, g), self.D1, reduce=self.reduce_flat)
            tensors, grads = [wlr, blr], [dw.contiguous(), db.contiguous()]
            dx1 = dx2 = None
            if input_side:
                qimg, v = ops.dplda_quadform_image(wlr, self.D1)
                dpaired = ops.rows_matmul(paired, qimg, bias=v, rowscale=g)
                dW1, db1, dx1, dx2 = ops.lda_backward(x1, x2, paired, rn, dpaired, W1, want_w=self.train_lda,
                                                      want_dx=self.want_dx)
                if self.train_lda:
                    if self.reduce_flat is not None:
                        both = self.reduce_flat(torch.cat([dW1.reshape(-1), db1]))
                        dW1, db1 = both[:dW1.numel()].view_as(dW1), both[dW1.numel():]
                    tensors += [W1, b1]
                    grads += [dW1.contiguous(), db1.contiguous()]
            self._adam(tensors + ths, grads + [dth[k:k + 1] for k in range(len(ths))])
        return (loss, dx1, dx2) if self.want_dx else loss


def train_gaussian_backend(nc, model, train_loader, mega_xvec_dict, num_to_id_dict, device=None):
    """xvector_GaussianBackend_pytorch.py:30-56 (`train`): one pass over the loader accumulating per-class counts,
    sums and second moments of the paired rows, then the closed-form means / inverse covariances.  The x-vectors are
    gathered on the device (the reference's script feeds the index batches to forward_getpaired, :41 — a bug; the
    x-vector batches it built one line above are what is meant)."""
    device = torch.device(device or next(model.parameters()).device)
    model.eval()
    stats = None
    with torch.no_grad():
        if (isinstance(train_loader, TrialLoader) and device.type == "cuda"
                and isinstance(train_loader.dataset, TrialIndexDataset) and train_loader.num_workers == 0
                and not train_loader.drop_last):
            # device-resident pass (see train()): the same batches, gathered from the resident table
            from . import ops
            table, row_map = _device_table(mega_xvec_dict, num_to_id_dict, device)
            for rows1, rows2, target in train_loader.device_batches(device, row_map):
                stats = model.accumulate_statistics(ops.gather_rows(table, rows1), ops.gather_rows(table, rows2),
                                                    target, stats)
        else:
            for data1, data2, target in train_loader:
                x1, x2 = load_xvec_trials_from_numbatch(mega_xvec_dict, num_to_id_dict, data1, data2, device)
                stats = model.accumulate_statistics(x1, x2, target.to(device), stats)
    if stats is None:
        raise ValueError("empty training loader")
    return model.fit_statistics(stats)


def make_optimizer(model, lr, weight_decay=1e-5, capturable=False):
    """Adam as the reference configures it (xvector_NeuralPlda_pytorch.py:139)."""
    return optim.Adam(model.parameters(), lr=lr, weight_decay=weight_decay, capturable=capturable)


def main_kaldiplda(configfile='conf/voices_config.cfg', use_graph=True):
    """xvector_NeuralPlda_pytorch.py:88-181."""
    timestamp = int(datetime.timestamp(datetime.now()))
    print(timestamp)
    os.makedirs('logs', exist_ok=True)
    os.makedirs('models', exist_ok=True)
    os.makedirs('scores', exist_ok=True)
    logging.basicConfig(filename='logs/kaldiplda_{}.log'.format(timestamp), filemode='a',
                        format='%(levelname)s: %(message)s', datefmt='%H:%M:%S', level=logging.DEBUG)
    nc = NpldaConf(configfile)
    torch.manual_seed(nc.seed)
    np.random.seed(nc.seed)
    random.seed(nc.seed)
    logging.info("Started at {}\n\n {} \n\n".format(datetime.now(), open(configfile).read()))
    use_cuda = torch.cuda.is_available()
    if not use_cuda:
        raise RuntimeError("neuralplda_amd needs a HIP device")
    device = torch.device(nc.device if str(nc.device).startswith("cuda") else "cuda")

    mega_xvec_dict = pickle.load(open(nc.mega_xvector_pkl, 'rb'))
    num_to_id_dict = {i: j for i, j in enumerate(list(mega_xvec_dict))}
    id_to_num_dict = {v: k for k, v in num_to_id_dict.items()}
    train_loader = combine_trials_and_get_loader(nc.training_data_trials_list, id_to_num_dict,
                                                 subsample_factors=nc.train_subsample_factors,
                                                 batch_size=nc.batch_size)
    valid_loaders = get_trials_loaders_dict(nc.validation_trials_list, id_to_num_dict,
                                            subsample_factors=nc.valid_subsample_factors,
                                            batch_size=5 * nc.batch_size)
    model = NeuralPlda(nc).to(device)
    if nc.initialization == 'kaldi':
        model.LoadPldaParamsFromKaldi(nc.meanvec, nc.transformmat, nc.kaldiplda)
    lr = nc.lr
    optimizer = make_optimizer(model, lr)
    step_fn = FusedTrainStep(model, lr, weight_decay=1e-5, batch_size=nc.batch_size, graph=use_graph)

    print("Initializing the thresholds... Training and validation loss printed now is not meaningful.")
    validate(nc, model, device, mega_xvec_dict, num_to_id_dict, valid_loaders[nc.heldout_set_for_th_init],
             update_thresholds=True)
    # epoch-0 validation of every set; the held-out minC opens the LR-decay history (xvector_NeuralPlda_pytorch.py:146-153)
    all_losses = []
    val_losses = {}
    for val_set, loader in valid_loaders.items():
        print("Validating on {}".format(val_set))
        val_losses[val_set], _ = validate(nc, model, device, mega_xvec_dict, num_to_id_dict, loader)
    all_losses.append(float(val_losses[nc.heldout_set_for_lr_decay]))
    for epoch in range(1, nc.n_epochs + 1):
        train(nc, model, device, train_loader, mega_xvec_dict, num_to_id_dict, optimizer, epoch, valid_loaders,
              step_fn=step_fn)
        val_losses = {}
        for val_set, loader in valid_loaders.items():
            print("Validating on {}".format(val_set))
            val_losses[val_set], _ = validate(nc, model, device, mega_xvec_dict, num_to_id_dict, loader)
        all_losses.append(float(val_losses[nc.heldout_set_for_lr_decay]))
        model.SaveModel("models/NPLDA_{}_{}.pt".format(epoch, timestamp))
        for trial_file in nc.test_trials_list:
            print("Generating scores for Epoch {} with trial file {}".format(epoch, trial_file))
            # the reference's file name, timestamp included (xvector_NeuralPlda_pytorch.py:172)
            nc.generate_scorefile("scores/kaldipldanet_epoch{}_{}_{}.txt".format(
                epoch, os.path.splitext(os.path.basename(trial_file))[0], timestamp), trial_file, mega_xvec_dict,
                model, device, 5 * nc.batch_size)
        # LR halving: three strictly increasing held-out minC values, optimiser re-created (:174-179)
        if len(all_losses) >= 3 and all_losses[-1] > all_losses[-2] > all_losses[-3]:
            lr = lr / 2
            print("REDUCING LEARNING RATE to {} since loss trend looks like {}".format(lr, all_losses[-3:]))
            logging.info("REDUCING LEARNING RATE to {} since loss trend looks like {}".format(lr, all_losses[-3:]))
            optimizer = make_optimizer(model, lr)  # moments reset, as the reference re-creates Adam (:177)
            step_fn = FusedTrainStep(model, lr, weight_decay=1e-5, batch_size=nc.batch_size, graph=use_graph)
    return model


def main_dplda(configfile='conf/voices_config.cfg', use_graph=True):
    """xvector_DPlda_pytorch.py:88-189: the same driver around DPlda — Kaldi LDA loaded and frozen (:131-147), Adam
    over logistic_regres + thresholds, eager steps (forward = fused quadratic form, backward = weighted moments)."""
    from .models import DPlda
    timestamp = int(datetime.timestamp(datetime.now()))
    for d in ('logs', 'models', 'scores'):
        os.makedirs(d, exist_ok=True)
    logging.basicConfig(filename='logs/dplda_{}.log'.format(timestamp), filemode='a',
                        format='%(levelname)s: %(message)s', datefmt='%H:%M:%S', level=logging.DEBUG)
    nc = NpldaConf(configfile)
    torch.manual_seed(nc.seed)
    np.random.seed(nc.seed)
    random.seed(nc.seed)
    if not torch.cuda.is_available():
        raise RuntimeError("neuralplda_amd needs a HIP device")
    device = torch.device(nc.device if str(nc.device).startswith("cuda") else "cuda")
    mega_xvec_dict = pickle.load(open(nc.mega_xvector_pkl, 'rb'))
    num_to_id_dict = {i: j for i, j in enumerate(list(mega_xvec_dict))}
    id_to_num_dict = {v: k for k, v in num_to_id_dict.items()}
    train_loader = combine_trials_and_get_loader(nc.training_data_trials_list, id_to_num_dict,
                                                 subsample_factors=nc.train_subsample_factors,
                                                 batch_size=nc.batch_size)
    valid_loaders = get_trials_loaders_dict(nc.validation_trials_list, id_to_num_dict,
                                            subsample_factors=nc.valid_subsample_factors,
                                            batch_size=5 * nc.batch_size)
    model = DPlda(nc).to(device)
    if nc.initialization == 'kaldi':
        model.LoadParamsFromKaldi(nc.meanvec, nc.transformmat)
    updatable = []
    for name, prm in model.named_parameters():
        if 'centering_and_LDA' in name:
            prm.requires_grad = False
        else:
            updatable.append(prm)
    lr = nc.lr
    optimizer = optim.Adam(updatable, lr=lr, weight_decay=1e-5)
    on_gpu = torch.device(device).type == "cuda"
    step_fn = (FusedDPldaStep(model, lr, weight_decay=1e-5, batch_size=nc.batch_size, graph=use_graph)
               if on_gpu else None)
    validate(nc, model, device, mega_xvec_dict, num_to_id_dict, valid_loaders[nc.heldout_set_for_th_init],
             update_thresholds=True)
    all_losses = []
    val_losses = {}
    for val_set, loader in valid_loaders.items():  # epoch 0, as xvector_DPlda_pytorch.py:153-160
        val_losses[val_set], _ = validate(nc, model, device, mega_xvec_dict, num_to_id_dict, loader)
    all_losses.append(float(val_losses[nc.heldout_set_for_lr_decay]))
    for epoch in range(1, nc.n_epochs + 1):
        train(nc, model, device, train_loader, mega_xvec_dict, num_to_id_dict, optimizer, epoch, step_fn=step_fn)
        val_losses = {}
        for val_set, loader in valid_loaders.items():
            val_losses[val_set], _ = validate(nc, model, device, mega_xvec_dict, num_to_id_dict, loader)
        all_losses.append(float(val_losses[nc.heldout_set_for_lr_decay]))
        model.SaveModel("models/NPLDA_{}_{}.pt".format(epoch, timestamp))
        for trial_file in nc.test_trials_list:
            nc.generate_scorefile("scores/kaldipldanet_epoch{}_{}_{}.txt".format(
                epoch, os.path.splitext(os.path.basename(trial_file))[0], timestamp), trial_file, mega_xvec_dict, model,
                device, 5 * nc.batch_size)
        if len(all_losses) >= 3 and all_losses[-1] > all_losses[-2] > all_losses[-3]:
            lr = lr / 2
            logging.info("REDUCING LEARNING RATE to {} since loss trend looks like {}".format(lr, all_losses[-3:]))
            optimizer = optim.Adam(updatable, lr=lr, weight_decay=1e-5)  # moments reset, as the reference re-creates Adam
            if on_gpu:
                step_fn = FusedDPldaStep(model, lr, weight_decay=1e-5, batch_size=nc.batch_size, graph=use_graph)
    return model


if __name__ == '__main__':
    import sys
    main_kaldiplda(*(sys.argv[1:2]))
