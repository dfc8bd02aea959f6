"""Host-side mirror of the reference's model classes for the hot path (utils/models.py).

`NeuralPlda` and `GaussianBackend` keep the reference's constructor, method names, state-dict keys
and instance attributes (utils/models.py:348-461, :571-665) so the training / scoring scripts
(xvector_NeuralPlda_pytorch.py, xvector_generate_scores.py) run unchanged — but every piece of
arithmetic on the path is a hand-written HIP kernel reached through the C ABI
(include/nplda_hip.h).  There is NO CPU implementation in this module: CPU tensors are staged
through the HIP device (the score file generators of the reference hand CPU tensors to forward(),
utils/scorefile_generator.py:25-34), and if no HIP device / library is present every compute
method raises.
"""
import pickle

import numpy as np
import torch
import torch.nn as nn

from . import _lib, kaldi_format, ops

__all__ = ["NeuralPlda", "GaussianBackend", "arr2val"]


def arr2val(x, retidx):
    """utils/models.py:23-27 (kept for scripts that import it)."""
    if x.size()[0] > 0:
        return x[retidx].cpu().item()
    return 1.


def _compute_device(*tensors):
    """Device the kernels run on: the first HIP tensor's device, else the current HIP device."""
    for t in tensors:
        if isinstance(t, torch.Tensor) and t.is_cuda:
            return t.device
    if not torch.cuda.is_available():
        raise _lib.NpldaHipError(
            "neuralplda_amd needs a HIP device: the NPLDA hot path has no CPU implementation "
            "(the CPU oracle under oracle/ is test infrastructure only)")
    return torch.device("cuda", torch.cuda.current_device())


def _to_dev(t, dev):
    t = t.detach() if t.requires_grad else t
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.device == dev else t.to(dev, non_blocking=True)


def _loss_kind(name):
    if isinstance(name, str):
        if name.lower() == "softcdet":  # the shipped configs spell it both 'SoftCdet' and 'softCdet'
            return ops.LOSS_SOFTCDET
        if name.lower() in ("crossentropy", "bce"):
            return ops.LOSS_BCE
    raise ValueError(f"unknown loss {name!r}: expected 'SoftCdet' or 'crossentropy' (utils/models.py:395-399)")


# ---------------------------------------------------------------------------------------------------
# autograd bridges
# ---------------------------------------------------------------------------------------------------

def _bump(*tensors):
    """Mark tensors that a raw kernel (or a `.data` write) has just modified in place as changed, so that the
    packed-image cache below and autograd's saved-tensor checks both notice."""
    for t in tensors:
        torch.autograd.graph.increment_version(t)


def _packed_for(cache, prm, precision):
    """The MFMA-fragment image of `prm` (six device tensors), rebuilt only when a parameter changed: keyed on each
    tensor's storage address and autograd version counter (optimizer steps, load_state_dict and every in-place op bump
    it; the fused steps and the Kaldi loaders, which write through raw kernels / `.data`, bump it explicitly).  A
    `forward()` on a 10 240-pair scoring chunk otherwise spends as long re-packing the 0.5 MB image as scoring."""
    if cache is None:
        return ops.pack_params(*prm, precision=precision)
    key = (precision,) + tuple((t.data_ptr(), t._version, t.device) for t in prm)
    hit = cache.get("key")
    old = cache.get("packed")
    if old is not None and old.buf.is_inference() and not torch.is_inference_mode_enabled():
        # an image built under torch.inference_mode() cannot be saved for a backward (and has no version counter): outside
        # inference mode it is built anew
        hit = old = None
        cache.pop("where", None)
    if hit != key:
        # (shape and stride belong to "the same six tensors": a parameter replaced by another tensor that lands on the
        # same address with another shape must not be repacked with the dims recorded at the first pack)
        where = (precision,) + tuple((t.data_ptr(), t.device, tuple(t.shape), t.stride()) for t in prm)
        if old is not None and cache.get("where") == where and cache.get("contig"):
            # the same six tensors with new values (an optimiser step between two forwards — every iteration of the
            # training loop): refresh the image IN PLACE with one raw launch.  The buffer's version is bumped: a graph that
            # saved it for its backward now fails autograd's saved-tensor check, exactly as it fails on the parameters
            # themselves, which the optimiser also changed in place.
            ops.repack_params(old, key[1:])
            if not old.buf.is_inference():  # (an image first built under torch.inference_mode() has no version counter)
                _bump(old.buf)
        else:
            cache["packed"] = ops.pack_params(*prm, precision=precision)
            cache["where"] = where
            cache["contig"] = all(t.is_contiguous() and t.dtype == torch.float32 and t.is_cuda for t in prm)
        cache["key"] = key
    return cache["packed"]


def _back(t, like):
    """Gradient `t` on the device / dtype of the tensor it belongs to."""
    if t is None:
        return None
    if t.dtype != like.dtype:
        t = t.to(like.dtype)
    return t if t.device == like.device else t.to(like.device)


class _PairScoreFn(torch.autograd.Function):
    """s = NeuralPlda.forward(x1, x2) with the hand-derived backward: parameter gradients from
    nplda_backward, and — when the x-vectors themselves carry a graph (an extractor trained jointly with the head, the
    reference's Etdnn_Xvec_NeuralPlda, utils/models.py:251-268) — dL/dx1, dL/dx2 = du . W1 from the same launch set."""

    @staticmethod
    def forward(ctx, x1, x2, opts, W1, b1, W2, b2, P_sqrt, Q):
        reduce_flat, precision, cache, grad_on = opts
        dev = _compute_device(x1, W1)
        params = (W1, b1, W2, b2, P_sqrt, Q)
        on_dev = all(t.device == dev and t.dtype == torch.float32 for t in params)
        prm = [_to_dev(t, dev) for t in params]
        # Function.forward runs with grad mode off, and ctx.needs_input_grad only says which inputs REQUIRE grad — it is
        # True for the parameters under torch.no_grad() too: `grad_on` is the caller's grad mode (validate() and the score
        # generators run under no_grad: they must take the scoring kernels, not the activation-saving training forward)
        need_x = grad_on and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1])
        need = grad_on and (need_x or any(ctx.needs_input_grad[3:]))
        # training always runs the exact-fp32 kernels; `precision` only selects the inference kernel
        packed = _packed_for(cache if on_dev else None, prm, "fp32" if need else precision)
        X1, X2 = _to_dev(x1, dev), _to_dev(x2, dev)
        ctx.need, ctx.need_x = need, need_x
        ctx.reduce_flat = reduce_flat
        if need:
            s, (X1, X2, ld, y, z, rn) = ops.forward_train(X1, X2, packed)
            ctx.save_for_backward(X1, X2, y, z, rn, packed.buf, prm[4], x1, x2, *params)
            ctx.ld, ctx.dims = ld, (packed.D0, packed.D1, packed.D2, packed.ldz)
        else:
            s = ops.score_pairs(X1, X2, packed)
        return s if s.device == x1.device else s.to(x1.device)

    @staticmethod
    def backward(ctx, gs):
        if not ctx.need:
            return (None,) * 9
        X1, X2, y, z, rn, buf, ps, x1, x2 = ctx.saved_tensors[:9]
        params = ctx.saved_tensors[9:]
        D0, D1, D2, ldz = ctx.dims
        packed = ops.PackedParams(buf, D0, D1, D2, ldz)
        out = ops.backward((X1, X2, ctx.ld, y, z, rn), _to_dev(gs, buf.device), packed, ps, want_dx=ctx.need_x)
        flat, dx1, dx2 = out if ctx.need_x else (out, None, None)
        if ctx.reduce_flat is not None:  # data parallel: ONE sum-all-reduce of the flat gradient
            flat = ctx.reduce_flat(flat)
        grads = [_back(g, t) if need else None
                 for g, t, need in zip(ops.split_flat_grad(flat, D0, D1, D2), params, ctx.needs_input_grad[3:])]
        return (_back(dx1, x1) if ctx.needs_input_grad[0] else None,
                _back(dx2, x2) if ctx.needs_input_grad[1] else None, None) + tuple(grads)


class _EmbedFn(torch.autograd.Function):
    """z = extract_plda_embeddings(x) (utils/models.py:366-370), differentiable like the reference's: gradients of
    W1, b1, W2, b2 and of x from nplda_embed_backward."""

    @staticmethod
    def forward(ctx, x, opts, W1, b1, W2, b2, P_sqrt, Q):
        reduce_flat, cache, grad_on = opts
        dev = _compute_device(x, W1)
        params = (W1, b1, W2, b2, P_sqrt, Q)
        on_dev = all(t.device == dev and t.dtype == torch.float32 for t in params)
        packed = _packed_for(cache if on_dev else None, [_to_dev(t, dev) for t in params], "fp32")
        need = grad_on and (ctx.needs_input_grad[0] or any(ctx.needs_input_grad[2:6]))
        ctx.need, ctx.reduce_flat = need, reduce_flat
        X = _to_dev(x, dev)
        if need:
            z, (X, ld, y, rn) = ops.embed_train(X, packed)
            ctx.save_for_backward(X, y, rn, packed.buf, x, W1, b1, W2, b2)
            ctx.ld, ctx.dims = ld, (packed.D0, packed.D1, packed.D2, packed.ldz)
        else:
            z, _ = ops.embed(X, packed, want_q=False)
        z = z[:, :packed.D2]
        return z if z.device == x.device else z.to(x.device)

    @staticmethod
    def backward(ctx, gz):
        if not ctx.need:
            return (None,) * 8
        X, y, rn, buf, x = ctx.saved_tensors[:5]
        params = ctx.saved_tensors[5:]
        D0, D1, D2, ldz = ctx.dims
        packed = ops.PackedParams(buf, D0, D1, D2, ldz)
        flat, dx = ops.embed_backward((X, ctx.ld, y, rn), _to_dev(gz, buf.device), packed,
                                      want_dx=ctx.needs_input_grad[0])
        if ctx.reduce_flat is not None:
            flat = ctx.reduce_flat(flat)
        grads = [_back(g, t) if need else None
                 for g, t, need in zip(ops.split_flat_grad(flat, D0, D1, D2)[:4], params, ctx.needs_input_grad[2:6])]
        return (_back(dx, x) if ctx.needs_input_grad[0] else None, None) + tuple(grads) + (None, None)


class _EmbScoreFn(torch.autograd.Function):
    """s = forward_from_plda_embeddings(z1, z2) (utils/models.py:372-376) with its backward (dz1, dz2, dP_sqrt, dQ)."""

    @staticmethod
    def forward(ctx, z1, z2, opts, P_sqrt, Q):
        reduce_flat, grad_on = opts
        dev = _compute_device(z1, Q)
        Z1, Z2, ps, q = _to_dev(z1, dev), _to_dev(z2, dev), _to_dev(P_sqrt, dev), _to_dev(Q, dev)
        s = ops.score_embeddings(Z1, Z2, ps, q)
        ctx.need = grad_on and any(ctx.needs_input_grad)
        ctx.reduce_flat = reduce_flat
        if ctx.need:
            ctx.save_for_backward(Z1, Z2, ps, q, z1, z2, P_sqrt, Q)
        return s if s.device == z1.device else s.to(z1.device)

    @staticmethod
    def backward(ctx, gs):
        if not ctx.need:
            return (None,) * 5
        Z1, Z2, ps, q, z1, z2, P_sqrt, Q = ctx.saved_tensors
        need = ctx.needs_input_grad
        dz1, dz2, dP, dQ = ops.score_embeddings_bwd(Z1, Z2, ps, q, _to_dev(gs, Z1.device), need[0], need[1])
        if ctx.reduce_flat is not None and (need[3] or need[4]):
            both = ctx.reduce_flat(torch.cat([dP, dQ]))
            dP, dQ = both[:dP.numel()], both[dP.numel():]
        return (_back(dz1, z1) if need[0] else None, _back(dz2, z2) if need[1] else None, None,
                _back(dP, P_sqrt) if need[3] else None, _back(dQ, Q) if need[4] else None)


class _DPldaScoreFn(torch.autograd.Function):
    """DPlda.forward (utils/models.py:492-495): fused LDA + normalise + quadratic form, saving the paired rows [y1, y2].
    Backward: the linear unit's gradient is one weighted-moments pass (sum_k g_k x x^T) folded into d wlr, d bias — what
    the recipe trains (xvector_DPlda_pytorch.py:140-147 freezes the LDA).  When the LDA or the x-vectors DO carry a graph
    (a joint fine-tune), dL/d[y1; y2] = g ((M + M^T) x + v) is one resident-matrix GEMM on the paired rows, followed by
    the F.normalize backward and the LDA wgrad / dgrad GEMMs (csrc/nplda_matmul.hip)."""

    @staticmethod
    def forward(ctx, x1, x2, opts, W1, b1, wlr, blr):
        reduce_sums64, grad_on = opts
        dev = _compute_device(x1, W1)
        D1 = W1.shape[0]
        W1d, b1d = _to_dev(W1, dev), _to_dev(b1, dev)
        packed = ops.dplda_pack(W1d, b1d, _to_dev(wlr, dev), _to_dev(blr, dev))
        need_unit = grad_on and (ctx.needs_input_grad[5] or ctx.needs_input_grad[6])
        need_lda = grad_on and any(ctx.needs_input_grad[i] for i in (0, 1, 3, 4))
        X1, X2 = _to_dev(x1, dev), _to_dev(x2, dev)
        ctx.need_unit, ctx.need_lda, ctx.D1 = need_unit, need_lda, D1
        ctx.reduce = reduce_sums64
        if need_lda:
            if D1 % 2:
                raise ValueError("a backward through DPlda's LDA needs an even layer1_LDA_dim (16-byte paired rows)")
            s, paired, rn = ops._gb_call(X1, X2, packed, True, True, want_rn=True)
            ctx.save_for_backward(paired, rn, X1, X2, W1d, _to_dev(wlr, dev), x1, x2, W1, b1, wlr, blr)
        elif need_unit:
            s, paired = ops._gb_call(X1, X2, packed, True, True)
            ctx.save_for_backward(paired, wlr, blr)
        else:
            s, _ = ops._gb_call(X1, X2, packed, True, False)
        return s if s.device == x1.device else s.to(x1.device)

    @staticmethod
    def backward(ctx, gs):
        if not (ctx.need_unit or ctx.need_lda):
            return (None,) * 7
        saved = ctx.saved_tensors
        paired = saved[0]
        g = gs.to(paired.device, torch.float32).contiguous()
        D1 = ctx.D1
        dw = db = dW1 = db1 = dx1 = dx2 = None
        if ctx.need_lda:
            _, rn, X1, X2, W1d, wlrd, x1, x2, W1, b1, wlr, blr = saved
        else:
            _, wlr, blr = saved
        if ctx.need_unit:
            dw, db = ops.dplda_fold_grad(*ops.weighted_moments(paired, g), D1, reduce=ctx.reduce)
            dw, db = _back(dw, wlr), _back(db, blr)
        if ctx.need_lda:
            M, v, _ = ops.dplda_quadform(wlrd, None, D1)
            dpaired = ops.rows_matmul(paired, ops.pack_matrix(M, mode=2), bias=v, rowscale=g)
            need = ctx.needs_input_grad
            dW1, db1, dx1, dx2 = ops.lda_backward(X1, X2, paired, rn, dpaired, W1d, want_w=need[3] or need[4],
                                                  want_dx=need[0] or need[1])
            if ctx.reduce is not None and dW1 is not None:
                both = ctx.reduce(torch.cat([dW1.reshape(-1), db1]))
                dW1, db1 = both[:dW1.numel()].view_as(dW1), both[dW1.numel():]
            dW1 = _back(dW1, W1) if need[3] else None
            db1 = _back(db1, b1) if need[4] else None
            dx1 = _back(dx1, x1) if need[0] else None
            dx2 = _back(dx2, x2) if need[1] else None
        return dx1, dx2, None, dW1, db1, dw, db


class _LossFn(torch.autograd.Function):
    """SoftCdet / BCE with the fused forward+backward kernels.  `reduce_sums` (optional callable)
    all-reduces the fp64 batch sums across data-parallel ranks before the gradient is formed."""

    @staticmethod
    def forward(ctx, output, target, kind, alpha, betas, opts, *thetas):
        reduce_sums, grad_on = opts
        dev = _compute_device(output, target)
        s, t = _to_dev(output, dev), _to_dev(target, dev)
        ths = [_to_dev(th, dev) for th in thetas]
        need = grad_on and (ctx.needs_input_grad[0] or any(ctx.needs_input_grad[6:]))
        if need and reduce_sums is None and kind != ops.LOSS_HARD_CDET:
            # one rank, training: both loss passes in ONE call (one launch up to 4096 pairs; same bits as the two below)
            loss, g, dth, _, joint = ops.loss_fwd_bwd(s, t, ths, betas, alpha, kind, want_joint=True)
        else:
            joint = None
            sums = ops.loss_sums(s, t, ths, alpha, kind)
            if reduce_sums is not None:
                sums = reduce_sums(sums)
            loss, g, dth = ops.loss_finish(s, t, ths, betas, alpha, kind, sums, want_grad=need)
        ctx.need = need
        ctx.nth = len(thetas)
        if need:
            if joint is not None:
                ctx.save_for_backward(joint, output, *thetas)
            else:
                ctx.save_for_backward(g, dth, output, *thetas)
            ctx.joint, ctx.B = joint is not None, s.shape[0]
        return loss if loss.device == output.device else loss.to(output.device)

    @staticmethod
    def backward(ctx, gl):
        if not ctx.need:
            return (None,) * (6 + ctx.nth)
        if ctx.joint:  # [g | dtheta] in one buffer: one product
            joint, output = ctx.saved_tensors[:2]
            thetas = ctx.saved_tensors[2:]
            if gl.device != joint.device:
                gl = gl.to(joint.device)
            gg, dthg = ops.loss_joint_views(joint * gl, ctx.B, ctx.nth)
        else:
            g, dth, output = ctx.saved_tensors[:3]
            thetas = ctx.saved_tensors[3:]
            gl = gl.to(g.device)
            gg, dthg = g * gl, dth * gl
        dths = [_back(dthg[k:k + 1], th) if need else None
                for k, (th, need) in enumerate(zip(thetas, ctx.needs_input_grad[6:]))]
        return (_back(gg, output) if ctx.needs_input_grad[0] else None, None, None, None, None, None) + tuple(dths)


# ---------------------------------------------------------------------------------------------------
# NeuralPlda
# ---------------------------------------------------------------------------------------------------

class NeuralPlda(nn.Module):
    """Drop-in for utils/models.py:348-461.  `nc` is any object with the NpldaConf fields the reference
    constructor reads (xvector_dim, layer1_LDA_dim, layer2_PLDA_spkfactor_dim, beta, alpha, device, loss)."""

    def __init__(self, nc):
        super(NeuralPlda, self).__init__()
        self.centering_and_LDA = nn.Linear(nc.xvector_dim, nc.layer1_LDA_dim)  # Centering, wccn
        self.centering_and_wccn_plda = nn.Linear(nc.layer1_LDA_dim, nc.layer2_PLDA_spkfactor_dim)
        self.P_sqrt = nn.Parameter(torch.rand(nc.layer2_PLDA_spkfactor_dim, requires_grad=True))
        self.Q = nn.Parameter(torch.rand(nc.layer2_PLDA_spkfactor_dim, requires_grad=True))
        self.threshold = {}
        for beta in nc.beta:
            self.threshold[beta] = nn.Parameter(0 * torch.rand(1, requires_grad=True))
            self.register_parameter("Th{}".format(int(beta)), self.threshold[beta])
        self.threshold_Xent = nn.Parameter(0 * torch.rand(1, requires_grad=True))
        # the reference keeps alpha as a plain tensor attribute (does not follow .to()); only its value is used here
        self.alpha = torch.tensor(float(nc.alpha))
        self.beta = nc.beta
        self.dropout = nn.Dropout(p=0.5)  # defined and never used by the reference (utils/models.py:362)
        self.lossfn = nc.loss
        self._reduce_sums = None  # both set by neuralplda_amd.dist.make_data_parallel()
        self._reduce_flat = None
        self._pack_cache = {}  # packed parameter image, rebuilt when a parameter's version changes (_packed_for)
        # inference kernel: "fp32" (exact fp32 MFMA, default) or "bf16x3" (split-bf16, fp32-class accuracy, ~1.5x faster)
        self.scoring_precision = "fp32"

    # -- pickles written by the reference (class path utils.models.NeuralPlda) lack our private attributes
    def __setstate__(self, state):
        super(NeuralPlda, self).__setstate__(state)
        self.__dict__.setdefault("_reduce_sums", None)
        self.__dict__.setdefault("_reduce_flat", None)
        self.__dict__.setdefault("_dp_group", None)
        self.__dict__["_pack_cache"] = {}
        self.__dict__.setdefault("scoring_precision", "fp32")

    def __getstate__(self):
        base = getattr(super(NeuralPlda, self), "__getstate__", None)
        state = dict(base()) if base is not None else self.__dict__.copy()
        state["_reduce_sums"] = None  # process-group closures do not pickle
        state["_reduce_flat"] = None
        state["_pack_cache"] = {}     # device buffers of a cache do not belong in a model file
        state.pop("_reduce_sums64", None)
        if "_dp_group" in state:
            state["_dp_group"] = None  # a ProcessGroup does not pickle (dist.make_data_parallel(group=...))
        return state

    def _params(self):
        return (self.centering_and_LDA.weight, self.centering_and_LDA.bias, self.centering_and_wccn_plda.weight,
                self.centering_and_wccn_plda.bias, self.P_sqrt, self.Q)

    def invalidate_packed(self):
        """Drop the cached MFMA-fragment image of the parameters.  The cache (`_packed_for`) is keyed on the parameters'
        autograd version counters, which every in-place op, optimizer step and load_state_dict bumps — but a write
        through `p.data` (`p.data.copy_(w)`, an EMA swap) or by a foreign captured graph does not.  Call this after such
        a write; `train()` / `eval()` mode switches call it too, so the usual train -> eval -> score sequence is safe
        whatever touched the weights in between."""
        self.__dict__["_pack_cache"] = {}

    def train(self, mode=True):
        self.invalidate_packed()
        return super(NeuralPlda, self).train(mode)

    def extract_plda_embeddings(self, x):
        """utils/models.py:366-370 -> (B, D2)."""
        x = x.reshape(-1, self.centering_and_LDA.in_features) if x.dim() != 2 else x
        return _EmbedFn.apply(x, (self._reduce_flat, self.__dict__.get("_pack_cache"), torch.is_grad_enabled()), *self._params())

    def forward_from_plda_embeddings(self, x1, x2):
        """utils/models.py:372-376."""
        return _EmbScoreFn.apply(x1, x2, (self._reduce_flat, torch.is_grad_enabled()), self.P_sqrt, self.Q)

    def forward(self, x1, x2):
        """utils/models.py:378-382: (B, D0), (B, D0) -> (B,).  B == 0 returns an empty tensor (the
        reference crashes there, utils/scorefile_generator.py:29-33)."""
        D0 = self.centering_and_LDA.in_features
        if x1.numel() == 0 and x2.numel() == 0:
            x1, x2 = x1.reshape(0, D0), x2.reshape(0, D0)
        if not torch.is_grad_enabled():
            # inference (validate(), the score generators): straight to the scoring call — no autograd node to build; the
            # bridge below costs ~40 us of host time per call, more than the kernel below ~8 000 pairs
            prm = self._params()
            w = prm[0]
            if (w.is_cuda and x1.device == w.device and x2.device == w.device and x1.dtype == x2.dtype
                    and x1.dtype in (torch.float32, torch.bfloat16)  # (bf16 rows: scored without an fp32 copy, ops.score_pairs)
                    and all(t.dtype == torch.float32 for t in prm)):
                packed = _packed_for(self.__dict__.get("_pack_cache"), prm,
                                     getattr(self, "scoring_precision", "fp32"))
                return ops.score_pairs(x1, x2, packed)
        return _PairScoreFn.apply(x1, x2, (self._reduce_flat, getattr(self, "scoring_precision", "fp32"),
                                           self.__dict__.get("_pack_cache"), torch.is_grad_enabled()), *self._params())

    def forward_rows(self, table, rows1, rows2):
        """Inference on pairs named by rows of a resident x-vector table (validate()'s device-resident loop): forward(
        table[rows1], table[rows2]) without materialising the gathered batches.  No graph is built (call under no_grad)."""
        prm = self._params()
        packed = _packed_for(self.__dict__.get("_pack_cache"), prm, getattr(self, "scoring_precision", "fp32"))
        with torch.no_grad():
            return ops.score_pairs_rows(table, rows1, rows2, packed)

    def forward_distinct(self, table, urows, j1, j2):
        """Inference on trials that name few distinct utterances: forward(table[urows[j1]], table[urows[j2]]) with every
        distinct row embedded ONCE (extract_plda_embeddings, utils/models.py:366-370) and the trials scored from the
        embedding table by index (forward_from_plda_embeddings, :372-376) — 2 x 199 k FLOP per DISTINCT utterance plus
        1.2 KB of table reads per trial instead of 398 k FLOP per trial.  Same scores as forward() to the fp32 tolerance
        (the association of the score's feature sum differs).  No graph is built."""
        prm = self._params()
        with torch.no_grad():
            packed = _packed_for(self.__dict__.get("_pack_cache"), prm, "fp32")
            z, _ = ops.embed_rows(table, urows, packed)
            return ops.score_indexed(z, None, j1, j2, packed)  # (self terms from the rows: no scattered q reads)

    # -- losses ----------------------------------------------------------------------------------
    def _alpha(self):
        return float(self.alpha.item()) if isinstance(self.alpha, torch.Tensor) else float(self.alpha)

    def softcdet(self, output, target):
        """utils/models.py:384-388."""
        thetas = [self.threshold[b] for b in self.beta]
        return _LossFn.apply(output, target, ops.LOSS_SOFTCDET, self._alpha(), [float(b) for b in self.beta],
                             (self._reduce_sums, torch.is_grad_enabled()), *thetas)

    def crossentropy(self, output, target):
        """utils/models.py:390-393."""
        return _LossFn.apply(output, target, ops.LOSS_BCE, 0.0, [], (self._reduce_sums, torch.is_grad_enabled()), self.threshold_Xent)

    def loss(self, output, target):
        """utils/models.py:395-399; accepts the config spelling 'softCdet' as well (conf/voices_config.cfg:25
        makes the reference return None and crash)."""
        if _loss_kind(self.lossfn) == ops.LOSS_SOFTCDET:
            return self.softcdet(output, target)
        return self.crossentropy(output, target)

    def cdet(self, output, target):
        """utils/models.py:401-404: hard detection cost at the model thresholds (strict < / >)."""
        dev = _compute_device(output, target)
        s, t = _to_dev(output, dev), _to_dev(target, dev)
        ths = [_to_dev(self.threshold[b], dev) for b in self.beta]
        sums = ops.loss_sums(s, t, ths, 0.0, ops.LOSS_HARD_CDET)
        loss, _, _ = ops.loss_finish(s, t, ths, [float(b) for b in self.beta], 0.0, ops.LOSS_HARD_CDET, sums,
                                     want_grad=False)
        return loss if loss.device == output.device else loss.to(output.device)

    def minc(self, output, target, update_thresholds=False, showplots=False, exact=False):
        """utils/models.py:406-436.  Default: the reference's semantics bit for bit (thresholds at target
        scores only, `arr2val` count-1 / 1.0-when-empty quirks), evaluated by a device sort + binary
        searches instead of the O(N_tgt * N) Python loop.  exact=True: true minimum over all thresholds."""
        from . import metrics
        minc_avg, minc_threshold = metrics.minc(output, target, self.beta, reference_semantics=not exact)
        if update_thresholds:
            for beta in self.beta:
                self.state_dict()["Th{}".format(int(beta))].data.copy_(minc_threshold[beta])
                _bump(self.threshold[beta])
        return minc_avg, minc_threshold

    # -- initialisation / persistence ---------------------------------------------------------------
    def LoadPldaParamsFromKaldi(self, mean_vec_file, transform_mat_file, PldaFile):
        """utils/models.py:441-457, reading the Kaldi files natively (no Kaldi binaries needed)."""
        kaldi_format.fold_init(self, mean_vec_file, transform_mat_file, PldaFile)

    def SaveModel(self, filename):
        """utils/models.py:459-461: pickle of the whole module."""
        with open(filename, 'wb') as f:
            pickle.dump(self, f)


# ---------------------------------------------------------------------------------------------------
# DPlda (forward / scoring only here; the reference never trains it in any shipped recipe)
# ---------------------------------------------------------------------------------------------------

class DPlda(NeuralPlda):
    """Drop-in for utils/models.py:463-569: same constructor fields, parameter names (centering_and_LDA,
    logistic_regres, Th<beta>) and methods.  forward never forms the 2 D^2 + D outer-product features per pair:
    the linear unit over them is the quadratic form x^T M x + x^T v + c on x = [y1; y2], evaluated by the fused
    LDA + normalise + quadratic-form kernel (MODE_GB).  Losses / cdet / minc are shared with NeuralPlda (the
    reference's DPlda.crossentropy has no threshold: threshold_Xent is fixed at 0 here).  Scores carry no
    autograd graph w.r.t. the LDA: like the reference's recipe (xvector_DPlda_pytorch.py:140-147) only logistic_regres
    and the thresholds train; their gradient is a weighted-moments pass over the paired rows (ops.weighted_moments)."""

    def __init__(self, nc):
        nn.Module.__init__(self)
        self.centering_and_LDA = nn.Linear(nc.xvector_dim, nc.layer1_LDA_dim)  # Centering, wccn
        self.logistic_regres = nn.Linear(nc.layer1_LDA_dim * nc.layer1_LDA_dim * 2 + nc.layer1_LDA_dim, 1)
        self.threshold = {}
        for beta in nc.beta:
            self.threshold[beta] = nn.Parameter(0 * torch.rand(1, requires_grad=True))
            self.register_parameter("Th{}".format(int(beta)), self.threshold[beta])
        self.alpha = torch.tensor(float(nc.alpha))
        self.beta = nc.beta
        self.dropout = nn.Dropout(p=0.5)
        self.lossfn = nc.loss
        self._reduce_sums = None
        self._reduce_flat = None
        self.scoring_precision = "fp32"

    def _lda(self, dev):
        return _to_dev(self.centering_and_LDA.weight, dev), _to_dev(self.centering_and_LDA.bias, dev)

    def _quadform(self, dev):
        D1 = self.centering_and_LDA.out_features
        return ops.dplda_quadform(_to_dev(self.logistic_regres.weight, dev), self.logistic_regres.bias, D1)

    def extract_plda_embeddings(self, x):
        """utils/models.py:479-482: normalize(LDA x) -> (B, D1)."""
        D0, D1 = self.centering_and_LDA.in_features, self.centering_and_LDA.out_features
        x = x.reshape(-1, D0) if x.dim() != 2 else x
        dev = _compute_device(x, self.centering_and_LDA.weight)
        with torch.no_grad():
            xd = _to_dev(x, dev)
            y = ops.gb_paired(xd, xd, *self._lda(dev))[:, :D1]
        return y if y.device == x.device else y.to(x.device)

    def forward_rows(self, table, rows1, rows2):
        """(NeuralPlda's fused gather + score does not apply to the quadratic-form head: gather, then forward.)"""
        return self.forward(ops.gather_rows(table, rows1), ops.gather_rows(table, rows2))

    def forward_from_plda_embeddings(self, x1, x2):
        """utils/models.py:484-490 on (B, D1) embeddings (used as they are, not re-normalised)."""
        D1 = self.centering_and_LDA.out_features
        dev = _compute_device(x1, self.logistic_regres.weight)
        with torch.no_grad():
            M, v, c = self._quadform(dev)
            Dp = (D1 + 3) // 4 * 4  # rows must be float4-addressable: identity layer 1 over zero-padded columns
            eye = torch.eye(D1, Dp, dtype=torch.float32, device=dev)
            packed = ops.quadform_pack(eye, torch.zeros(D1, dtype=torch.float32, device=dev), M, v, c)
            pad = (lambda t: torch.nn.functional.pad(_to_dev(t, dev).float(), (0, Dp - D1)))
            s = ops.quadform_score_rows(pad(x1), pad(x2), packed)
        return s if s.device == x1.device else s.to(x1.device)

    def forward(self, x1, x2):
        """utils/models.py:492-495: (B, D0) x 2 -> (B,)."""
        D0 = self.centering_and_LDA.in_features
        if x1.numel() == 0 and x2.numel() == 0:
            x1, x2 = x1.reshape(0, D0), x2.reshape(0, D0)
        return _DPldaScoreFn.apply(x1, x2, (self.__dict__.get("_reduce_sums64"), torch.is_grad_enabled()), self.centering_and_LDA.weight,
                                   self.centering_and_LDA.bias, self.logistic_regres.weight, self.logistic_regres.bias)

    def crossentropy(self, output, target):
        """utils/models.py:503-506: BCE(sigmoid(output), target) — no threshold."""
        zero = torch.zeros(1, dtype=torch.float32, device=output.device)
        return _LossFn.apply(output, target, ops.LOSS_BCE, 0.0, [], (self._reduce_sums, torch.is_grad_enabled()), zero)

    def LoadParamsFromKaldi(self, mean_vec_file, transform_mat_file):
        """utils/models.py:551-564."""
        kaldi_format.fold_init(self, mean_vec_file, transform_mat_file)

    def LoadPldaParamsFromKaldi(self, *a, **k):
        raise AttributeError("DPlda has no PLDA layer; use LoadParamsFromKaldi(mean_vec_file, transform_mat_file)")


# ---------------------------------------------------------------------------------------------------
# GaussianBackend (forward only is usable in the reference)
# ---------------------------------------------------------------------------------------------------

class GaussianBackend(nn.Module):
    """Drop-in for utils/models.py:571-665 (constructor, forward, forward_getpaired, Kaldi LDA loading)."""

    def __init__(self, nc):
        super(GaussianBackend, self).__init__()
        self.centering_and_LDA = nn.Linear(nc.xvector_dim, nc.layer1_LDA_dim)
        self.centering_and_LDA.weight.requires_grad = False
        self.centering_and_LDA.bias.requires_grad = False
        self.paired_mean_target = torch.rand(2 * nc.layer1_LDA_dim)
        self.paired_cov_inv_target = torch.rand(2 * nc.layer1_LDA_dim, 2 * nc.layer1_LDA_dim)
        self.paired_mean_nontarget = torch.rand(2 * nc.layer1_LDA_dim)
        self.paired_cov_inv_nontarget = torch.rand(2 * nc.layer1_LDA_dim, 2 * nc.layer1_LDA_dim)
        # The reference's GaussianBackend.softcdet/cdet/minc (utils/models.py:603-651) read alpha / beta / threshold /
        # lossfn that its constructor never sets; here they are taken from nc when it has them, as
        # plain attributes (state_dict stays {centering_and_LDA.*} like the reference's).
        self.beta = list(getattr(nc, "beta", []))
        self.alpha = torch.tensor(float(getattr(nc, "alpha", 1.0)))
        self.threshold = {b: torch.zeros(1) for b in self.beta}
        self.lossfn = getattr(nc, "loss", "SoftCdet")
        self._reduce_sums = None

    def __setstate__(self, state):
        super(GaussianBackend, self).__setstate__(state)
        for k, v in (("beta", []), ("alpha", torch.tensor(1.0)), ("threshold", {}), ("lossfn", "SoftCdet"),
                     ("_reduce_sums", None)):
            self.__dict__.setdefault(k, v)

    _alpha = NeuralPlda._alpha
    softcdet = NeuralPlda.softcdet
    loss = NeuralPlda.loss
    cdet = NeuralPlda.cdet

    def crossentropy(self, output, target):
        """utils/models.py:609-612: BCE(sigmoid(output), target)."""
        zero = torch.zeros(1, dtype=torch.float32, device=output.device)
        return _LossFn.apply(output, target, ops.LOSS_BCE, 0.0, [], (self._reduce_sums, torch.is_grad_enabled()), zero)

    def minc(self, output, target, update_thresholds=False, showplots=False, exact=False):
        """utils/models.py:625-651 (thresholds live in self.threshold, not in the state dict)."""
        from . import metrics
        minc_avg, minc_threshold = metrics.minc(output, target, self.beta, reference_semantics=not exact)
        if update_thresholds:
            for beta in self.beta:
                self.threshold[beta] = minc_threshold[beta].detach().reshape(1).clone()
        return minc_avg, minc_threshold

    def _stats(self, dev):
        return [_to_dev(t, dev) for t in (self.paired_mean_target, self.paired_cov_inv_target,
                                          self.paired_mean_nontarget, self.paired_cov_inv_nontarget)]

    def forward(self, x1, x2):
        """utils/models.py:584-593."""
        dev = _compute_device(x1, self.centering_and_LDA.weight)
        W1, b1 = _to_dev(self.centering_and_LDA.weight, dev), _to_dev(self.centering_and_LDA.bias, dev)
        with torch.no_grad():
            s = ops.gb_score_pairs(_to_dev(x1, dev), _to_dev(x2, dev), W1, b1, *self._stats(dev))
        return s if s.device == x1.device else s.to(x1.device)

    def forward_getpaired(self, x1, x2):
        """utils/models.py:595-601: [normalize(LDA x1), normalize(LDA x2)] -> (B, 2 D1)."""
        dev = _compute_device(x1, self.centering_and_LDA.weight)
        W1, b1 = _to_dev(self.centering_and_LDA.weight, dev), _to_dev(self.centering_and_LDA.bias, dev)
        with torch.no_grad():
            x = ops.gb_paired(_to_dev(x1, dev), _to_dev(x2, dev), W1, b1)
        return x if x.device == x1.device else x.to(x1.device)

    # -- closed-form "training" (xvector_GaussianBackend_pytorch.py:30-56) -------------------------------------------
    def accumulate_statistics(self, x1, x2, target, stats=None):
        """One batch of the accumulation loop (:40-52): paired rows x = forward_getpaired(x1, x2), then per class
        count, sum x and sum x x^T in a single weighted-moments pass (class 0 = target > 0.5, class 1 = target < 0.5).
        Returns the running (cnt (2,), sum (2, 2 D1), sq (2, 2 D1, 2 D1)) fp64 device tensors."""
        dev = _compute_device(x1, self.centering_and_LDA.weight)
        W1, b1 = _to_dev(self.centering_and_LDA.weight, dev), _to_dev(self.centering_and_LDA.bias, dev)
        with torch.no_grad():
            x = ops.gb_paired(_to_dev(x1, dev), _to_dev(x2, dev), W1, b1)
            t = _to_dev(target, dev).float()
            return ops.weighted_moments(x, (t > 0.5).float(), (t < 0.5).float(), out=stats)

    def fit_statistics(self, stats):
        """:53-56, quirk included: the non-target class divides both moments by (count - 1), the target class by
        count.  The 2 D1 x 2 D1 inverses are taken in fp64 (the reference: fp32 torch.inverse)."""
        cnt, sm, sq = stats
        ct, cn = cnt[0], cnt[1] - 1
        mu_t, mu_n = sm[0] / ct, sm[1] / cn
        cov_t = sq[0] / ct - torch.outer(mu_t, mu_t)
        cov_n = sq[1] / cn - torch.outer(mu_n, mu_n)
        self.paired_mean_target, self.paired_cov_inv_target = mu_t.float(), torch.linalg.inv(cov_t).float()
        self.paired_mean_nontarget, self.paired_cov_inv_nontarget = mu_n.float(), torch.linalg.inv(cov_n).float()
        return self

    def LoadPldaParamsFromKaldi(self, mean_vec_file, transform_mat_file):
        """utils/models.py:653-658."""
        kaldi_format.fold_init(self, mean_vec_file, transform_mat_file)

    def SaveModel(self, filename):
        with open(filename, 'wb') as f:
            pickle.dump(self, f)
