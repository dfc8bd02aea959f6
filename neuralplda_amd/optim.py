"""torch.optim.Adam as the reference configures it (xvector_NeuralPlda_pytorch.py:139: `optim.Adam(model.parameters(),
lr=nc.lr, weight_decay=1e-5)`) — ONE launch per step (nplda_adam_step_f32) instead of torch's eleven foreach launches and
their Python over eight small tensors, which is 45 % of the reference's literal loop body on this build's modules
(profiles/r04i_dropin_torchprof.txt).

`FusedAdam` is a `torch.optim.Optimizer`: `zero_grad()`, `param_groups` (the reference halves `lr` by re-creating the
optimiser, :174-179; writing `param_groups[0]['lr']` works too), `state_dict()` round trips.  Same update as
torch.optim.Adam (L2 weight decay folded into the gradient, bias-corrected moments, eps added to the root) on float32 HIP
tensors; anything else (amsgrad, maximize, CPU tensors, several devices) is refused, and `compat.install(fused_adam=True)`
then leaves torch's own class in charge.
"""
import ctypes

import torch
import torch.optim.optimizer as _opt

from . import _lib

__all__ = ["FusedAdam", "adam_factory"]

_MAX_SEG = 12  # nplda_adam_step_f32: segments per launch


class FusedAdam(torch.optim.Optimizer):
    """Adam over float32 HIP parameters, one kernel launch per group of <= 12 tensors per step."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= weight_decay:
            raise ValueError("FusedAdam: lr, eps and weight_decay must be non-negative")
        if not (0.0 <= betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError("FusedAdam: betas must be in [0, 1)")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        for g in self.param_groups:
            for p in g["params"]:
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise ValueError("FusedAdam needs contiguous float32 parameters on a HIP device")
            # a group's moments and step counters live in flat buffers on ONE device and every launch of the group runs
            # there (_init_group): a group that spans devices would hand the kernel wrong-device pointers
            if len({p.device for p in g["params"]}) > 1:
                raise ValueError("FusedAdam: the parameters of one group must live on one device")
        self._lib = _lib.load()
        self._plans = {}  # group index -> pre-marshalled launch blocks

    # the moments of a group live in ONE flat buffer each (views per parameter in self.state: state_dict() sees tensors)
    def _init_group(self, group):
        ps = list(group["params"])
        dev = ps[0].device
        total = sum(p.numel() for p in ps)
        m = torch.zeros(total, dtype=torch.float32, device=dev)
        v = torch.zeros(total, dtype=torch.float32, device=dev)
        o, taken = 0, {}
        for p in ps:
            st = self.state[p]
            n = p.numel()
            if "exp_avg" in st:  # load_state_dict() handed moments over: adopt their values
                m[o:o + n].copy_(st["exp_avg"].reshape(-1))
                v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            st["exp_avg"], st["exp_avg_sq"] = m[o:o + n].view_as(p), v[o:o + n].view_as(p)
            s = st.get("step")
            taken[p] = int(s.item() if torch.is_tensor(s) else s) if s is not None else 0
            o += n
        # counters: (parameter addresses of a launch) -> device [steps taken as a float, launch scratch]
        return {"m": m, "v": v, "dev": dev, "taken": taken, "counters": {}, "blocks": {}}

    def _block(self, plan, live):
        """Pre-marshalled argument block of one launch over `live` = [(param, grad), ...] (<= 12 segments that have taken
        the same number of steps), cached on the tensors' addresses: a step then costs one ctypes call."""
        pkey = tuple(p.data_ptr() for p, _ in live)
        want = plan["taken"][live[0][0]]
        ctr = plan["counters"].get(pkey)
        if ctr is None or ctr[1] != want:
            # torch counts steps per parameter (one that had no gradient for a while lags behind): a launch's device
            # counter starts at what its parameters have taken
            if ctr is None:
                ctr = plan["counters"][pkey] = [torch.zeros(2, dtype=torch.float32, device=plan["dev"]), want]
            ctr[0][0] = float(want)
            ctr[1] = want
            for p, _ in live:
                self.state[p]["step"] = ctr[0][0:1].view(())
        ctr[1] += 1  # (the launch this block is for counts the step on the device)
        step = ctr[0]
        key = pkey + tuple(g.data_ptr() for _, g in live)
        blk = plan["blocks"].get(key)
        if blk is None:
            if len(plan["blocks"]) >= 16:  # (gradients re-allocated at ever new addresses: keep the cache bounded)
                plan["blocks"].clear()
            n = len(live)
            arr = lambda ptrs: (ctypes.c_void_p * n)(*ptrs)  # noqa: E731
            blk = plan["blocks"][key] = (
                arr(pkey), arr([g.data_ptr() for _, g in live]),
                arr([self.state[p]["exp_avg"].data_ptr() for p, _ in live]),
                arr([self.state[p]["exp_avg_sq"].data_ptr() for p, _ in live]),
                (ctypes.c_int64 * n)(*[p.numel() for p, _ in live]), n, step.data_ptr())
        return blk

    def zero_grad(self, set_to_none=True):
        if not set_to_none:
            return super().zero_grad(set_to_none=False)
        for group in self.param_groups:  # (what torch's does, without its profiler range and per-device bookkeeping)
            for p in group["params"]:
                p.grad = None

    def step(self, closure=None):
        """One launch per <= 12 tensors.  torch wraps every optimiser's step() in a profiler range plus its hook loops
        (Optimizer.profile_hook_step, ~8 us per call); that wrapper is taken only when a step hook is registered."""
        if (self._optimizer_step_pre_hooks or self._optimizer_step_post_hooks or _opt._global_optimizer_pre_hooks
                or _opt._global_optimizer_post_hooks):
            return self._step_hooked(closure)
        return self._step(closure)

    step.hooked = True  # (Optimizer.__init__ leaves a step() marked like this alone)

    def _step(self, closure=None):
        # (no torch op below touches a parameter: raw launches on data_ptr()s; nothing to shield from autograd)
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            if not group["params"]:
                continue  # (an empty group: nothing to plan, nothing to step)
            plan = self._plans.get(gi)
            if plan is None:
                plan = self._plans[gi] = self._init_group(group)
            taken = plan["taken"]
            live = []
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if g.is_sparse or g.dtype != torch.float32 or g.device != p.device:
                    raise RuntimeError("FusedAdam: gradients must be dense float32 tensors on the parameter's device")
                if not g.is_contiguous():
                    g = p.grad = g.contiguous()
                live.append((p, g))
            if not live:
                continue
            if len({taken[p] for p, _ in live}) > 1:  # launches of parameters that have taken the same number of steps
                live.sort(key=lambda pg: taken[pg[0]])
            b1, b2 = group["betas"]
            with _lib.on_device(plan["dev"]):
                st = _lib.current_stream(plan["dev"])
                lo = 0
                while lo < len(live):
                    hi = lo + 1
                    while hi < len(live) and hi - lo < _MAX_SEG and taken[live[hi][0]] == taken[live[lo][0]]:
                        hi += 1
                    pa, ga, ma, va, na, n, step_ptr = self._block(plan, live[lo:hi])
                    code = self._lib.nplda_adam_step_f32(pa, ga, ma, va, na, n, step_ptr, float(group["lr"]), float(b1),
                                                         float(b2), float(group["eps"]), float(group["weight_decay"]), st)
                    if code:
                        _lib.check(code, "nplda_adam_step_f32")
                    lo = hi
            for p, _ in live:
                taken[p] += 1
                # a raw kernel wrote the parameter: autograd's saved-tensor checks and the packed-image cache
                # (models._packed_for) key on the version counter
                torch.autograd.graph.increment_version(p)
        return loss

    _step_hooked = torch.optim.Optimizer.profile_hook_step(_step)

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._plans = {}  # rebuilt from the loaded moments at the next step


def adam_factory(real_adam):
    """A stand-in for `torch.optim.Adam` (compat.install(fused_adam=True)): FusedAdam where it applies — float32 HIP
    parameters, torch's default flags — and torch's own Adam otherwise.  isinstance(opt, torch.optim.Adam) keeps working for
    the torch instances; the fused ones are torch.optim.Optimizer."""

    class Adam(real_adam):
        __doc__ = real_adam.__doc__

        def __new__(cls, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kw):
            params = list(params)
            # an ALLOW-list: FusedAdam only when every further keyword sits at torch's default (False / None).  Anything
            # else — `decoupled_weight_decay=True` (AdamW-style decay, which FusedAdam's L2 decay is not), a keyword a
            # later torch adds — goes to torch's own Adam with the keyword intact; this patch is process-wide.
            plain = (not amsgrad and all(v is None or v is False for v in kw.values())
                     and not torch.is_tensor(lr) and len(params) > 0 and all(torch.is_tensor(p) for p in params)
                     and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in params)
                     and len({p.device for p in params}) == 1)
            if plain:
                try:
                    return FusedAdam(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
                except (_lib.NpldaHipError, ValueError):
                    pass
            return real_adam(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, **kw)

    Adam.__name__ = Adam.__qualname__ = "Adam"
    return Adam
