"""Detection-cost metrics of the validation loop (utils/models.py:406-436).

The reference's `minc` is an O(N_tgt * N) Python loop with a `.cpu().item()` per element; here it is one call of
nplda_detcost_sweep_f32 (device radix sort of (score, label) + prefix scan + one sweep kernel, csrc/nplda_detcost.hip).
CPU tensors are moved to the HIP device first — like the rest of the package there is no CPU implementation.  Two modes:

* reference_semantics=True (default): bit-compatible with the reference INCLUDING its quirks
  (utils/models.py:23-27 `arr2val`): the "count" is the last index of torch.where (= count - 1) and
  1.0 when the set is empty; thresholds are swept over the target scores only.
* reference_semantics=False: the exact minimum of P_miss + beta * P_fa over all thresholds.
"""
import torch

from . import _lib, ops

__all__ = ["minc", "eer", "minc_exact"]


def _on_device(output, target):
    dev = None
    for t in (output, target):
        if isinstance(t, torch.Tensor) and t.is_cuda:
            dev = t.device
            break
    if dev is None:
        if not torch.cuda.is_available():
            raise _lib.NpldaHipError("detection-cost metrics need a HIP device (there is no CPU implementation)")
        dev = torch.device("cuda", torch.cuda.current_device())
    s = output.detach().reshape(-1).to(dev, torch.float32)
    t = target.detach().reshape(-1).to(dev, torch.float32)
    return s, t


def _sweep(output, target, betas, exact, want_eer=False):
    s, t = _on_device(output, target)
    mc, th, avg, e = ops.detcost_sweep(s, t, betas, exact=exact, want_eer=want_eer)
    odev = output.device
    back = (lambda x: x if x is None or x.device == odev else x.to(odev))
    return back(mc), back(th), back(avg), back(e)


def minc(output, target, betas, reference_semantics=True):
    """Returns (minc_avg: 0-d float32 tensor, {beta: 0-d threshold tensor})."""
    _, th, avg, _ = _sweep(output, target, list(betas), exact=not reference_semantics)
    return avg.reshape(()), {beta: th[k] for k, beta in enumerate(betas)}


def minc_exact(output, target, betas):
    """True minimum detection cost: decide 'target' iff s >= th, th swept over every score and +inf."""
    return minc(output, target, betas, reference_semantics=False)


def eer(output, target):
    """Equal error rate (linear interpolation at the P_miss / P_fa crossing)."""
    _, _, _, e = _sweep(output, target, [1.0], exact=True, want_eer=True)
    return float(e.item())
