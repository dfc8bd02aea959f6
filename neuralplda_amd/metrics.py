"""Detection-cost metrics of the validation loop (utils/models.py:406-436).

The reference's `minc` is an O(N_tgt * N) Python loop with a `.cpu().item()` per element; here it is a
device sort plus binary searches (torch ops on whatever device the scores live on — this module is
device-agnostic host logic, not part of the HIP hot path).  Two modes:

* reference_semantics=True (default): bit-compatible with the reference INCLUDING its quirks
  (utils/models.py:23-27 `arr2val`): the "count" is the last index of torch.where (= count - 1) and
  1.0 when the set is empty; thresholds are swept over the target scores only.
* reference_semantics=False: the exact minimum of P_miss + beta * P_fa over all thresholds.
"""
import torch

__all__ = ["minc", "eer", "minc_exact"]


def _split(output, target):
    output = output.detach().reshape(-1).float()
    target = target.detach().reshape(-1).to(output.device)
    st, _ = torch.sort(output[target > 0.5])
    sn, _ = torch.sort(output[target < 0.5])
    return output, target, st, sn


def minc(output, target, betas, reference_semantics=True):
    """Returns (minc_avg: 0-d float32 tensor, {beta: 0-d threshold tensor})."""
    if not reference_semantics:
        return minc_exact(output, target, betas)
    output, target, st, sn = _split(output, target)
    nt = target.float().sum()
    nn = (1 - target.float()).sum()
    c_lt = torch.searchsorted(st, st, right=False)                    # targets strictly below st[i]
    c_ge = sn.shape[0] - torch.searchsorted(sn, st, right=False)      # non-targets >= st[i]
    one = torch.ones((), dtype=torch.float32, device=output.device)
    pmiss_arr = torch.where(c_lt > 0, (c_lt - 1).float(), one)
    pfa_arr = torch.where(c_ge > 0, (c_ge - 1).float(), one)
    pmiss = pmiss_arr / nt
    pfa = pfa_arr / nn
    mincs, ths = [], {}
    for beta in betas:
        c = pmiss + beta * pfa
        v, idx = torch.min(c, 0)
        mincs.append(v)
        ths[beta] = st[idx]
    return sum(mincs) / len(mincs), ths


def minc_exact(output, target, betas):
    """True minimum detection cost: decide 'target' iff s >= th, th swept over every score and +inf."""
    output, target, st, sn = _split(output, target)
    th = torch.cat([torch.unique(output), torch.full((1,), float("inf"), device=output.device)])
    pmiss = torch.searchsorted(st, th, right=False).double() / max(st.shape[0], 1)
    pfa = (sn.shape[0] - torch.searchsorted(sn, th, right=False)).double() / max(sn.shape[0], 1)
    mincs, ths = [], {}
    for beta in betas:
        c = pmiss + beta * pfa
        v, idx = torch.min(c, 0)
        mincs.append(v)
        ths[beta] = th[idx]
    return (sum(mincs) / len(mincs)).float(), ths


def eer(output, target):
    """Equal error rate (linear interpolation at the P_miss / P_fa crossing)."""
    output, target, st, sn = _split(output, target)
    th = torch.unique(output)
    pmiss = torch.searchsorted(st, th, right=False).double() / st.shape[0]
    pfa = (sn.shape[0] - torch.searchsorted(sn, th, right=False)).double() / sn.shape[0]
    d = pmiss - pfa
    i = int(torch.argmax((d >= 0).to(torch.int8)).item())
    if i == 0:
        return float((pmiss[0] + pfa[0]) / 2)
    x0, x1 = d[i - 1], d[i]
    w = float(-x0 / (x1 - x0)) if float(x1 - x0) != 0 else 0.5
    return float(pmiss[i - 1] + w * (pmiss[i] - pmiss[i - 1]))
