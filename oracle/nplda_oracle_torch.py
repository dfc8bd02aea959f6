"""TEST INFRASTRUCTURE — torch-CPU restatement of the reference's forward, used ONLY as bench.py's `cpu_baseline`
(kind "port") and by tests/test_oracle_golden.py, never by the product path (neuralplda_amd/ has no CPU path).

Why a second restatement next to the NumPy oracle: the reference executes `NeuralPlda.forward` as torch CPU ops
(nn.Linear -> F.normalize -> nn.Linear, then seven elementwise / reduce ops: utils/models.py:366-382), and BASELINE.md §3
asks for the CPU baseline in exactly that form (torch intra-op threads = physical cores, and 1), so that the number is
what a user of the reference sees on the same host.  Pinned against the reference's own outputs (G2) like the NumPy one.
The batch gather of utils/sv_trials_loaders.py:418-426 (a Python loop with two dict look-ups per pair — the reference's
real bottleneck) is restated too, for the gather-inclusive figure."""
import numpy as np
import torch
import torch.nn.functional as F


class TorchParams:
    def __init__(self, W1, b1, W2, b2, P_sqrt, Q):
        as_t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))  # noqa: E731
        self.W1, self.b1, self.W2, self.b2, self.P_sqrt, self.Q = (as_t(a) for a in (W1, b1, W2, b2, P_sqrt, Q))


def extract_plda_embeddings(x, p):
    """utils/models.py:366-370."""
    x = F.linear(x, p.W1, p.b1)
    x = F.normalize(x)
    return F.linear(x, p.W2, p.b2)


def forward(x1, x2, p):
    """utils/models.py:372-382, op for op."""
    z1 = extract_plda_embeddings(x1, p)
    z2 = extract_plda_embeddings(x2, p)
    P = p.P_sqrt * p.P_sqrt
    Q = p.Q
    return (z1 * Q * z1).sum(dim=1) + (z2 * Q * z2).sum(dim=1) + 2 * (z1 * P * z2).sum(dim=1)


def gather_numbatch(mega_dict, num_to_id_dict, data1, data2):
    """utils/sv_trials_loaders.py:418-426: per-pair Python dict look-ups, then one array -> tensor conversion."""
    d1 = np.asarray([mega_dict[num_to_id_dict[int(i)]] for i in data1])
    d2 = np.asarray([mega_dict[num_to_id_dict[int(i)]] for i in data2])
    return torch.from_numpy(d1).float(), torch.from_numpy(d2).float()
