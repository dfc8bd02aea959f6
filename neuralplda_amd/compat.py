"""Make the reference's import paths resolve to this package.

The reference's scripts do `from utils.models import NeuralPlda`, `from utils.sv_trials_loaders import ...`,
`from utils.NpldaConf import NpldaConf`, and its checkpoints are whole-module pickles that name the class
`utils.models.NeuralPlda` (utils/models.py:459-461, xvector_generate_scores.py:39).  `install()` registers
alias modules under those names so that both the unchanged scripts and existing pickles bind to the
MI355X-native implementations:

    import neuralplda_amd.compat as compat; compat.install()
    model = pickle.load(open('models/NPLDA_12_1600000000.pt', 'rb'))   # -> neuralplda_amd.models.NeuralPlda
"""
import sys
import types

__all__ = ["install", "uninstall"]

_ALIASES = ("utils", "utils.models", "utils.sv_trials_loaders", "utils.scorefile_generator", "utils.NpldaConf")
_saved = {}


def install(force=False, fused_adam=False, inline_backward=None, deferred_keyerror=False):
    """Register the `utils.*` aliases.  Refuses (unless force=True) if a real `utils` package is loaded.

    fused_adam=True (opt-in) also replaces the name `torch.optim.Adam` by a class that builds `neuralplda_amd.optim.FusedAdam`
    — torch.optim.Adam's update as ONE launch per step — when it is handed plain float32 HIP tensors with torch's default
    flags (what xvector_NeuralPlda_pytorch.py:139 does: `optim.Adam(model.parameters(), lr=nc.lr, weight_decay=1e-5)`), and
    torch's own Adam for everything else.  torch's foreach Adam is eleven launches and ~0.1 - 0.18 ms of host time per step
    over a NeuralPlda's eight small tensors: 45 % of the reference's loop body on this build.  `uninstall()` restores it.

    inline_backward (default: the value of fused_adam) runs `loss.backward()` on the calling thread
    (`torch.autograd.set_multithreading_enabled(False)`) instead of handing the graph to autograd's per-device worker
    thread: the reference's graph is ten nodes on one device, and the hand-over plus the two threads taking turns on the
    interpreter lock cost 0.06 - 0.1 ms of a 0.3 - 0.4 ms loop body (`loss.backward()` 0.155 - 0.19 -> 0.10 ms, the
    `optimizer.step()` that follows 0.06 - 0.08 -> 0.04 ms; profiles/r05p_dropin_ab.txt).  Process-wide, like torch's own
    switch; `uninstall()` restores the previous setting.

    deferred_keyerror=True (opt-in, off by default) lets `load_xvec_trials_from_numbatch` return without waiting for its
    gather kernel: the reference's KeyError for a trial number that names no x-vector (utils/sv_trials_loaders.py:418-426)
    is then raised by the NEXT loader call, by `validate()` or by `ops.check_trial_indices()` instead of at once, the bad
    batch's rows being NaN in the meantime (ops.gather_pairs_mapped).  Saves the loop one device round trip per batch."""
    from . import NpldaConf, models, scorefile_generator, sv_trials_loaders
    if "utils" in sys.modules and not getattr(sys.modules["utils"], "__neuralplda_amd_alias__", False) and not force:
        raise RuntimeError("a different `utils` package is already imported; pass force=True to shadow it")
    if fused_adam and "torch.optim.Adam" not in _saved:
        import torch
        from . import optim as _optim
        _saved["torch.optim.Adam"] = torch.optim.Adam
        torch.optim.Adam = _optim.adam_factory(torch.optim.Adam)
    if inline_backward is None:
        inline_backward = bool(fused_adam)
    if inline_backward and "autograd.multithreading" not in _saved:
        import torch
        _saved["autograd.multithreading"] = torch.autograd.is_multithreading_enabled()
        torch.autograd.set_multithreading_enabled(False)
    if deferred_keyerror and "ops.KEYERROR_DEFERRED" not in _saved:
        from . import ops
        _saved["ops.KEYERROR_DEFERRED"] = ops.KEYERROR_DEFERRED
        ops.KEYERROR_DEFERRED = True
    for name in _ALIASES:
        if name in sys.modules and name not in _saved:
            _saved[name] = sys.modules[name]
    pkg = types.ModuleType("utils")
    pkg.__path__ = []
    pkg.__neuralplda_amd_alias__ = True
    pkg.models, pkg.sv_trials_loaders = models, sv_trials_loaders
    pkg.scorefile_generator, pkg.NpldaConf = scorefile_generator, NpldaConf
    sys.modules["utils"] = pkg
    sys.modules["utils.models"] = models
    sys.modules["utils.sv_trials_loaders"] = sv_trials_loaders
    sys.modules["utils.scorefile_generator"] = scorefile_generator
    sys.modules["utils.NpldaConf"] = NpldaConf
    return pkg


def uninstall():
    if "torch.optim.Adam" in _saved:
        import torch
        torch.optim.Adam = _saved.pop("torch.optim.Adam")
    if "autograd.multithreading" in _saved:
        import torch
        torch.autograd.set_multithreading_enabled(_saved.pop("autograd.multithreading"))
    if "ops.KEYERROR_DEFERRED" in _saved:
        from . import ops
        ops.KEYERROR_DEFERRED = _saved.pop("ops.KEYERROR_DEFERRED")
    for name in _ALIASES:
        if name in _saved:
            sys.modules[name] = _saved.pop(name)
        else:
            sys.modules.pop(name, None)
