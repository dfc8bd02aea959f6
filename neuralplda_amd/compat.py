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


def install(force=False, fused_adam=False):
    """Register the `utils.*` aliases.  Refuses (unless force=True) if a real `utils` package is loaded.

    fused_adam=True (opt-in) also replaces the name `torch.optim.Adam` by a class that builds `neuralplda_amd.optim.FusedAdam`
    — torch.optim.Adam's update as ONE launch per step — when it is handed plain float32 HIP tensors with torch's default
    flags (what xvector_NeuralPlda_pytorch.py:139 does: `optim.Adam(model.parameters(), lr=nc.lr, weight_decay=1e-5)`), and
    torch's own Adam for everything else.  torch's foreach Adam is eleven launches and ~0.1 - 0.18 ms of host time per step
    over a NeuralPlda's eight small tensors: 45 % of the reference's loop body on this build.  `uninstall()` restores it."""
    from . import NpldaConf, models, scorefile_generator, sv_trials_loaders
    if "utils" in sys.modules and not getattr(sys.modules["utils"], "__neuralplda_amd_alias__", False) and not force:
        raise RuntimeError("a different `utils` package is already imported; pass force=True to shadow it")
    if fused_adam and "torch.optim.Adam" not in _saved:
        import torch
        from . import optim as _optim
        _saved["torch.optim.Adam"] = torch.optim.Adam
        torch.optim.Adam = _optim.adam_factory(torch.optim.Adam)
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
    for name in _ALIASES:
        if name in _saved:
            sys.modules[name] = _saved.pop(name)
        else:
            sys.modules.pop(name, None)
