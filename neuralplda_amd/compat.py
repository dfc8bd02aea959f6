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


def install(force=False):
    """Register the `utils.*` aliases.  Refuses (unless force=True) if a real `utils` package is loaded."""
    from . import NpldaConf, models, scorefile_generator, sv_trials_loaders
    if "utils" in sys.modules and not getattr(sys.modules["utils"], "__neuralplda_amd_alias__", False) and not force:
        raise RuntimeError("a different `utils` package is already imported; pass force=True to shadow it")
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
    for name in _ALIASES:
        if name in _saved:
            sys.modules[name] = _saved.pop(name)
        else:
            sys.modules.pop(name, None)
