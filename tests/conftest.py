import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def hip_lib():
    """Build (if stale) and load libnplda_hip.so; fails loudly when it cannot."""
    from neuralplda_amd import _lib, build
    # build.build() is incremental: it recompiles only the csrc/ units newer than their objects and relinks only when
    # something changed, so a library that is stale relative to csrc/ (e.g. a snapshot pushed to the GPU box with an old
    # .so) is rebuilt here, not just a missing one.  Without hipcc an existing library is used as it is.
    import shutil
    if shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc") or not os.path.exists(_lib.LIB_PATH):
        build.build()
    return _lib.load()
