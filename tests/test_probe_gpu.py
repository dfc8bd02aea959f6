"""The shader-clock probe behind bench.py's `roofline.sclk_mhz_under_kernel` (nplda_clock_probe)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_clock_probe_reads_a_plausible_clock(hip_lib):
    from neuralplda_amd import _lib
    lib = _lib.load()
    ticks = torch.zeros(2, dtype=torch.int64, device="cuda")
    _lib.check(lib.nplda_clock_probe(_lib.ptr(ticks), 2000, _lib.current_stream()), "nplda_clock_probe")
    torch.cuda.synchronize()
    cycles, t100 = (int(v) for v in ticks.cpu())
    assert 2000 * 100 <= t100 <= 2200 * 100          # the window, in ticks of the constant 100 MHz counter
    mhz = 100.0 * cycles / t100
    assert 500.0 < mhz < 2600.0                       # idle chips may sit below the 2.4 GHz boost clock
    # argument validation
    assert lib.nplda_clock_probe(None, 100, None) < 0
    assert lib.nplda_clock_probe(_lib.ptr(ticks), 0, None) < 0
