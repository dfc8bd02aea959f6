"""The forward dispatch model on the host (csrc/nplda_fwd_dispatch.h: pair_kernel_choice), no GPU: which kernel a batch of n
pairs takes and what the model says it costs — compiled host-only (tests/c/dispatch_table.hip, ~3 s).  Properties of the
round-6 FWD_SPLIT form: it is chosen only where it is cheaper than every single-kernel form by the model's own margin, its
first part is a whole number of full rounds of the persistent grid, its remainder is shorter than one round, and the modelled
cost never rises by more than one mid-kernel tile when a batch grows (no cliff between the regimes)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL, MID, STREAM, SPLIT = 0, 1, 2, 3


@pytest.fixture(scope="module")
def table_bin(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = str(tmp_path_factory.mktemp("dispatch") / "dispatch_table")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "--cuda-host-only", "-O1", "-std=c++17", "-Wno-unused-result",
                        "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "neuralplda_amd", "csrc"),
                        os.path.join(ROOT, "tests", "c", "dispatch_table.hip"), "-o", out], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return out


def _table(binary, D, cus):
    out = subprocess.run([binary, str(D), str(cus)], capture_output=True, text=True, timeout=60, check=True).stdout
    return [tuple(int(v) for v in ln.split()) for ln in out.splitlines()]


@pytest.mark.parametrize("D,cus", [(150, 256), (170, 256), (150, 304), (170, 64)])
def test_split_dispatch_properties(table_bin, D, cus):
    rows = _table(table_bin, D, cus)
    assert len(rows) > 100
    rnd = 128 * cus
    nsplit = 0
    prev = None
    for n, k, cost, full, k0, cost0 in rows:
        assert k in (SMALL, MID, STREAM, SPLIT) and k0 in (SMALL, MID, STREAM) and cost > 0 and cost0 > 0
        assert full == n // rnd * rnd
        if k == SPLIT:
            nsplit += 1
            assert 0 < full < n and n - full < rnd       # whole rounds first, less than a round left
            assert cost + 30 < cost0                      # cheaper than the best single kernel by the model's 3 us margin
        else:
            assert (k, cost) == (k0, cost0)              # otherwise exactly the round-5 choice
        if n <= 8 * cus:
            assert k == MID                               # lone half tiles up to 8 pairs per CU
        if prev is not None and n > 16 * cus:
            # growing the batch by g pairs never costs more than the mid kernel's rate for g pairs + one tile (no cliff)
            pn, pc = prev
            per_tile = 133 if D == 150 else 168
            assert cost <= pc + ((n - pn) // (16 * cus) + 2) * per_tile, (pn, pc, n, cost)
        prev = (n, cost)
    assert nsplit > 10  # the sizes past a multiple of the round do take the split form
    # BASELINE's 1 M pairs on 256 CUs are 32 full rounds: streamed whole
    if cus == 256:
        big = [r for r in rows if r[0] == 1 << 20]
        assert not big or big[0][1] == STREAM
