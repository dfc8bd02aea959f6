"""The C ABI from plain C: include/nplda_hip.h must compile as C99 and libnplda_hip.so must link and answer without
Python, torch or a GPU (argument validation and the host-side text entry points only)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_program_links_and_runs(tmp_path, hip_lib):
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    libdir = os.path.join(ROOT, "neuralplda_amd")
    exe = str(tmp_path / "abi_smoke")
    rocm_lib = "/opt/rocm/lib"
    cmd = [gcc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-o", exe, "-L", libdir, "-lnplda_hip",
           f"-Wl,-rpath,{libdir}", f"-Wl,-rpath,{rocm_lib}", f"-Wl,-rpath-link,{rocm_lib}"]
    subprocess.run(cmd, check=True, capture_output=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert out.stdout.startswith("abi ")


def test_dynamic_symbol_table_is_the_declared_abi_only(hip_lib):
    """The library is built -fvisibility=hidden with a linker version script: `nm -D` must list the entry points
    include/nplda_hip.h declares and nothing else (no C++ kernel stubs, no helper such as the former `step_ws`)."""
    import re
    nm = shutil.which("nm")
    if nm is None:
        pytest.skip("no nm")
    so = os.path.join(ROOT, "neuralplda_amd", "libnplda_hip.so")
    out = subprocess.run([nm, "-D", "--defined-only", so], check=True, capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    hdr = open(os.path.join(ROOT, "include", "nplda_hip.h")).read()
    declared = set(re.findall(r"\b((?:nplda|gb)_[a-z0-9_]+)\s*\(", hdr))
    assert exported == declared, sorted(exported ^ declared)


def test_cohort_sizes_include_the_split_image(hip_lib):
    """Round 6: the fixed part of a cohort workspace / a CohortState holds the cohort's split image (three bf16 pieces in
    fragment order: 12 * ceil(16 NB / 32) KiB per 64 cohort rows) — pure host arithmetic, no device needed; a cohort whose
    image would pass 1 GiB is not eligible for the fused path (the sizing functions budget 4 GiB per workspace)."""
    import ctypes
    lib = hip_lib
    for fn in (lib.nplda_cohort_state_bytes, lib.nplda_cohort_fused_min_workspace_bytes):
        fn.restype = ctypes.c_size_t
        fn.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    for D, kib in ((150, 60), (170, 72)):
        M = 10000
        image = (M + 63) // 64 * kib * 1024
        state = lib.nplda_cohort_state_bytes(M, 500, D, D)
        assert image < state < image + (4 << 20), (D, state, image)
        assert lib.nplda_cohort_fused_min_workspace_bytes(M, 500, D, D) > state
    assert lib.nplda_cohort_fused_min_workspace_bytes(1_000_000, 500, 150, 150) > 0
    assert lib.nplda_cohort_state_bytes(1_200_000, 500, 150, 150) == 0       # 1.07 GiB of image: the spilling path
    assert lib.nplda_cohort_fused_min_workspace_bytes(1_200_000, 500, 150, 150) == 0
