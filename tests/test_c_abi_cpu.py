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
