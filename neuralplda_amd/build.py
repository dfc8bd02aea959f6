"""Build libnplda_hip.so (gfx950) in-tree with hipcc.

The shared library is a plain C-ABI object (include/nplda_hip.h): no torch headers, no
pybind — it is bound from Python with ctypes (neuralplda_amd/_lib.py).  hipcc cross-compiles
for gfx950 without a GPU, so this runs in the CPU-only build container too.
"""
import glob
import hashlib
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libnplda_hip.so")
ARCH = "gfx950"
VERSION_SCRIPT = os.path.join(CSRC, "libnplda_hip.map")  # exports nplda_* / gb_* only


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libnplda_hip.so")
    return exe


def sources():
    """Device code (*.hip) and the host-only translation units (*.cpp), all built by hipcc into one library."""
    return sorted(glob.glob(os.path.join(CSRC, "*.hip"))) + sorted(glob.glob(os.path.join(CSRC, "*.cpp")))


def source_sha():
    """sha256 over csrc/* and include/nplda_hip.h (names and contents, sorted), first 16 hex digits; None without sources."""
    files = sorted(glob.glob(os.path.join(CSRC, "*")) + glob.glob(os.path.join(PKG_DIR, "..", "include", "*.h")))
    files = [f for f in files if os.path.isfile(f)]
    if not files:
        return None
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every csrc/*.hip into objects and link libnplda_hip.so. Returns the library path."""
    hipcc = _hipcc()
    srcs = sources()
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(PKG_DIR, "..", "include", "*.h"))
    objdir = os.path.join(PKG_DIR, "build")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    sha = source_sha() or "unknown"
    for src in srcs:
        obj = os.path.join(objdir, os.path.splitext(os.path.basename(src))[0] + ".o")
        objs.append(obj)
        is_id = os.path.basename(src) == "nplda_build_id.cpp"
        if is_id:  # rebuilt whenever the digest of the sources changes
            stamp = os.path.join(objdir, "build_id.txt")
            have = open(stamp).read().strip() if os.path.exists(stamp) else ""
            need = have != sha or not os.path.exists(obj)
        else:
            need = _stale(obj, [src] + hdrs)
        if force or need:
            cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-c", src, "-o", obj]
            if is_id:
                cmd.insert(1, f'-DNPLDA_SRC_SHA="{sha}"')
                with open(stamp, "w") as fh:
                    fh.write(sha)
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode(errors='replace')}")
    if force or procs or _stale(LIB_PATH, objs + [VERSION_SCRIPT]):
        cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", f"-Wl,--version-script={VERSION_SCRIPT}",
               "-o", LIB_PATH] + objs
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout.decode(errors='replace')}")
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
