"""Build libcpmpc.so (HIP kernels + C-ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libcpmpc.so")
SOURCES = [os.path.join(CSRC, "cpmpc_api.hip")]
DEPS = SOURCES + [
    os.path.join(CSRC, "mpc_kernels.hpp"),
    os.path.join(CSRC, "cartpole_device.hpp"),
    os.path.join(HERE, "..", "include", "cpmpc.h"),
]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm; set HIPCC)")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build_lib(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-o", LIB] + SOURCES
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
