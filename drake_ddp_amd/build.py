"""Build libmi_ilqr.so (HIP, gfx950) in-tree with hipcc.

    python -m drake_ddp_amd.build [--force]

The .so lands in drake_ddp_amd/lib/ (git-ignored, but it travels to the GPU box
with the gpurun snapshot).  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libmi_ilqr.so")
SOURCES = [os.path.join(CSRC, "mi_ilqr.hip")]
DEPS = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "mi_ilqr.h")]

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-fast-math",
         "-ffp-contract=fast", "-Wall", "-Wno-unused-function"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS if os.path.exists(d))


def build(force=False, verbose=True, extra=()):
    os.makedirs(LIBDIR, exist_ok=True)
    if not force and not needs_build():
        return LIB
    cmd = [HIPCC] + FLAGS + list(extra) + SOURCES + ["-o", LIB, "-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
