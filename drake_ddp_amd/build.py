"""Build libmi_ilqr.so (HIP, gfx950) in-tree with hipcc.

    python -m drake_ddp_amd.build [--force] [--resource-usage]

One translation unit per model's kernels (csrc/k_<model>.hip) plus the host side of the C ABI
(csrc/mi_ilqr.hip), compiled in parallel and linked into drake_ddp_amd/lib/libmi_ilqr.so (git-ignored,
but it travels to the GPU box with the gpurun snapshot).  hipcc cross-compiles without a GPU.
Objects are cached in drake_ddp_amd/lib/obj and rebuilt when their source, any header or the flags change.
"""
import concurrent.futures
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libmi_ilqr.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -amdgpu-mfma-vgpr-form: MFMA accumulators / results in the architectural VGPRs (gfx90a+ has one register file) instead of the
# AGPRs the compiler's heuristic picks - in the backward passes every MFMA result feeds VALU work (Vxx' = H + 2Q - ..., stores)
# and 29 % of the arm's hottest MFMA loop were v_accvgpr moves.  Same-box A/B, round 5: backward step of the arm 5.5 k -> 5.15 k
# cycles, C5 350 k -> 359 k, C6 166 k -> 171 k, C3 18.7 M -> 19.1 M it/s; results bitwise unchanged (same instructions, other registers).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-fast-math",
         "-ffp-contract=fast", "-Wall", "-Wno-unused-function", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def headers():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")) + \
        [os.path.join(HERE, "..", "include", "mi_ilqr.h")]


def _obj(src, tag):
    return os.path.join(OBJDIR, os.path.basename(src)[:-4] + tag + ".o")


def _stale(obj, src):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in [src] + headers() if os.path.exists(d))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in sources() + headers() if os.path.exists(d))


def _compile(src, obj, flags, verbose):
    cmd = [HIPCC] + flags + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    return src, r.returncode, r.stdout


def build(force=False, verbose=True, extra=(), lib=LIB):
    """Compile (in parallel) and link.  `extra`: additional compiler flags (they get their own object cache);
    `lib`: output path."""
    os.makedirs(OBJDIR, exist_ok=True)
    if not force and not extra and lib == LIB and not needs_build():
        return lib
    flags = FLAGS + list(extra)
    tag = ("-" + hashlib.sha1(" ".join(extra).encode()).hexdigest()[:8]) if extra else ""
    jobs = [(s, _obj(s, tag)) for s in sources()]
    todo = [(s, o) for s, o in jobs if force or _stale(o, s)]
    logs = {}
    if todo:
        workers = max(1, min(len(todo), os.cpu_count() or 4))
        with concurrent.futures.ThreadPoolExecutor(workers) as ex:
            for src, rc, out in ex.map(lambda so: _compile(so[0], so[1], flags, verbose), todo):
                logs[src] = out
                if rc != 0:
                    sys.stderr.write(out)
                    raise subprocess.CalledProcessError(rc, [HIPCC, src])
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + [o for _, o in jobs] + ["-o", lib, "-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    if "-Rpass-analysis=kernel-resource-usage" in extra:
        with open(os.path.join(LIBDIR, "resource_usage.txt"), "w") as f:
            for s in sorted(logs):
                f.write(logs[s])
    return lib


ASAN_LIB = os.path.join(LIBDIR, "libmi_ilqr_asan.so")
ASAN_RT = "/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so"


def build_asan(verbose=False):
    """libmi_ilqr_asan.so: the HOST side of the C ABI (csrc/mi_ilqr.hip) instrumented with AddressSanitizer, linked
    with the regular kernel objects (device code cannot be instrumented on gfx950).  tests/test_gpu_asan.py drives it."""
    build(verbose=verbose)
    host_src = os.path.join(CSRC, "mi_ilqr.hip")
    obj = os.path.join(OBJDIR, "mi_ilqr-asan.o")
    if _stale(obj, host_src):
        cmd = [HIPCC] + [f for f in FLAGS if f != "-O3"] + ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address", "-shared-libsan",
                                                           "-Wno-option-ignored", "-c", host_src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    if not os.path.exists(ASAN_LIB) or os.path.getmtime(ASAN_LIB) < max(os.path.getmtime(obj), os.path.getmtime(LIB)):
        objs = [_obj(s_, "") for s_ in sources() if os.path.basename(s_) != "mi_ilqr.hip"]
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address", "-shared-libsan", "-Wno-option-ignored",
                               obj] + objs + ["-o", ASAN_LIB, "-ldl"])
    return ASAN_LIB


if __name__ == "__main__":
    ex = ["-Rpass-analysis=kernel-resource-usage"] if "--resource-usage" in sys.argv else []
    build(force="--force" in sys.argv, extra=ex)
    if "--asan" in sys.argv or os.path.exists(ASAN_LIB):      # (an existing sanitizer build follows the headers: a stale one refuses plugins)
        build_asan(verbose="--asan" in sys.argv)
