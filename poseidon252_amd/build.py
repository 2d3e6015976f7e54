"""Compiles libposeidon252_hip.so for gfx950 with hipcc (in-tree, next to this file)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libposeidon252_hip.so")
HOSTTEST_LIB = os.path.join(CSRC, "libp252_hosttest.so")

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    # the round bodies are fully unrolled straight-line code; the default pragma-unroll budget
    # silently turns them into scratch-indexed loops
    "-mllvm", "-pragma-unroll-threshold=1000000",
]
SOURCES = ["kernels.hip", "openings.hip", "merkle2.hip", "api.cpp", "comm.cpp", "rccl_dyn.cpp"]
# comm.cpp: the RCCL communicator of the multi-GPU entry points (ncclBroadcast of the constants, ncclAllGather of subtree roots).
# RCCL is NOT linked: rccl_dyn.cpp resolves it with dlopen on first use (a hashing-only deployment needs no RCCL, and a
# process that already holds a copy — torch's — gets that one); only its header is needed at build time.
# (ROCM_PATH / HIP_PATH name the ROCm prefix when it is not /opt/rocm — ADVICE r4)
ROCM = os.environ.get("ROCM_PATH") or os.environ.get("HIP_PATH") or "/opt/rocm"
LINK_FLAGS = ["-ldl"]
HEADERS = ["fr29.hpp", "fr_host.hpp", "hades29.hpp", "coop29.hpp", "tables.hpp", "kernels.h", "blake2b.hpp", "ctx.hpp", "openings.h", "fastdiv.hpp", "rccl_dyn.hpp",
           os.path.join("..", "..", "include", "poseidon252_hip.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), os.path.join(ROCM, "bin", "hipcc"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _gen_assets():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_assets_inc
    gen_assets_inc.main(os.path.join(CSRC, "_gen", "assets.inc"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_library(force=False, verbose=False):
    """every source to its own object (only what changed is recompiled: kernels.hip takes ~50 s, the host files seconds),
    then one link.  Objects live in csrc/_gen/obj (git-ignored)."""
    _gen_assets()
    hdeps = [os.path.join(CSRC, f) for f in HEADERS] + [os.path.join(CSRC, "_gen", "assets.inc"), os.path.abspath(__file__)]
    objdir = os.path.join(CSRC, "_gen", "obj")
    os.makedirs(objdir, exist_ok=True)
    objs, relink = [], force or not os.path.exists(LIB)
    for src in SOURCES:
        obj = os.path.join(objdir, src + ".o")
        if force or _stale(obj, [os.path.join(CSRC, src)] + hdeps):
            cmd = [_hipcc()] + HIPCC_FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd, cwd=CSRC)
            relink = True
        objs.append(obj)
    if relink or _stale(LIB, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + LINK_FLAGS + ["-o", LIB]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=CSRC)
    return LIB


def build_hosttest(force=False):
    """CPU build of the device arithmetic headers, used by the not-gpu unit tests only."""
    _gen_assets()
    deps = [os.path.join(CSRC, f) for f in ["hosttest.cpp"] + HEADERS]
    if force or _stale(HOSTTEST_LIB, deps):
        # -DP252_TRACK_BOUNDS: the reductions record the largest column / top digit they meet (test_dynamic_bounds)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wno-unknown-pragmas", "-DP252_TRACK_BOUNDS",
                               os.path.join(CSRC, "hosttest.cpp"), "-o", HOSTTEST_LIB])
    return HOSTTEST_LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
