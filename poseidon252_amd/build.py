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
SOURCES = ["kernels.hip", "api.cpp"]
HEADERS = ["fr29.hpp", "fr_host.hpp", "hades29.hpp", "coop29.hpp", "tables.hpp", "kernels.h", "blake2b.hpp",
           os.path.join("..", "..", "include", "poseidon252_hip.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
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
    _gen_assets()
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(CSRC, "_gen", "assets.inc")]
    if force or _stale(LIB, deps):
        cmd = [_hipcc()] + HIPCC_FLAGS + ["-shared"] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
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
