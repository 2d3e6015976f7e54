"""The library's multi-rank logic with MORE THAN ONE RANK on a one-GPU box.  Real RCCL refuses two ranks on a device and no
multi-GPU node was available, so csrc/comm.cpp's rank logic — communicator creation by id (one thread per rank: what one process
per GPU does) and over an array of contexts, the validated broadcast of the constants in a group of N, the all-gather of the N
subtree roots, the top levels on every rank (N = 3, 5, 6: zero-padded top nodes, hash.rs:22-26), a rank whose local build fails
(the sentinel its peers must detect), teardown in either order — runs here against tests/cpp/mock_rccl.cpp, a stand-in
for the ten RCCL calls.  Since ABI 8 the library resolves RCCL at run time (csrc/rccl_dyn.hpp), so the mock comes in the way any
RCCL does: a shared object named by P252_RCCL_PATH, under the SHIPPED library — not a special link of the library's objects.
The real backend is covered at one rank (tests/test_comm_forest.py, tests/c/abi_smoke.c, tests/cpp/test_hash_api.cpp) and by the
driver's 8-GPU run."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_mock(tmp_path):
    from poseidon252_amd import build as b
    so = str(tmp_path / "libmock_rccl.so")
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared",
                           os.path.join(ROOT, "tests", "cpp", "mock_rccl.cpp"), "-o", so])
    return so


def test_mock_exports_exactly_what_the_resolver_asks_for(tmp_path):
    """(CPU) the ten symbols csrc/rccl_dyn.cpp looks up are the ten the mock defines: nothing else of RCCL is called, and the
    shipped library resolves the mock through P252_RCCL_PATH (no device needed for that)"""
    import re
    so = _build_mock(tmp_path)
    exported = set(l.split()[-1] for l in subprocess.check_output(["nm", "-D", "--defined-only", so]).decode().splitlines() if " T nccl" in l)
    asked = set(re.findall(r'\{"(nccl[A-Za-z]+)"', open(os.path.join(ROOT, "poseidon252_amd", "csrc", "rccl_dyn.cpp")).read()))
    assert len(asked) == 10 and asked == exported, (sorted(asked), sorted(exported))
    code = ("import ctypes\nfrom poseidon252_amd import _lib\nL = _lib.lib()\nbuf = ctypes.create_string_buffer(4096)\n"
            "rc = L.p252_comm_backend(buf, 4096)\nprint(rc, buf.value.decode())\n"
            "i = ctypes.create_string_buffer(128)\nprint(L.p252_comm_unique_id(i, 128), i.raw[:8])\n")
    out = subprocess.check_output([sys.executable, "-c", code], cwd=ROOT, env=dict(os.environ, P252_RCCL_PATH=so)).decode().splitlines()
    assert out[-2].split()[0] == "0" and os.path.samefile(out[-2].split()[1], so), out
    assert out[-1].startswith("0 ") and "MOCKRCCL" in out[-1], out


@pytest.mark.gpu
def test_multi_rank_logic_against_the_mock(tmp_path, gpu_ctx, oracle_mod):
    so = _build_mock(tmp_path)
    env = dict(os.environ, P252_RCCL_PATH=so, P252_COMM_ALLOW_SHARED_DEVICE="1")
    env.pop("P252_LIB_PATH", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "mock_ranks_driver.py")], cwd=ROOT, env=env, capture_output=True, timeout=900)
    assert r.returncode == 0, r.stdout.decode()[-2000:] + r.stderr.decode()[-3000:]
    rep = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert len(rep) == 4, rep
