"""The library's multi-rank logic with MORE THAN ONE RANK on a one-GPU box.  Real RCCL refuses two ranks on a device and no
multi-GPU node was available, so csrc/comm.cpp's rank logic — communicator creation by id (one thread per rank: what one process
per GPU does) and over an array of contexts, the validated broadcast of the constants in a group of N, the all-gather of the N
subtree roots, the top levels on every rank (N = 3, 5, 6: zero-padded top nodes, hash.rs:22-26), teardown in either order — runs
here against tests/cpp/mock_rccl.cpp: the library's own objects linked with a stand-in for the ten RCCL calls instead of librccl.
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
    b.build_library()
    objdir = os.path.join(b.CSRC, "_gen", "obj")
    objs = [os.path.join(objdir, s + ".o") for s in b.SOURCES]
    assert all(os.path.exists(o) for o in objs), objs
    so = str(tmp_path / "libposeidon252_hip_mockrccl.so")
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared"] + objs +
                          [os.path.join(ROOT, "tests", "cpp", "mock_rccl.cpp"), "-o", so])
    return so


def test_mock_library_links_without_rccl(tmp_path):
    """(CPU) the library's objects + the mock resolve every RCCL symbol comm.cpp uses: nothing else of RCCL is called"""
    so = _build_mock(tmp_path)
    dyn = subprocess.check_output(["readelf", "-d", so]).decode()
    assert "librccl" not in dyn and "libamdhip64" in dyn
    undefined = subprocess.check_output(["nm", "-D", "--undefined-only", so]).decode()
    assert " nccl" not in undefined, undefined


@pytest.mark.gpu
def test_multi_rank_logic_against_the_mock(tmp_path, gpu_ctx, oracle_mod):
    so = _build_mock(tmp_path)
    env = dict(os.environ, P252_LIB_PATH=so, P252_COMM_ALLOW_SHARED_DEVICE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "mock_ranks_driver.py")], cwd=ROOT, env=env, capture_output=True, timeout=900)
    assert r.returncode == 0, r.stdout.decode()[-2000:] + r.stderr.decode()[-3000:]
    rep = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert len(rep) == 3, rep
