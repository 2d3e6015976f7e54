import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_artefacts():
    """fresh checkout: build libposeidon252_hip.so (hipcc cross-compiles gfx950 on a CPU-only host) and the
    oracle once per session, exactly as __graft_entry__.build() does; a no-op when they are up to date"""
    from poseidon252_amd import build as b
    import oracle
    if not os.path.exists(b.LIB) or not os.path.exists(oracle._LIB_PATH):
        try:
            b.build_library()
        except Exception as e:  # GPU box without hipcc would still have the shipped .so; otherwise tests fail loudly later
            print("build_library failed:", e)
        try:
            oracle.build()
        except Exception as e:
            print("oracle build failed:", e)
    yield


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """plain `pytest tests` on a host without a GPU: every gpu-marked test skips, whether or not it asks for the gpu_ctx
    fixture (several drive subprocesses of their own — ADVICE r2)"""
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def hosttest_lib():
    """CPU build of the device arithmetic headers (same code the kernels run)."""
    import ctypes
    from poseidon252_amd import build as b
    path = b.HOSTTEST_LIB
    if not os.path.exists(path) or os.path.exists("/usr/bin/g++"):
        try:
            path = b.build_hosttest()
        except Exception:
            if not os.path.exists(path):
                raise
    return ctypes.CDLL(path)


@pytest.fixture(scope="session")
def gpu_ctx():
    if not _has_gpu():
        pytest.skip("no GPU")
    import poseidon252_amd as P
    return P.Context(0)
