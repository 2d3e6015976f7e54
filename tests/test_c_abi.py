"""include/poseidon252_hip.h is valid C: tests/c/abi_smoke.c is compiled with gcc -std=c11 -pedantic -Werror and linked
against libposeidon252_hip.so (CPU: it runs, sees no device, checks the host helpers and the loud failure; GPU: the
whole hot path through the C ABI from C)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "abi_smoke")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-L", os.path.join(ROOT, "poseidon252_amd"), "-lposeidon252_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "poseidon252_amd"), "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-o", exe])
    return exe


def test_header_compiles_as_c11_and_links():
    import torch
    exe_dir = os.path.join(ROOT, ".pytest_cache", "c_abi")
    os.makedirs(exe_dir, exist_ok=True)
    import pathlib
    exe = _build(pathlib.Path(exe_dir))
    if not torch.cuda.is_available():
        out = subprocess.run([exe], capture_output=True, timeout=120)
        assert out.returncode == 0, out.stdout.decode() + out.stderr.decode()
        assert b"no device" in out.stdout and b"gfx950" in out.stdout


@pytest.mark.gpu
def test_c_consumer_on_gpu(tmp_path, gpu_ctx):
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, timeout=600)
    assert out.returncode == 0, out.stdout.decode() + out.stderr.decode()
    assert b"ABI SMOKE PASSED" in out.stdout
