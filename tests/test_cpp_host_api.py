"""Compiles and runs tests/cpp/test_hash_api.cpp — the C++ host-side mirror (include/poseidon252.hpp)
of the reference's Hash/Domain API over the C ABI — on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "test_hash_api")
    cmd = ["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "test_hash_api.cpp"),
           "-L", os.path.join(ROOT, "poseidon252_amd"), "-lposeidon252_hip", "-L", os.path.join(ROOT, "oracle"), "-lp252_oracle",
           "-Wl,-rpath," + os.path.join(ROOT, "poseidon252_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64", "-o", exe]
    subprocess.check_call(cmd)
    return exe


def test_cpp_mirror_compiles(tmp_path, oracle_mod):
    """(CPU) the header-only mirror compiles and links against the C ABI"""
    _build(tmp_path)


@pytest.mark.gpu
def test_cpp_mirror_on_gpu(tmp_path, oracle_mod, gpu_ctx):
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, timeout=600)
    assert out.returncode == 0, out.stdout.decode() + out.stderr.decode()
    assert b"ALL PASSED" in out.stdout
