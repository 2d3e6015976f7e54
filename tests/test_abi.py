"""The C-ABI shared library loads and exports every symbol include/poseidon252_hip.h declares; host
helpers (no device needed) behave like the reference's src/hash.rs; compute entry points fail LOUDLY
without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "poseidon252_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(p252_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from poseidon252_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build with python -m poseidon252_amd.build"
    L = ctypes.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), "missing export " + s
    assert sorted(_lib.ABI_SYMBOLS) == syms
    L.p252_version.restype = ctypes.c_char_p
    assert b"gfx950" in L.p252_version()


def test_library_has_gfx950_code_object_and_no_oracle():
    from poseidon252_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    assert b"p252o_" not in blob  # the product library contains nothing of oracle/


def test_product_package_never_imports_oracle():
    import subprocess, sys
    code = "import sys; import poseidon252_amd; assert 'oracle' not in sys.modules; print('ok')"
    assert subprocess.check_output([sys.executable, "-c", code], cwd=ROOT).strip() == b"ok"
    for dirpath, _, files in os.walk(os.path.join(ROOT, "poseidon252_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")) and f != "hosttest.cpp":
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "p252_oracle" not in src, f


def test_domain_separators_and_io_pattern():
    import poseidon252_amd as P
    assert [d.separator() for d in P.Domain] == [0xF, 0x3, 0x1_0000_0000, 0]  # hash.rs:38-56
    assert [int(d) for d in (P.Domain.Merkle4, P.Domain.Merkle2, P.Domain.Encryption, P.Domain.Other)] == [0, 1, 2, 3]
    P.check_io_pattern(P.Domain.Merkle4, [4], 1)
    P.check_io_pattern(P.Domain.Merkle4, [1, 3], 1)
    P.check_io_pattern(P.Domain.Other, [3, 39], 5)
    for dom, lens, out in [(P.Domain.Merkle4, [3], 1), (P.Domain.Merkle4, [4], 2), (P.Domain.Merkle2, [4], 1), (P.Domain.Merkle2, [1], 1)]:
        with pytest.raises(P.IOPatternViolation):  # hash.rs:70-78
            P.check_io_pattern(dom, lens, out)
    for dom, lens, out in [(P.Domain.Other, [], 1), (P.Domain.Other, [2, 0], 1), (P.Domain.Other, [2], 0), (P.Domain.Encryption, [0], 1)]:
        with pytest.raises(P.InvalidIOPattern):
            P.check_io_pattern(dom, lens, out)


def test_tag_helper_matches_oracle_recipe(oracle_mod):
    import poseidon252_amd as P
    for dom, lens, out in [(0, [4], 1), (1, [2], 1), (3, [42], 5), (3, [42], 1), (3, [3, 39], 5), (2, [6], 1), (3, [1], 1), (3, [1000], 77)]:
        assert np.array_equal(P.compute_tag(P.Domain(dom), lens, out), oracle_mod.tag(dom, lens, out))


def test_truncate250(oracle_mod):
    import poseidon252_amd as P
    x = oracle_mod.fill_random(5, 64)
    got = P.truncate250(x)
    assert np.array_equal(got, np.stack([oracle_mod.truncate250(v) for v in x]))
    assert (got[:, 3] >> np.uint64(58)).max() == 0  # top 6 bits cleared (hash.rs:167-172)


def test_levels_len():
    from poseidon252_amd import _lib, levels_len
    L = _lib.lib()
    for n, exp in [(1, 0), (2, 1), (4, 1), (5, 3), (16, 5), (17, 8), (1 << 24, 5592405)]:
        assert L.p252_merkle4_levels_len(n) == exp == levels_len(n)
    assert L.p252_merkle4_levels_len(0) == 0


def test_no_gpu_fails_loudly():
    import torch
    import poseidon252_amd as P
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(P.DeviceError, match="no CPU fallback"):
        P.Context(0)
    with pytest.raises(P.DeviceError):
        P.Hash.digest(P.Domain.Merkle4, np.zeros((4, 4), dtype=np.uint64))


def test_missing_extension_fails_loudly(monkeypatch):
    from poseidon252_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libposeidon252_hip.so")
    with pytest.raises(_lib.ExtensionMissing, match="no CPU fallback"):
        _lib.lib()


def test_abi_version_is_checked_by_the_binding():
    """argument lists changed between library versions under unchanged names; the header carries a version, the
    library reports the one it was built from, and the Python binding refuses a library of another version"""
    from poseidon252_amd import _lib
    text = open(HEADER).read()
    ver = int(re.search(r"#define P252_ABI_VERSION (\d+)", text).group(1))
    L = ctypes.CDLL(_lib.LIB_PATH)
    L.p252_abi_version.restype = ctypes.c_int
    assert L.p252_abi_version() == ver == _lib.ABI_VERSION
    assert _lib.lib().p252_abi_version() == ver  # (lib() would have raised ExtensionMissing on a mismatch)
    # the staging-lane budget of the multi entry points: clamp(floor(cpus / n_ctx), 1, 3), no device needed
    L.p252_staging_lanes.argtypes = [ctypes.c_size_t]
    import bench
    cpus = bench.usable_cpus()
    for n_ctx in (2, 4, 8, 64):
        assert L.p252_staging_lanes(n_ctx) == max(1, min(3, cpus // n_ctx)), n_ctx
    assert L.p252_staging_lanes(1) == L.p252_staging_lanes(0) == (2 if cpus < 4 else 3)
