"""The CPU-side code under AddressSanitizer + UndefinedBehaviorSanitizer: the oracle (all its composite paths on exactly
sized heap buffers) and the DEVICE arithmetic headers compiled for the host — where UBSan's signed-overflow check turns
"the signed 64-bit columns of the lazy arithmetic never overflow" from an analysis (tables.hpp max_column_bound29) into an
executed check on random and saturated inputs, for the one-lane schedules and the lane-group schedule alike.
(The reference is single-threaded safe Rust with no sanitizer tooling of its own — SURVEY.md §5; this is the build's.)"""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-O1"]  # (no -g: halves the build time)


@pytest.mark.skipif(shutil.which("g++") is None or shutil.which("gcc") is None, reason="needs gcc / g++")
def test_oracle_and_host_arithmetic_under_asan_ubsan(tmp_path, oracle_mod, hosttest_lib):
    # (the two fixtures make sure the generated asset includes exist)
    obj = tmp_path / "oracle.o"
    subprocess.check_call(["gcc", "-std=gnu11", "-c"] + SAN + [os.path.join(ROOT, "oracle", "p252_oracle.c"), "-o", str(obj)])
    exe = tmp_path / "sanitize_run"
    subprocess.check_call(["g++", "-std=c++17", "-Wno-unknown-pragmas", "-DP252_TRACK_BOUNDS", "-pthread"] + SAN +
                          [os.path.join(ROOT, "tests", "cpp", "sanitize_run.cpp"), str(obj), "-o", str(exe)])
    kat = json.load(open(os.path.join(ROOT, "tests", "golden", "hades_kat.json")))["expected_be_hex"]
    args = [x for n, h in kat.items() for x in (n, h)]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([str(exe)] + args, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "clean" in r.stdout and "6 KAT digests" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
