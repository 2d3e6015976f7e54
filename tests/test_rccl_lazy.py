"""RCCL is a RUN-TIME dependency of the communicator entry points only (ABI 8; csrc/rccl_dyn.hpp): the library
loads and hashes without it, says exactly what it looked for when a communicator is asked for, takes the copy the process already
holds, and accepts an explicit file (P252_RCCL_PATH — how the suite's mock comes in).  All of this is host logic: no GPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROBE = r"""
import ctypes, json, sys
sys.path.insert(0, %r)
from poseidon252_amd import _lib
L = _lib.lib()
out = {}
tag = (ctypes.c_uint64 * 4)()
lens = (ctypes.c_size_t * 1)(4)
out["tag_rc"] = L.p252_tag(0, lens, 1, 1, tag)
out["tag"] = list(tag)
sep = ctypes.c_uint64(0)
out["sep_rc"] = L.p252_domain_separator(0, ctypes.byref(sep))
buf = ctypes.create_string_buffer(128)
out["uid_rc"] = L.p252_comm_unique_id(buf, 128)
out["uid_err"] = L.p252_last_error(None).decode()
path = ctypes.create_string_buffer(4096)
out["backend_rc"] = L.p252_comm_backend(path, 4096)
out["backend"] = path.value.decode()
out["maps"] = sorted(set(l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l))
print(json.dumps(out))
""" % ROOT


def _probe(env_extra, pre=""):
    import json
    env = dict(os.environ, **env_extra)
    for k in [k for k, v in env_extra.items() if v is None]:
        env.pop(k)
    r = subprocess.run([sys.executable, "-c", pre + PROBE], cwd=ROOT, env={k: v for k, v in env.items() if v is not None}, capture_output=True, timeout=300)
    assert r.returncode == 0, r.stdout.decode() + r.stderr.decode()[-3000:]
    return json.loads(r.stdout.decode().strip().splitlines()[-1])


def test_without_rccl_hashing_entry_points_work_and_the_communicator_says_what_it_tried(tmp_path):
    """RCCL hidden (P252_RCCL_PATH names a file that is not there: the explicit path is the only thing tried): the library loads,
    p252_tag / p252_domain_separator succeed, p252_comm_unique_id returns P252_ERR_COMM and p252_last_error lists the attempt"""
    missing = str(tmp_path / "no_such_librccl.so")
    o = _probe({"P252_RCCL_PATH": missing})
    assert o["tag_rc"] == 0 and any(o["tag"]) and o["sep_rc"] == 0
    assert o["uid_rc"] == -6 and o["backend_rc"] == -6 and o["backend"] == ""
    assert "RCCL is not available" in o["uid_err"] and "P252_RCCL_PATH=" + missing in o["uid_err"], o["uid_err"]
    assert o["maps"] == []  # and nothing of RCCL was mapped on the way


def _stub(tmp_path, drop=None):
    names = ["ncclGetUniqueId", "ncclCommInitRank", "ncclCommInitAll", "ncclCommDestroy", "ncclCommAbort", "ncclGroupStart", "ncclGroupEnd",
             "ncclBroadcast", "ncclAllGather", "ncclGetErrorString"]
    body = "#include <string.h>\n"
    for n in names:
        if n == drop:
            continue
        if n == "ncclGetUniqueId":
            body += "int ncclGetUniqueId(char* id) { memset(id, 0, 128); memcpy(id, \"STUBRCCL\", 8); return 0; }\n"
        elif n == "ncclGetErrorString":
            body += "const char* ncclGetErrorString(int r) { (void)r; return \"stub\"; }\n"
        else:
            body += "int %s(void) { return 1; }\n" % n
    src = tmp_path / ("stub_%s.c" % (drop or "full"))
    src.write_text(body)
    so = tmp_path / ("libstub_rccl_%s.so" % (drop or "full"))
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-o", str(so), str(src)])
    return str(so)


def test_an_explicit_file_is_used_and_an_incomplete_one_is_refused(tmp_path):
    full = _stub(tmp_path)
    o = _probe({"P252_RCCL_PATH": full})
    assert o["uid_rc"] == 0 and o["backend_rc"] == 0 and os.path.samefile(o["backend"], full), o
    assert [m for m in o["maps"] if "stub" not in m] == []  # the explicit file only: no second RCCL next to it
    part = _stub(tmp_path, drop="ncclAllGather")
    o = _probe({"P252_RCCL_PATH": part})
    assert o["uid_rc"] == -6 and "no symbol ncclAllGather" in o["uid_err"], o["uid_err"]  # never a table mixed from two copies


def test_a_copy_the_process_already_holds_is_preferred():
    """after `import torch` the process holds torch's bundled librccl (SONAME librccl.so.1, loaded RTLD_LOCAL): the resolver finds it
    with dlopen(RTLD_NOLOAD) and maps nothing else — one RCCL per process, the state to be in with several real ranks"""
    pytest.importorskip("torch")
    o = _probe({"P252_RCCL_PATH": None}, pre="import torch\n")
    assert o["backend_rc"] == 0 and len(o["maps"]) == 1 and os.path.realpath(o["maps"][0]) == os.path.realpath(o["backend"]), o
    assert os.sep + "torch" + os.sep in o["backend"], o


def test_a_c_caller_without_torch_gets_the_system_copy():
    """no torch in the process, no explicit path (a C or Rust host program): the loader's search path / $ROCM_PATH/lib serves it.
    The raw ctypes load below does what such a program does — it does not go through the Python wrappers' torch preference."""
    if not os.path.exists("/opt/rocm/lib/librccl.so.1"):
        pytest.skip("no system RCCL in this image")
    code = ("import ctypes\nL = ctypes.CDLL(%r)\nbuf = ctypes.create_string_buffer(4096)\nrc = L.p252_comm_backend(buf, 4096)\n"
            "print(rc, buf.value.decode())\n" % os.path.join(ROOT, "poseidon252_amd", "libposeidon252_hip.so"))
    env = dict(os.environ)
    env.pop("P252_RCCL_PATH", None)
    out = subprocess.check_output([sys.executable, "-c", code], env=env, timeout=300).decode().split()
    assert out[0] == "0" and "librccl.so" in out[1] and os.sep + "torch" + os.sep not in out[1], out
