"""The Rust shim cannot be compiled here (no toolchain), so its FFI surface is kept honest mechanically:
bindings/rust/src/sys.rs is generated from include/poseidon252_hip.h, must be up to date, and must declare exactly the
symbols the shared library exports with the header's arity; lib.rs may only call functions sys.rs declares."""
import ctypes
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_sys_rs_is_in_sync_with_the_header():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_sys.py"), "--check"], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()


def test_every_header_function_is_declared_with_the_same_arity():
    import gen_rust_sys as g
    _, protos = g.parse_header()
    text = open(g.SYS_RS).read()
    rust = {m.group(1): m.group(2) for m in re.finditer(r"pub fn (p252_\w+)\((.*?)\)", text)}
    assert sorted(rust) == sorted(name for name, _, _ in protos) and len(rust) >= 35
    for name, _, params in protos:
        n_rust = 0 if not rust[name].strip() else rust[name].count(":")
        assert n_rust == len(params), name
    from poseidon252_amd import _lib
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in rust:
        assert hasattr(L, name), name
    assert sorted(rust) == sorted(_lib.ABI_SYMBOLS)


def test_lib_rs_calls_only_declared_functions_and_no_private_dusk_api():
    lib = open(os.path.join(ROOT, "bindings", "rust", "src", "lib.rs")).read()
    sysrs = open(os.path.join(ROOT, "bindings", "rust", "src", "sys.rs")).read()
    declared = set(re.findall(r"pub fn (p252_\w+)", sysrs))
    code = re.sub(r"//.*", "", lib)  # the doc comments mention p252_tag, which the shim deliberately never calls
    used = set(re.findall(r"\b(p252_\w+)\s*\(", code))
    assert used and used <= declared, used - declared
    assert "tag_input(" not in code                        # dusk_safe::tag_input is crate-private
    assert "impl Safe<BlsScalar, 5> for TagProbe" in lib    # the tag comes from a tag-capturing Safe (src/hades.rs:63-92 pattern)
    parity = open(os.path.join(ROOT, "bindings", "rust", "tests", "parity.rs")).read()
    for shape in ("(Domain::Other, 3, 3, ", "(Domain::Other, 5, 2, ", "(Domain::Other, 4, 7, ", "[2usize, 21, 42]",
                  "(Domain::Merkle4, 4, 1, 20_000)"):  # the reference's tests/hash.rs shapes; a batch beyond the lane-group kernels
        assert shape in parity, shape
    # what round 5 added is covered too — the truncated form (pins the `&`), trees of both arities as loops of
    # Hash::digest, one bulk verification of openings; each reports a RUSTPARITY line run_parity.sh collects and the prediction holds
    for needle in ("fn gpu_matches_reference_truncated()", "hb.digest_truncated_raw(&input)", "h.finalize_truncated()", "JubJubScalar::from_raw(",
                   "fn gpu_trees_match_loops_of_hash_digest()", "h4.merkle4_root(&leaves)", "h2.merkle2_root(&leaves)", "h4.merkle4_path_roots(",
                   "RUSTPARITY truncated", "RUSTPARITY trees"):
        assert needle in parity, needle
    import json
    exp = json.load(open(os.path.join(ROOT, "bindings", "rust", "RUSTPARITY.expected.json")))
    assert exp["truncated"]["matches"] is True and exp["trees"] == {"merkle4": True, "merkle2": True, "openings_verify": True}
    assert set(exp["tests"]) >= {"gpu_matches_reference_truncated", "gpu_trees_match_loops_of_hash_digest"}


def test_run_parity_script_names_only_files_that_exist():
    """bindings/rust/run_parity.sh cannot run here (no cargo): at least every repository path it and
    RUN.md name exists, the script is valid bash, its collector is valid Python, and parity.rs prints the lines it collects"""
    rust = os.path.join(ROOT, "bindings", "rust")
    script = open(os.path.join(rust, "run_parity.sh")).read()
    assert subprocess.run(["bash", "-n", os.path.join(rust, "run_parity.sh")]).returncode == 0
    py = re.search(r"<<'PY'\n(.*?)\nPY\n", script, re.S).group(1)
    compile(py, "run_parity collector", "exec")
    for rel in ("poseidon252_amd/build.py", "bindings/rust/Cargo.toml", "bindings/rust/build.rs", "bindings/rust/src/lib.rs", "bindings/rust/src/sys.rs",
                "bindings/rust/tests/parity.rs", "bindings/rust/.cargo/config.toml", "bindings/rust/RUN.md"):
        assert os.path.exists(os.path.join(ROOT, rel)), rel
    assert "python -m poseidon252_amd.build" in script and "cargo test --release" in script and "RUSTPARITY.json" in script
    assert "run_parity.sh" in open(os.path.join(rust, "RUN.md")).read()
    parity = open(os.path.join(rust, "tests", "parity.rs")).read()
    assert parity.count("RUSTPARITY tag_input") >= 3 and "RUSTPARITY encryption" in parity
    # the collector understands the lines parity.rs prints
    sample = ('test gpu_matches_reference_hash ... ok\nRUSTPARITY tag_input {"pattern": "merkle4_4_1", "bytes": [128, 0, 0, 4], "tag_limbs": [1, 2, 3, 4]}\n'
              'RUSTPARITY encryption {"len": 21, "stream": true, "duplex": false, "tag_input": [1], "permutations": 13}\n')
    import json, tempfile
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "RUSTPARITY.log"), "w").write(sample)
        r = subprocess.run([sys.executable, "-c", py, "0"], cwd=d, capture_output=True)
        assert r.returncode == 0, r.stderr.decode()
        out = json.load(open(os.path.join(d, "RUSTPARITY.json")))
    assert out["ok"] and out["tests"] == {"gpu_matches_reference_hash": "ok"} and out["encryption"]["21"]["stream"] is True
    assert out["tag_inputs"]["merkle4_4_1"]["bytes"] == [128, 0, 0, 4]
    assert "p252_abi_version()" in open(os.path.join(rust, "src", "lib.rs")).read()


def test_rustparity_expected_is_the_librarys_prediction():
    """bindings/rust/RUSTPARITY.expected.json holds what the library predicts the parity run will print —
    kept in sync with the library's host helpers (and an independent hashlib re-derivation) by the generator; run_parity.sh
    diffs its result against it.  Here: in sync, every pattern parity.rs prints is predicted, and the differ works."""
    import json
    gen = os.path.join(ROOT, "tools", "gen_rustparity_expected.py")
    r = subprocess.run([sys.executable, gen, "--check"], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    exp = json.load(open(os.path.join(ROOT, "bindings", "rust", "RUSTPARITY.expected.json")))
    parity = open(os.path.join(ROOT, "bindings", "rust", "tests", "parity.rs")).read()
    printed = set(re.findall(r'\\"pattern\\": \\"([a-z0-9_+]+)\\"', parity)) | set(re.findall(r'\("([a-z0-9_+]+)", Domain::', parity))
    assert printed == set(exp["tag_inputs"]) and set(exp["encryption"]) == {"2", "21", "42"}, (printed, set(exp["tag_inputs"]))
    assert exp["tag_inputs"]["merkle4_4_1"]["bytes"] == [0x80, 0, 0, 4, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0xF]
    assert exp["tag_inputs"]["other_42_1"] == dict(exp["tag_inputs"]["other_3+39_1"], pattern="other_42_1")  # adjacent absorbs aggregate
    assert [exp["encryption"][k]["duplex"] for k in ("2", "21", "42")] == [True, False, False]
    assert "RUSTPARITY.expected.json" in open(os.path.join(ROOT, "bindings", "rust", "run_parity.sh")).read() or "--diff RUSTPARITY.json" in open(os.path.join(ROOT, "bindings", "rust", "run_parity.sh")).read()
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        same = os.path.join(d, "same.json")
        json.dump(dict(exp, log="RUSTPARITY.log"), open(same, "w"))
        r = subprocess.run([sys.executable, gen, "--diff", same], capture_output=True)
        assert r.returncode == 0 and b"PINNED" in r.stdout
        other = json.loads(json.dumps(exp))
        other["tag_inputs"]["merkle4_4_1"]["bytes"][0] = 0
        other["encryption"]["21"]["stream"], other["encryption"]["21"]["duplex"] = False, True
        del other["tag_inputs"]["other_42_5"]
        json.dump(other, open(same, "w"))
        r = subprocess.run([sys.executable, gen, "--diff", same], capture_output=True)
        lines = r.stdout.decode().splitlines()
        assert r.returncode == 1 and len(lines) == 4, lines
        assert any("tag_inputs.merkle4_4_1.bytes" in l for l in lines) and any("encryption.21.stream" in l for l in lines) and any("missing from the run" in l for l in lines)


def test_rust_sources_are_lexically_balanced():
    """No Rust toolchain here: the least that can be checked mechanically is that every source of the shim and of refgen has balanced
    (), [], {} outside comments, strings and char literals — a truncated edit or a stray brace is caught before someone with cargo is"""
    def balanced(path):
        s = open(path).read()
        out, i, n = [], 0, len(s)
        while i < n:
            if s.startswith("//", i):
                j = s.find("\n", i)
                i = n if j < 0 else j
            elif s.startswith("/*", i):
                i = s.find("*/", i + 2) + 2
            elif s[i] == '"':
                j = i + 1
                while j < n and s[j] != '"':
                    j += 2 if s[j] == "\\" else 1
                i = j + 1
            elif s[i] == "'" and re.match(r"'(\\.|[^\\'])'", s[i:]):
                i += re.match(r"'(\\.|[^\\'])'", s[i:]).end()
            else:
                out.append(s[i])
                i += 1
        stack, pairs = [], {")": "(", "]": "[", "}": "{"}
        for ch in out:
            if ch in "([{":
                stack.append(ch)
            elif ch in ")]}":
                if not stack or stack.pop() != pairs[ch]:
                    return False
        return not stack
    rust = os.path.join(ROOT, "bindings", "rust")
    for rel in ("src/lib.rs", "src/sys.rs", "tests/parity.rs", "build.rs", "refgen/src/main.rs"):
        assert balanced(os.path.join(rust, rel)), rel
