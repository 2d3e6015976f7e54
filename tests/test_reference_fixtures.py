"""Reference fixtures: tests/golden/reference_fixtures.json is written by bindings/rust/refgen — a crate that depends
only on the dusk crates, no GPU, no libposeidon252_hip.so — from the REFERENCE itself: Safe::tag (src/hades/permutation/scalar.rs:29-31)
for every io-pattern shape, Hash::finalize / finalize_truncated (src/hash.rs:128-183) and encrypt / decrypt (src/encryption.rs:62-95)
on RNG-free inputs.  It cannot be made in this image (no cargo); until someone drops it in, the tests that need it SKIP with that
reason, and what runs is the same comparison against tests/golden/reference_fixtures.predicted.json — this repository's side of the
same schema (recollected tag recipe + CPU oracle) — so that the day the file appears, a wrong recollection reads as a one-line diff.

CPU: oracle == fixtures.  GPU (`-m gpu`): HIP library == fixtures, through the host-buffer C ABI."""
import copy
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
import reference_fixtures_lib as F  # noqa: E402

REFERENCE = os.environ.get("P252_REFERENCE_FIXTURES") or os.path.join(HERE, "golden", "reference_fixtures.json")
PREDICTED = os.path.join(HERE, "golden", "reference_fixtures.predicted.json")
ABSENT = ("tests/golden/reference_fixtures.json is absent: it is written by `cd bindings/rust/refgen && cargo run --release` on any machine "
          "with cargo and the dusk crates (no GPU needed) — this image has no Rust toolchain; tag / truncation / encryption stay UNPINNED")


class OracleBackend(F.Backend):
    name = "the oracle"

    def __init__(self, oracle_mod):
        self.o = oracle_mod

    def tag(self, domain, lens, out_len):
        return self.o.tag(F.DOMAIN_ID[domain], lens, out_len)

    def hash(self, tag, inp, in_len, out_len):
        return self.o.hash_batch(np.asarray(tag, dtype=np.uint64), inp.reshape(1, in_len, 4), in_len, out_len).reshape(out_len, 4)

    def hash_truncated(self, tag, inp, in_len, out_len):
        return np.stack([self.o.truncate250(row) for row in self.hash(tag, inp, in_len, out_len)])

    def encryption_tag(self, ln):
        return self.o.encryption_tag(ln)

    def encrypt(self, tag, message, secret, nonce):
        return self.o.encrypt_batch(tag, message.reshape(1, -1, 4), secret.reshape(1, 2, 4), nonce.reshape(1, 4)).reshape(-1, 4)

    def decrypt(self, tag, cipher, secret, nonce):
        back, ok = self.o.decrypt_batch(tag, cipher.reshape(1, -1, 4), secret.reshape(1, 2, 4), nonce.reshape(1, 4))
        return back.reshape(-1, 4), bool(ok[0])


class HipBackend(F.Backend):
    """the product: libposeidon252_hip.so through its Python binding (host-buffer C ABI entry points; kernels on the GPU)"""
    name = "the HIP library"

    def __init__(self, ctx):
        import poseidon252_amd as P
        from poseidon252_amd import encryption as Enc
        self.P, self.Enc, self.ctx = P, Enc, ctx

    def tag(self, domain, lens, out_len):
        return self.P.compute_tag(self.P.Domain(F.DOMAIN_ID[domain]), lens, out_len)

    def hash(self, tag, inp, in_len, out_len):
        return self.ctx.hash_batch(tag, np.ascontiguousarray(inp).reshape(1, in_len, 4), in_len, out_len).reshape(out_len, 4)

    def hash_truncated(self, tag, inp, in_len, out_len):
        return self.ctx.hash_batch(tag, np.ascontiguousarray(inp).reshape(1, in_len, 4), in_len, out_len, truncated=True).reshape(out_len, 4)

    def encryption_tag(self, ln):
        return self.Enc.encryption_tag(ln)

    def encrypt(self, tag, message, secret, nonce):
        return self.Enc.encrypt_batch(message.reshape(1, -1, 4), secret.reshape(1, 2, 4), nonce.reshape(1, 4), ctx=self.ctx, tag=tag).reshape(-1, 4)

    def decrypt(self, tag, cipher, secret, nonce):
        back, ok = self.Enc.decrypt_batch(cipher.reshape(1, -1, 4), secret.reshape(1, 2, 4), nonce.reshape(1, 4), ctx=self.ctx, tag=tag)
        return back.reshape(-1, 4), bool(ok[0])


def check_fixture_file(path, backend):
    """AssertionError listing one line per difference (schema problems first)"""
    fx = json.load(open(path))
    bad = F.validate_schema(fx)
    assert not bad, "%s is not a schema-1 fixture file:\n%s" % (path, "\n".join(bad))
    diff = F.compare(fx, backend)
    if diff:  # the same again with the fixture's tags as inputs: says whether the sponge agrees once the tag is right
        rest = F.compare(fx, backend, use_fixture_tags=True)
        note = "with the fixture's own tags as inputs %d differences remain" % len(rest)
        raise AssertionError("%s differs from %s (%s) in %d places (%s):\n%s" % (backend.name, os.path.relpath(path, ROOT), fx["source"], len(diff), note, "\n".join(diff)))
    return fx


# ------------------------------------------------------------------------------------------------------------------ CPU
def test_prediction_file_is_well_formed_and_in_sync_with_the_oracle(oracle_mod):
    r = subprocess.run([sys.executable, os.path.join(HERE, "golden", "gen_reference_predicted.py"), "--check"], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    fx = check_fixture_file(PREDICTED, OracleBackend(oracle_mod))
    assert fx["source"] == "prediction"
    assert [r["name"] for r in fx["tags"]] == [F.shape_name(*s) for s in F.SHAPES] and [r["len"] for r in fx["encryption"]] == F.ENC_LENS
    # every shape on the seq and inv7pow families, the KAT family where the crate's ten inputs suffice
    assert len(fx["digests"]) == len(fx["truncated"]) == sum(2 + (sum(l) <= 10) for _, l, _ in F.SHAPES)
    # the chunked pattern hashes like the one-chunk one (adjacent absorbs aggregate, README.md:31-44) — in the PREDICTION; the reference file decides
    by = {(r["name"], r["input"]): r for r in fx["digests"]}
    assert by[("other_3+39_1", "seq")]["output_limbs"] == by[("other_42_1", "seq")]["output_limbs"]
    # the KAT family really is the crate's ten inputs
    kat = json.load(open(os.path.join(HERE, "golden", "hades_kat.json")))["inputs_le_hex"]
    assert [F.limbs_to_int(l) for l in by[("other_5_1", "kat")]["input_limbs"]] == [int.from_bytes(bytes.fromhex(h), "little") for h in kat[:5]]


def test_oracle_matches_the_reference_fixtures(oracle_mod):
    if not os.path.exists(REFERENCE):
        pytest.skip(ABSENT)
    fx = check_fixture_file(REFERENCE, OracleBackend(oracle_mod))
    assert fx["source"] == "reference", "tests/golden/reference_fixtures.json must come from bindings/rust/refgen (source: reference)"


def test_a_fixture_that_differs_in_one_tag_limb_fails_with_a_one_line_diff(oracle_mod, tmp_path):
    """what the first run with a real file looks like if the recollected tag recipe is wrong in one place: one line, naming it"""
    fx = json.load(open(PREDICTED))
    bad = copy.deepcopy(fx)
    bad["source"] = "reference"
    bad["tags"][0]["tag_limbs"][2] ^= 1
    p = tmp_path / "reference_fixtures.json"
    p.write_text(json.dumps(bad))
    with pytest.raises(AssertionError) as e:
        check_fixture_file(str(p), OracleBackend(oracle_mod))
    lines = str(e.value).splitlines()
    assert len(lines) == 2 and lines[1].startswith("tags.merkle4_4_1.tag_limbs: the fixture has ") and "in 1 places" in lines[0], lines
    # the file-to-file differ of the generator says the same
    r = subprocess.run([sys.executable, os.path.join(HERE, "golden", "gen_reference_predicted.py"), "--diff", str(p)], capture_output=True)
    out = r.stdout.decode().splitlines()
    assert r.returncode == 1 and len(out) == 1 and out[0].startswith("tags.merkle4_4_1.tag_limbs: the reference gives"), out
    # a wrong ciphertext element, a wrong truncation, a malformed file: each named
    bad = copy.deepcopy(fx)
    bad["encryption"][1]["cipher_limbs"][7][0] ^= 1
    bad["truncated"][3]["output_le_hex"][0] = "01" + bad["truncated"][3]["output_le_hex"][0][2:]
    p.write_text(json.dumps(bad))
    with pytest.raises(AssertionError) as e:
        check_fixture_file(str(p), OracleBackend(oracle_mod))
    lines = str(e.value).splitlines()[1:]
    assert len(lines) == 3 and lines[0].startswith("truncated.") and lines[1].startswith("encryption.21.cipher_limbs") and "element 7" in lines[1], lines
    assert lines[2].startswith("encryption.21: the oracle does not decrypt the fixture's cipher")  # (a tampered cipher no longer decrypts)
    del bad["digests"][0]["output_limbs"]
    p.write_text(json.dumps(bad))
    with pytest.raises(AssertionError, match="not a schema-1 fixture file"):
        check_fixture_file(str(p), OracleBackend(oracle_mod))
    # and an identical copy passes
    p.write_text(json.dumps(fx))
    check_fixture_file(str(p), OracleBackend(oracle_mod))
    r = subprocess.run([sys.executable, os.path.join(HERE, "golden", "gen_reference_predicted.py"), "--diff", str(p)], capture_output=True)
    assert r.returncode == 0 and b"PINNED" in r.stdout


def test_refgen_crate_needs_nothing_but_the_dusk_crates_and_writes_this_schema():
    """the generator cannot be compiled here; what can be checked is that it stays GPU-free and library-free (the whole point:
    run_parity.sh needs an MI355X and cargo on ONE machine, this needs cargo), and that its shapes and keys are the ones the tests read"""
    crate = os.path.join(ROOT, "bindings", "rust", "refgen")
    toml = open(os.path.join(crate, "Cargo.toml")).read()
    deps = re.findall(r"^([a-z0-9_-]+) *=", toml.split("[dependencies]")[1], re.M)
    assert sorted(deps) == ["dusk-bls12_381", "dusk-bytes", "dusk-jubjub", "dusk-poseidon", "dusk-safe"], deps
    assert not os.path.exists(os.path.join(crate, "build.rs")) and "build =" not in toml and "[workspace]" in toml
    src = open(os.path.join(crate, "src", "main.rs")).read()
    code = re.sub(r"//.*", "", src)
    assert "p252_" not in code and "poseidon252_hip" not in code and "extern \"C\"" not in code and "rand" not in toml
    dom = {"Merkle4": "merkle4", "Merkle2": "merkle2", "Other": "other"}
    shapes = [(dom[d], [int(x) for x in l.split(",")], int(o)) for d, l, o in re.findall(r"\(Domain::(\w+), vec!\[([0-9, ]+)\], (\d+)\)", code)]
    assert shapes == [(d, l, o) for d, l, o in F.SHAPES], shapes
    assert re.search(r"for len in \[2usize, 21, 42\]", code) and "JubJubScalar::from(12345u64)" in code and "0x6e6f6e6365u64" in code
    for sect, keys in F.KEYS.items():
        assert '("%s", &%s)' % (sect, sect) in code, sect
        for k in keys:
            assert '\\"%s\\":' % k in src, (sect, k)
    assert "hades_kat.json" in src and os.path.exists(os.path.join(crate, "..", "..", "..", "tests", "golden", "hades_kat.json"))
    run = open(os.path.join(ROOT, "bindings", "rust", "RUN.md")).read()
    assert "refgen" in run and "reference_fixtures.json" in run and "gen_reference_predicted.py --diff" in run


# ------------------------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_hip_library_reproduces_the_prediction(gpu_ctx):
    """the HIP library against the committed prediction: every tag (p252_tag / p252_encryption_tag), digest, truncated digest
    (the fused output stage), ciphertext, decryption and rejected nonce — so the prediction speaks for the product, not only the oracle"""
    check_fixture_file(PREDICTED, HipBackend(gpu_ctx))


@pytest.mark.gpu
def test_hip_library_matches_the_reference_fixtures(gpu_ctx):
    if not os.path.exists(REFERENCE):
        pytest.skip(ABSENT)
    fx = check_fixture_file(REFERENCE, HipBackend(gpu_ctx))
    assert fx["source"] == "reference"
