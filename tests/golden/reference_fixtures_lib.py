"""Shared by tests/golden/gen_reference_predicted.py and tests/test_reference_fixtures.py: the schema of the reference fixture file
(written by bindings/rust/refgen from the real dusk crates — VERDICT r5 item 1), its inputs, and the comparison that turns a fixture
file plus a backend (the CPU oracle, the HIP library) into a list of one-line differences.

The fixture file is DATA: inputs and the reference's outputs.  A scalar is the 4 little-endian u64 limbs of its Montgomery form."""
import hashlib

import numpy as np

P = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
R = 1 << 256
SEP = {"merkle4": 0xF, "merkle2": 0x3, "encryption": 1 << 32, "other": 0}  # From<Domain> for u64, src/hash.rs:38-56
DOMAIN_ID = {"merkle4": 0, "merkle2": 1, "encryption": 2, "other": 3}
# the io-patterns of bindings/rust/refgen/src/main.rs (tests/hash.rs:101-116, 188-203, 277-292 shapes, the Merkle domains, configs[3])
SHAPES = [("merkle4", [4], 1), ("merkle2", [2], 1), ("other", [3], 1), ("other", [5], 1), ("other", [15], 1), ("other", [42], 1),
          ("other", [3], 3), ("other", [5], 2), ("other", [4], 7), ("other", [42], 5), ("other", [3, 39], 1)]
ENC_LENS = [2, 21, 42]
ENC_SECRET_SCALAR = 12345
ENC_NONCE = 0x6e6f6e6365
KEYS = {
    "tags": ["name", "domain", "absorb_lens", "output_len", "tag_input", "tag_limbs"],
    "digests": ["name", "domain", "absorb_lens", "output_len", "input", "input_limbs", "output_limbs"],
    "truncated": ["name", "domain", "absorb_lens", "output_len", "input", "output_le_hex"],
    "encryption": ["len", "secret_limbs", "nonce_limbs", "message", "message_limbs", "cipher_limbs", "tag_input", "tag_limbs", "permutations",
                   "wrong_nonce_fails"],
}


def shape_name(domain, lens, out_len):
    return "%s_%s_%d" % (domain, "+".join(str(l) for l in lens), out_len)


def mont_limbs(v):
    m = v % P * R % P
    return [(m >> (64 * k)) & ((1 << 64) - 1) for k in range(4)]


def limbs_to_int(l):
    return sum(int(x) << (64 * k) for k, x in enumerate(l)) * pow(R, -1, P) % P


def family(name, n, kat_le_hex):
    """the three RNG-free input families of refgen, as canonical integers"""
    if name == "seq":
        return [j + 1 for j in range(n)]
    if name == "inv7pow":
        g = pow(7, -1, P)
        return [pow(g, j + 1, P) for j in range(n)]
    if name == "kat":
        return [int.from_bytes(bytes.fromhex(h), "little") for h in kat_le_hex[:n]]
    raise ValueError(name)


def jubjub_generator_times(k):
    """(u, v) of GENERATOR_EXTENDED * k on JubJub (-u^2 + v^2 = 1 + d u^2 v^2, d = -10240/10241 over the BLS12-381 scalar field).
    The generator is RECOLLECTED (v = 18 and the matching root; the curve equation is asserted below) — the comparison never depends on it:
    a fixture carries its own secret_limbs, and a wrong guess here only shows as a difference in that one field of the prediction."""
    d = -10240 * pow(10241, -1, P) % P
    gu, gv = 0x3fd2814c43ac65a6f1fbf02d0fd6cce62e3ebb21fd6c54ed4df7b7ffec7beaca, 18
    assert (-gu * gu + gv * gv) % P == (1 + d * gu * gu % P * gv * gv) % P

    def add(a, b):
        (u1, v1), (u2, v2) = a, b
        t = d * u1 * u2 % P * v1 * v2 % P
        return ((u1 * v2 + v1 * u2) * pow(1 + t, -1, P) % P, (v1 * v2 + u1 * u2) * pow(1 - t, -1, P) % P)
    acc, base = (0, 1), (gu, gv)
    while k:
        if k & 1:
            acc = add(acc, base)
        base = add(base, base)
        k >>= 1
    return acc


def recollected_tag_input(domain, lens, out_len):
    """dusk-safe's tag input AS RECOLLECTED (UNPINNED until a reference fixture confirms it): one big-endian u32 per aggregated call
    (top bit = absorb; adjacent absorbs aggregate, README.md:31-44), then the domain separator as a big-endian u64"""
    return b"".join(w.to_bytes(4, "big") for w in (0x80000000 | sum(lens), out_len)) + SEP[domain].to_bytes(8, "big")


def recollected_encryption_tag_input(ln):
    """dusk_safe::encrypt's io-pattern as recollected (STREAM): [Absorb(2), Absorb(1), Squeeze(len), Absorb(len), Squeeze(1)]"""
    return b"".join(w.to_bytes(4, "big") for w in (0x80000003, ln, 0x80000000 | ln, 1)) + SEP["encryption"].to_bytes(8, "big")


def recollected_hash_to_scalar(data):
    """BlsScalar::hash_to_scalar as recollected: BLAKE2b-512 read as a 512-bit little-endian integer mod p"""
    return mont_limbs(int.from_bytes(hashlib.blake2b(data, digest_size=64).digest(), "little") % P)


def validate_schema(fx):
    """list of problems (empty = a well-formed fixture file of schema 1)"""
    bad = []
    if fx.get("schema") != 1 or fx.get("source") not in ("reference", "prediction"):
        bad.append("schema / source: want schema 1 and source reference | prediction, got %r / %r" % (fx.get("schema"), fx.get("source")))
    for sect, keys in KEYS.items():
        rows = fx.get(sect)
        if not isinstance(rows, list) or not rows:
            bad.append("%s: missing or empty" % sect)
            continue
        for i, row in enumerate(rows):
            miss = [k for k in keys if k not in row]
            if miss:
                bad.append("%s[%d]: missing %s" % (sect, i, miss))
    if bad:
        return bad
    def is_scalar(l):
        return isinstance(l, list) and len(l) == 4 and all(isinstance(x, int) and 0 <= x < 1 << 64 for x in l) and sum(x << (64 * k) for k, x in enumerate(l)) < P
    for i, r in enumerate(fx["tags"]):
        if not is_scalar(r["tag_limbs"]) or not all(isinstance(b, int) and 0 <= b < 256 for b in r["tag_input"]):
            bad.append("tags[%d] (%s): tag_limbs / tag_input malformed" % (i, r["name"]))
    for i, r in enumerate(fx["digests"]):
        if len(r["input_limbs"]) != sum(r["absorb_lens"]) or len(r["output_limbs"]) != r["output_len"] or not all(map(is_scalar, r["input_limbs"] + r["output_limbs"])):
            bad.append("digests[%d] (%s / %s): lengths or scalars malformed" % (i, r["name"], r["input"]))
    for i, r in enumerate(fx["truncated"]):
        if len(r["output_le_hex"]) != r["output_len"] or not all(len(h) == 64 and int.from_bytes(bytes.fromhex(h), "little") < 1 << 250 for h in r["output_le_hex"]):
            bad.append("truncated[%d] (%s / %s): not output_len values below 2^250" % (i, r["name"], r["input"]))
    for i, r in enumerate(fx["encryption"]):
        if len(r["message_limbs"]) != r["len"] or len(r["cipher_limbs"]) != r["len"] + 1 or len(r["secret_limbs"]) != 2 or \
                not all(map(is_scalar, r["message_limbs"] + r["cipher_limbs"] + r["secret_limbs"] + [r["nonce_limbs"], r["tag_limbs"]])):
            bad.append("encryption[%d] (len %s): lengths or scalars malformed" % (i, r["len"]))
    return bad


class Backend:
    """what a fixture file is compared with.  Every method takes / returns uint64 numpy arrays of Montgomery limbs."""
    name = "?"

    def tag(self, domain, lens, out_len):  # -> (4,)
        raise NotImplementedError

    def tag_input(self, domain, lens, out_len):  # -> bytes, or None when the backend does not expose the bytes it hashes
        return recollected_tag_input(domain, lens, out_len)

    def hash(self, tag, inp, in_len, out_len):  # inp (in_len, 4) -> (out_len, 4)
        raise NotImplementedError

    def hash_truncated(self, tag, inp, in_len, out_len):  # -> (out_len, 4) raw limbs below 2^250
        raise NotImplementedError

    def encryption_tag(self, ln):
        raise NotImplementedError

    def encrypt(self, tag, message, secret, nonce):  # (len,4), (2,4), (4,) -> (len+1, 4)
        raise NotImplementedError

    def decrypt(self, tag, cipher, secret, nonce):  # -> (message (len,4), ok bool)
        raise NotImplementedError


def _arr(l):
    return np.array(l, dtype=np.uint64).reshape(-1, 4)


def compare(fx, be, use_fixture_tags=False):
    """one line per difference between the fixture file's outputs and backend `be` computed on the fixture's own inputs.
    use_fixture_tags: hash with the fixture's tag instead of the backend's (separates "the tag recipe is wrong" from "the sponge is
    wrong": with the reference's tag as input, digests must match even if the backend's own tag recipe does not)."""
    out = []
    ref_tag = {}
    for r in fx["tags"]:
        ref_tag[r["name"]] = np.array(r["tag_limbs"], dtype=np.uint64)
        got_in = be.tag_input(r["domain"], r["absorb_lens"], r["output_len"])
        if got_in is not None and list(got_in) != r["tag_input"]:
            out.append("tags.%s.tag_input: the fixture has %s, %s hashes %s" % (r["name"], bytes(r["tag_input"]).hex(), be.name, bytes(got_in).hex()))
        got = [int(x) for x in be.tag(r["domain"], r["absorb_lens"], r["output_len"])]
        if got != r["tag_limbs"]:
            out.append("tags.%s.tag_limbs: the fixture has %s, %s gives %s" % (r["name"], r["tag_limbs"], be.name, got))
    for r in fx["digests"]:
        tag = ref_tag[r["name"]] if use_fixture_tags and r["name"] in ref_tag else be.tag(r["domain"], r["absorb_lens"], r["output_len"])
        got = be.hash(tag, _arr(r["input_limbs"]), sum(r["absorb_lens"]), r["output_len"])
        if [[int(x) for x in row] for row in got] != r["output_limbs"]:
            out.append("digests.%s.%s: %s differs from the fixture (first output %s, fixture %s)" % (r["name"], r["input"], be.name, [int(x) for x in got[0]], r["output_limbs"][0]))
    dig = {(r["name"], r["input"]): r for r in fx["digests"]}
    for r in fx["truncated"]:
        src = dig.get((r["name"], r["input"]))
        if src is None:
            out.append("truncated.%s.%s: no digest row with these inputs" % (r["name"], r["input"]))
            continue
        tag = ref_tag[r["name"]] if use_fixture_tags and r["name"] in ref_tag else be.tag(r["domain"], r["absorb_lens"], r["output_len"])
        got = be.hash_truncated(tag, _arr(src["input_limbs"]), sum(r["absorb_lens"]), r["output_len"])
        got_hex = [b"".join(int(x).to_bytes(8, "little") for x in row).hex() for row in got]
        if got_hex != r["output_le_hex"]:
            out.append("truncated.%s.%s: %s gives %s, the fixture %s" % (r["name"], r["input"], be.name, got_hex[0], r["output_le_hex"][0]))
    for r in fx["encryption"]:
        ln = r["len"]
        got_tag = [int(x) for x in be.encryption_tag(ln)]
        if got_tag != r["tag_limbs"]:
            out.append("encryption.%d.tag_limbs: the fixture has %s, %s gives %s" % (ln, r["tag_limbs"], be.name, got_tag))
        if r["tag_input"] != list(recollected_encryption_tag_input(ln)):
            out.append("encryption.%d.tag_input: the fixture's io-pattern bytes %s are not the recollected STREAM pattern %s" %
                       (ln, bytes(r["tag_input"]).hex(), recollected_encryption_tag_input(ln).hex()))
        tag = np.array(r["tag_limbs"], dtype=np.uint64) if use_fixture_tags else be.encryption_tag(ln)
        secret, nonce = _arr(r["secret_limbs"]), np.array(r["nonce_limbs"], dtype=np.uint64)
        got = be.encrypt(tag, _arr(r["message_limbs"]), secret, nonce)
        if [[int(x) for x in row] for row in got] != r["cipher_limbs"]:
            first = next(i for i, row in enumerate(got) if [int(x) for x in row] != r["cipher_limbs"][i])
            out.append("encryption.%d.cipher_limbs: %s differs from the fixture from element %d on" % (ln, be.name, first))
        back, ok = be.decrypt(tag, _arr(r["cipher_limbs"]), secret, nonce)
        if not ok or [[int(x) for x in row] for row in back] != r["message_limbs"]:
            out.append("encryption.%d: %s does not decrypt the fixture's cipher to its message" % (ln, be.name))
        wrong = np.array(mont_limbs(limbs_to_int(r["nonce_limbs"]) + 1), dtype=np.uint64)
        _, ok2 = be.decrypt(tag, _arr(r["cipher_limbs"]), secret, wrong)
        if bool(ok2) == bool(r["wrong_nonce_fails"]):
            out.append("encryption.%d.wrong_nonce_fails: the fixture says %s, %s %s" % (ln, r["wrong_nonce_fails"], be.name, "accepts the wrong nonce" if ok2 else "rejects it"))
        if r["permutations"] != -(-ln // 4) + (ln - 1) // 4 + 1:
            out.append("encryption.%d.permutations: the fixture counts %d, the STREAM construction makes %d" % (ln, r["permutations"], -(-ln // 4) + (ln - 1) // 4 + 1))
    return out
