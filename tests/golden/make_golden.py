#!/usr/bin/env python3
"""Generates tests/golden/vectors.json: seeded inputs -> outputs of the KAT-pinned CPU oracle.

The reference (Rust) cannot be built or imported in this image (no cargo/rustc; dusk-bls12_381 and
dusk-safe are un-vendored), so these vectors come from oracle/ AFTER it reproduced the reference's
six known-answer digests (tests/test_oracle_kat.py).  They freeze today's behaviour so that a later
change to oracle AND kernels cannot drift together unnoticed.  Run from the repo root:
    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402


def hx(a):
    return ["%016x" % int(v) for v in np.asarray(a, dtype=np.uint64).reshape(-1)]


def main():
    out = {"_doc": "limbs are hex u64, 4 per BlsScalar (Montgomery form, little-endian limb order); "
                   "inputs = oracle.fill_random(seed, count) (splitmix64, BASELINE.md §2)"}
    # permutations
    st = oracle.fill_random(0x1001, 5 * 8)
    out["permute"] = {"seed": 0x1001, "n": 8, "out": hx(oracle.permute_batch(st.reshape(8, 5, 4)))}
    # special states
    special = np.zeros((3, 5, 4), dtype=np.uint64)
    special[1] = np.stack([oracle.mont_from_int(17)] * 5)
    special[2] = np.stack([oracle.mont_from_int(i) for i in range(5)])
    out["permute_special"] = {"states": ["0,0,0,0,0", "17 x5", "0,1,2,3,4"], "out": hx(oracle.permute_batch(special))}
    # hashes: (domain, in_len, out_len)
    cases = []
    for dom, i, o in [(0, 4, 1), (1, 2, 1), (3, 3, 1), (3, 5, 1), (3, 15, 1), (3, 3, 3), (3, 5, 2), (3, 4, 7), (3, 42, 5), (3, 42, 1), (3, 1, 1), (2, 6, 1)]:
        tag = oracle.tag(dom, [i], o)
        seed = 0x2000 + 64 * i + o
        m = oracle.fill_random(seed, 4 * i)
        cases.append({"domain": dom, "in_len": i, "out_len": o, "seed": seed, "n": 4, "tag_UNPINNED": hx(tag),
                      "out": hx(oracle.hash_batch(tag, m.reshape(4, i, 4), i, o))})
    out["hash"] = cases
    # trees
    trees = []
    tag = oracle.tag(0, [4], 1)
    for n in [1, 3, 4, 5, 16, 21, 64, 257]:
        lv = oracle.fill_random(0x3000 + n, n)
        root, perms = oracle.merkle4_tree(tag, lv)
        trees.append({"n_leaves": n, "seed": 0x3000 + n, "root": hx(root), "perms": int(perms)})
    out["merkle4_tree"] = trees
    # encryption (src/encryption.rs:62-95): both candidate call sequences (oracle/p252_oracle.h), message lengths incl. the
    # reference tests' 21 and 42 (tests/encryption.rs:33,49); UNPINNED against the real dusk-safe
    enc = []
    for variant in (0, 1):
        for ln in (2, 5, 21, 42):
            seed = 0x4000 + 100 * variant + ln
            tag = oracle.encryption_tag(ln, variant)
            msgs = oracle.fill_random(seed, 2 * ln).reshape(2, ln, 4)
            secrets = oracle.fill_random(seed + 1, 4).reshape(2, 2, 4)
            nonces = oracle.fill_random(seed + 2, 2)
            enc.append({"variant": variant, "len": ln, "seed": seed, "n": 2, "tag_UNPINNED": hx(tag),
                        "cipher": hx(oracle.encrypt_batch(tag, msgs, secrets, nonces, variant))})
    out["encrypt_UNPINNED"] = enc
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vectors.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
