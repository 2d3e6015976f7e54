"""One-off heavier parity soak on the GPU box (developer tool, not part of the test-suite): 2^18 full-state
permutations and a few thousand random sponge shapes against the oracle."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import oracle
import poseidon252_amd as P

ctx = P.Context(0)
t0 = time.time()
st = oracle.fill_random(77, 5 * (1 << 18)).reshape(-1, 5, 4)
got = ctx.permute_batch(st)
exp = oracle.permute_batch(st)
assert np.array_equal(got, exp), "permute mismatch"
print("2^18 permutations (all 5 lanes) bit-exact, %.1f s" % (time.time() - t0))
rng = np.random.default_rng(5)
checked = 0
t0 = time.time()
while time.time() - t0 < 40:
    in_len, out_len, n = int(rng.integers(1, 60)), int(rng.integers(1, 12)), int(rng.integers(1, 700))
    tag = oracle.fill_random(int(rng.integers(1, 1 << 30)), 1).reshape(4)
    msg = oracle.fill_random(int(rng.integers(1, 1 << 30)), n * in_len).reshape(n, in_len, 4)
    assert np.array_equal(ctx.hash_batch(tag, msg, in_len, out_len), oracle.hash_batch(tag, msg, in_len, out_len, threads=8)), (in_len, out_len, n)
    checked += n
print("%d random-shape sponge messages bit-exact" % checked)
# Merkle digests through the cooperative kernels (<= 8,192 nodes: 8 lanes per node; <= 16,384: 4 lanes) and across the
# switches to the one-lane kernels
checked = 0
t0 = time.time()
mtag = oracle.tag(0, [4], 1)
while time.time() - t0 < 25:
    n = int(rng.choice([int(rng.integers(1, 600)), int(rng.integers(600, 8193)), int(rng.integers(8193, 16385)), int(rng.integers(16385, 70000))]))
    x = oracle.fill_random(int(rng.integers(1, 1 << 30)), 4 * n).reshape(n, 4, 4)
    assert np.array_equal(ctx.hash_batch(mtag, x, 4, 1), oracle.hash_batch(mtag, x, 4, 1, threads=8)), n
    checked += n
print("%d Merkle4 digests in batches of 1 .. 70,000 bit-exact" % checked)
