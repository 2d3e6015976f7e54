"""The low-latency digest kernels (k_merkle4_coop<8> / <4>: one node per group of eight / four lanes, coop29.hpp) — what
every launch of at most 8,192 / 16,384 Merkle digests runs — against the oracle: batch sizes around a group / a wave /
a block / the switches between the kernels, ragged children, arity 2, saturated limbs, and byte equality with the
one-lane kernels."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("n", [1, 2, 7, 8, 9, 31, 32, 33, 255, 256, 257, 8191, 8192, 8193, 12345, 16383, 16384, 16385])
def test_digest_batches_around_the_group_and_switch_sizes(gpu_ctx, oracle_mod, n):
    tag = oracle_mod.tag(0, [4], 1)
    x = oracle_mod.fill_random(50 + n, 4 * n).reshape(n, 4, 4)
    assert np.array_equal(gpu_ctx.hash_batch(tag, x, 4, 1), oracle_mod.hash_batch(tag, x, 4, 1))
    tag2 = oracle_mod.tag(1, [2], 1)  # Merkle2 node: two absorbed elements, the other two lanes start at zero
    assert np.array_equal(gpu_ctx.hash_batch(tag2, x[:, :2], 2, 1), oracle_mod.hash_batch(tag2, np.ascontiguousarray(x[:, :2]), 2, 1))


@pytest.mark.parametrize("n_items", [600, 9000])  # the 8-lane and the 4-lane kernel
def test_saturated_limbs_and_edge_values(gpu_ctx, oracle_mod, n_items):
    P = oracle_mod.P
    pats = [(1 << 256) - 1, (1 << 255) + 12345, P, P + 1, 2 * P - 1, int("55" * 32, 16), int("aa" * 32, 16), 0, 1, P - 1,
            sum(((1 << 29) - 1) << (29 * i) for i in range(8)) | (((1 << 24) - 1) << 232), (1 << 256) - (1 << 200)]
    rng = np.random.default_rng(3)
    pick = rng.integers(0, len(pats), size=(n_items, 4))
    raw = np.array([[oracle_mod.int_to_limbs(pats[k]) for k in row] for row in pick], dtype=np.uint64)
    red = np.array([[oracle_mod.int_to_limbs(pats[k] % P) for k in row] for row in pick], dtype=np.uint64)
    for tag in (oracle_mod.tag(0, [4], 1), np.array(oracle_mod.int_to_limbs((1 << 256) - 5), dtype=np.uint64)):
        tag_red = np.array(oracle_mod.int_to_limbs(int.from_bytes(np.asarray(tag, dtype=np.uint64).tobytes(), "little") % P), dtype=np.uint64)
        got = gpu_ctx.hash_batch(tag, raw, 4, 1)
        assert np.array_equal(got, oracle_mod.hash_batch(tag_red, red, 4, 1))


@pytest.mark.parametrize("n_leaves", [6, 13, 4 ** 6, 4 ** 6 + 1, 3 * 4 ** 6 + 5, 4 ** 7 + 3, 4 ** 8, 50001])
def test_trees_whose_levels_run_on_the_cooperative_kernel(gpu_ctx, oracle_mod, n_leaves):
    tag = oracle_mod.tag(0, [4], 1)
    lv = oracle_mod.fill_random(n_leaves, n_leaves)
    root, levels = gpu_ctx.merkle4_tree(tag, lv, want_levels=True)
    o_root, o_levels, _ = oracle_mod.merkle4_tree(tag, lv, want_levels=True)
    assert np.array_equal(root, o_root) and np.array_equal(levels, o_levels)


@pytest.mark.parametrize("n", [1, 7, 8, 9, 300, 8191, 8192, 8193])
def test_permute_sponge_and_openings_on_lane_groups(gpu_ctx, oracle_mod, n):
    """batches of <= 8,192 run k_permute_coop / k_sponge_coop / k_merkle4_path_coop (lane i of a group holds state element
    i); 8,193 is the first size on the one-lane kernels"""
    st = oracle_mod.fill_random(11 + n, 5 * n).reshape(n, 5, 4)
    assert np.array_equal(gpu_ctx.permute_batch(st), oracle_mod.permute_batch(st))
    tag = oracle_mod.fill_random(3, 1).reshape(4)
    m = min(n, 1200)  # (the oracle is the slow side)
    for in_len, out_len in ((1, 1), (3, 3), (4, 7), (5, 2), (9, 6), (42, 5)):
        msg = oracle_mod.fill_random(100 * in_len + out_len + n, m * in_len).reshape(m, in_len, 4)
        assert np.array_equal(gpu_ctx.hash_batch(tag, msg, in_len, out_len), oracle_mod.hash_batch(tag, msg, in_len, out_len, threads=8)), (in_len, out_len)
    mtag = oracle_mod.tag(0, [4], 1)
    rng = np.random.default_rng(n)
    for depth in (0, 1, 5, 12):
        leaves = oracle_mod.fill_random(7 * n + depth, m)
        sib = oracle_mod.fill_random(9 * n + depth, m * depth * 3).reshape(m, depth, 3, 4) if depth else np.zeros((m, 0, 3, 4), dtype=np.uint64)
        pos = rng.integers(0, 4, size=(m, depth), dtype=np.uint8)
        assert np.array_equal(gpu_ctx.merkle4_path_batch(mtag, leaves, sib, pos), oracle_mod.merkle4_path_batch(mtag, leaves, sib, pos)), depth


@pytest.mark.parametrize("n", [1, 9, 700, 8192, 8193])
@pytest.mark.parametrize("length", [1, 2, 4, 5, 21, 42])
def test_encryption_on_lane_groups(gpu_ctx, oracle_mod, n, length):
    """k_crypt_coop (<= 8,192 messages): ciphers equal the oracle's under both call sequences, decryption restores the
    messages (a plaintext element decrypted by one lane is absorbed by another), a flipped limb fails the MAC"""
    from poseidon252_amd import encryption as E
    m = min(n, 400 if length > 5 else n)
    msg = oracle_mod.fill_random(length * 1000 + n, m * length).reshape(m, length, 4)
    sec = oracle_mod.fill_random(length * 1000 + n + 1, m * 2).reshape(m, 2, 4)
    non = oracle_mod.fill_random(length * 1000 + n + 2, m)
    for variant in (E.STREAM, E.DUPLEX):
        c = E.encrypt_batch(msg, sec, non, ctx=gpu_ctx, variant=variant)
        assert np.array_equal(c, oracle_mod.encrypt_batch(E.encryption_tag(length, variant), msg, sec, non, variant=variant))
        back, ok = E.decrypt_batch(c, sec, non, ctx=gpu_ctx, variant=variant)
        assert ok.all() and np.array_equal(back, msg)
        bad = c.copy()
        bad[m - 1, length // 2, 0] ^= np.uint64(1)
        _, ok2 = E.decrypt_batch(bad, sec, non, ctx=gpu_ctx, variant=variant)
        assert not ok2[m - 1] and ok2[:m - 1].all()


def test_one_lane_kernels_still_pass_the_parity_suite():
    """the parity suite's batches are small: by default they run on the lane-group kernels (the reference's six KAT digests
    through k_sponge_coop, for one).  With P252_COOP_MAX_NODES=0 the same suite runs on the one-lane kernels again."""
    env = dict(os.environ, P252_COOP_MAX_NODES="0")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(HERE, "test_gpu_parity.py"), os.path.join(HERE, "test_next_rows.py"), os.path.join(HERE, "test_encryption.py"),
                        "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider"], cwd=os.path.dirname(HERE), env=env, capture_output=True, timeout=900)
    assert r.returncode == 0, r.stdout.decode()[-3000:] + r.stderr.decode()[-2000:]


def test_same_bytes_as_the_one_lane_kernels():
    """P252_COOP_MAX_NODES=0 (read once per process, hence the subprocesses) sends every launch to the one-lane kernels:
    digests and tree levels must be the same bytes either way"""
    code = r'''
import hashlib, numpy as np, oracle, poseidon252_amd as P
ctx = P.Context(0)
tag = P.merkle4_tag()
h = hashlib.sha256()
for n in (1, 9, 300, 8192, 12000):
    x = oracle.fill_random(70 + n, 4 * n).reshape(n, 4, 4)
    h.update(ctx.hash_batch(tag, x, 4, 1).tobytes())
root, levels = P.merkle4_tree(oracle.fill_random(9, 70001), tag=tag, ctx=ctx, want_levels=True)
h.update(root.tobytes()); h.update(levels.tobytes())
st = oracle.fill_random(5, 5 * 3000).reshape(3000, 5, 4)
h.update(ctx.permute_batch(st).tobytes())
h.update(ctx.hash_batch(st[0, 0], st.reshape(-1, 4)[:42 * 300].reshape(300, 42, 4), 42, 5).tobytes())
pos = (np.arange(2000 * 6) % 4).astype(np.uint8).reshape(2000, 6)
h.update(ctx.merkle4_path_batch(tag, st[:2000, 0], oracle.fill_random(8, 2000 * 18).reshape(2000, 6, 3, 4), pos).tobytes())
print("DIGEST", h.hexdigest())
'''
    outs = []
    for coop in ("16384", "0"):
        env = dict(os.environ, P252_COOP_MAX_NODES=coop)
        r = subprocess.run([sys.executable, "-c", code], cwd=os.path.dirname(HERE), env=env, capture_output=True, timeout=600)
        assert r.returncode == 0, r.stdout.decode() + r.stderr.decode()
        outs.append([l for l in r.stdout.decode().splitlines() if l.startswith("DIGEST")][0])
    assert outs[0] == outs[1]
