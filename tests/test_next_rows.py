"""SURVEY §8(f) "next" rows: truncated outputs (hash.rs:164-183) on the device, and batched Merkle
openings (branch re-hash, the poseidon-merkle use named in AGENTS.md:62-66)."""
import numpy as np
import pytest


# ------------------------------------------------------------------ CPU: oracle + host bookkeeping
def test_oracle_openings_reproduce_root(oracle_mod):
    """an opening extracted from the built tree re-hashes to the tree's root (oracle only, no GPU)"""
    from poseidon252_amd.merkle import merkle4_openings
    tag = oracle_mod.tag(0, [4], 1)
    for n in (1, 4, 5, 16, 21, 64, 257):
        lv = oracle_mod.fill_random(0x900 + n, n)
        root, levels, _ = oracle_mod.merkle4_tree(tag, lv, want_levels=True)
        idx = sorted(set([0, n - 1, n // 2, n // 3]))
        sib, pos = merkle4_openings(lv, levels, idx)
        roots = oracle_mod.merkle4_path_batch(tag, lv[idx], sib, pos)
        assert all(np.array_equal(r, root) for r in roots), n
        if n > 1:  # a tampered leaf or a wrong position must not verify
            bad = lv[idx].copy()
            bad[0, 0] ^= np.uint64(1)
            assert not np.array_equal(oracle_mod.merkle4_path_batch(tag, bad, sib, pos)[0], root)
            pos2 = pos.copy()
            pos2[0, 0] = (pos2[0, 0] + 1) & 3
            assert not np.array_equal(oracle_mod.merkle4_path_batch(tag, lv[idx], sib, pos2)[0], root)


def test_oracle_path_rejects_bad_position(oracle_mod):
    z = np.zeros((1, 4), dtype=np.uint64)
    with pytest.raises(ValueError):
        oracle_mod.merkle4_path_batch(z[0], z, np.zeros((1, 1, 3, 4), dtype=np.uint64), np.array([[4]], dtype=np.uint8))


def test_oracle_merkle2_tree_is_composition_of_merkle2_digests(oracle_mod):
    tag = oracle_mod.tag(1, [2], 1)  # Domain::Merkle2, [Absorb(2), Squeeze(1)], separator 0x3
    lv = oracle_mod.fill_random(0x222, 5)  # 5 -> 3 (padded) -> 2 (padded) -> 1
    root, levels, perms = oracle_mod.merkle2_tree(tag, lv, want_levels=True)
    assert perms == 3 + 2 + 1 and levels.shape[0] == 6
    pad = np.zeros((6, 4), dtype=np.uint64)
    pad[:5] = lv
    l1 = oracle_mod.hash_batch(tag, pad.reshape(3, 2, 4), 2, 1).reshape(3, 4)
    assert np.array_equal(levels[:3], l1)
    pad2 = np.zeros((4, 4), dtype=np.uint64)
    pad2[:3] = l1
    l2 = oracle_mod.hash_batch(tag, pad2.reshape(2, 2, 4), 2, 1).reshape(2, 4)
    assert np.array_equal(root, oracle_mod.hash_batch(tag, l2.reshape(1, 2, 4), 2, 1).reshape(4))


# ------------------------------------------------------------------ GPU parity
@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2, 3, 5, 8, 33, 1000, 4097])
def test_merkle2_tree_on_gpu(gpu_ctx, oracle_mod, n):
    """Domain::Merkle2 (hash.rs:27-31): arity-2 tree, and Merkle2-shaped digests through hash_batch"""
    import poseidon252_amd as P
    tag = P.compute_tag(P.Domain.Merkle2, [2], 1)
    lv = oracle_mod.fill_random(0x2000 + n, n)
    root, levels = gpu_ctx.merkle2_tree(tag, lv, want_levels=True)
    oroot, olevels, _ = oracle_mod.merkle2_tree(tag, lv, want_levels=True)
    assert np.array_equal(root, oroot) and np.array_equal(levels, olevels)
    assert np.array_equal(gpu_ctx.merkle2_tree(tag, lv), oroot)
    if n >= 2:
        pairs = lv[: 2 * (n // 2)].reshape(-1, 2, 4)
        hb = P.HashBatch(P.Domain.Merkle2, 2, ctx=gpu_ctx)
        assert np.array_equal(hb.digest(pairs), oracle_mod.hash_batch(tag, pairs, 2, 1))


@pytest.mark.gpu
def test_truncate250_device_matches_reference_rule(gpu_ctx, oracle_mod):
    import torch
    P = oracle_mod.P
    vals = [0, 1, P - 1, (1 << 250) - 1, 1 << 250, (1 << 250) + 5, (1 << 254), pow(2, -256, P)]
    x = np.concatenate([np.stack([oracle_mod.mont_from_int(v) for v in vals]), oracle_mod.fill_random(0x250, 5000)])
    d = torch.from_numpy(x.view(np.int64)).cuda()
    out = torch.empty_like(d)
    gpu_ctx.truncate250_device(d, out, x.shape[0])
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint64)
    exp = np.stack([oracle_mod.truncate250(v) for v in x])
    assert np.array_equal(got, exp)
    # definition check against big ints: canonical value & (2^250 - 1)
    for v, g in zip(vals, got):
        assert oracle_mod.limbs_to_int(g) == (v % P) & ((1 << 250) - 1)
    gpu_ctx.truncate250_device(d, d, x.shape[0])  # in place
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy().view(np.uint64), exp)


@pytest.mark.gpu
def test_digest_truncated_batch_like_reference(gpu_ctx, oracle_mod):
    """tests/hash.rs:188-203 shape (5 inputs, truncated) — batched, host and device buffers"""
    import torch
    import poseidon252_amd as P
    hb = P.HashBatch(P.Domain.Other, 5, ctx=gpu_ctx)
    m = oracle_mod.fill_random(0xbeef, 5 * 300).reshape(300, 5, 4)
    exp = np.stack([oracle_mod.truncate250(v) for v in oracle_mod.hash_batch(hb.tag, m, 5, 1).reshape(-1, 4)]).reshape(300, 1, 4)
    assert np.array_equal(hb.digest_truncated(m), exp)
    d = hb.digest_truncated(torch.from_numpy(m.view(np.int64)).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy().view(np.uint64), exp)
    single = P.Hash.digest_truncated(P.Domain.Other, m[0], ctx=gpu_ctx)
    assert np.array_equal(single, exp[0])


@pytest.mark.gpu
@pytest.mark.parametrize("n_leaves", [1, 4, 21, 256, 4099])
def test_merkle_openings_on_gpu(gpu_ctx, oracle_mod, n_leaves):
    from poseidon252_amd.merkle import merkle4_openings, merkle4_path_roots
    tag = oracle_mod.tag(0, [4], 1)
    lv = oracle_mod.fill_random(0xa00 + n_leaves, n_leaves)
    root, levels = gpu_ctx.merkle4_tree(tag, lv, want_levels=True)
    rng = np.random.default_rng(n_leaves)
    idx = rng.integers(0, n_leaves, size=min(n_leaves, 700))
    sib, pos = merkle4_openings(lv, levels, idx)
    roots = merkle4_path_roots(lv[idx], sib, pos, tag=tag, ctx=gpu_ctx)
    assert np.array_equal(roots, oracle_mod.merkle4_path_batch(tag, lv[idx], sib, pos))
    assert all(np.array_equal(r, root) for r in roots)


@pytest.mark.gpu
def test_merkle_path_random_and_device_api(gpu_ctx, oracle_mod):
    """arbitrary (not tree-derived) paths, depth 0..12, host and device entry points"""
    import torch
    import poseidon252_amd as P
    tag = oracle_mod.fill_random(5, 1)[0]
    for depth in (0, 1, 3, 12):
        n = 513
        leaves = oracle_mod.fill_random(100 + depth, n)
        sib = oracle_mod.fill_random(200 + depth, n * depth * 3).reshape(n, depth, 3, 4) if depth else np.zeros((n, 0, 3, 4), dtype=np.uint64)
        pos = np.random.default_rng(depth).integers(0, 4, size=(n, depth)).astype(np.uint8)
        exp = oracle_mod.merkle4_path_batch(tag, leaves, sib, pos)
        assert np.array_equal(gpu_ctx.merkle4_path_batch(tag, leaves, sib, pos), exp)
        if depth:
            d_l = torch.from_numpy(leaves.view(np.int64)).cuda()
            d_s = torch.from_numpy(sib.view(np.int64)).cuda()
            d_p = torch.from_numpy(pos).cuda()
            d_r = torch.empty((n, 4), dtype=torch.int64, device="cuda")
            gpu_ctx.merkle4_path_batch_device(tag, d_l, d_s, d_p, depth, d_r, n)
            torch.cuda.synchronize()
            assert np.array_equal(d_r.cpu().numpy().view(np.uint64), exp)
    with pytest.raises(ValueError):
        gpu_ctx.merkle4_path_batch(tag, leaves[:1], np.zeros((1, 1, 3, 4), dtype=np.uint64), np.array([[7]], dtype=np.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("n_leaves,k", [(1, 1), (5, 2), (4 ** 6, 1), (4 ** 6, 300), (4 ** 6 + 3, 4 ** 6 + 3), (70001, 9000), (70001, 7), (4 ** 8, 20000)])
def test_incremental_tree_update(gpu_ctx, oracle_mod, n_leaves, k):
    """p252_merkle4_update_device: k leaves of a stored tree change; leaves, every level and the root must
    equal a fresh build over the updated leaves (oracle).  k <= 8,192 runs the lane-group kernel, more the one-lane one;
    updates cluster under shared parents and touch the ragged last node."""
    import torch
    tag = oracle_mod.tag(0, [4], 1)
    leaves = oracle_mod.fill_random(3 * n_leaves + k, n_leaves)
    total = oracle_mod.levels_total(n_leaves)
    d_leaves = torch.from_numpy(leaves.view(np.int64).copy()).cuda()
    d_levels = torch.zeros((max(total, 1), 4), dtype=torch.int64, device="cuda")
    d_root = torch.zeros(4, dtype=torch.int64, device="cuda")
    gpu_ctx.merkle4_tree_device(tag, d_leaves, n_leaves, d_root, d_levels)
    rng = np.random.default_rng(n_leaves + k)
    idx = rng.permutation(n_leaves)[:k].astype(np.int64)
    if k >= 4 and n_leaves >= 8:
        idx[:4] = [n_leaves - 1, n_leaves - 2, 0, 1]  # the ragged last node and one shared parent
        idx = np.unique(idx)
        k = idx.shape[0]
    new = oracle_mod.fill_random(99 + k, k)
    d_idx = torch.from_numpy(idx.astype(np.int32)).cuda()
    d_new = torch.from_numpy(new.view(np.int64).copy()).cuda()
    gpu_ctx.merkle4_update_device(tag, d_leaves, n_leaves, d_levels, d_idx, d_new, k, d_root)
    torch.cuda.synchronize()
    updated = leaves.copy()
    updated[idx] = new
    o_root, o_levels, _ = oracle_mod.merkle4_tree(tag, updated, want_levels=True)
    assert np.array_equal(d_leaves.cpu().numpy().view(np.uint64), updated)
    assert np.array_equal(d_levels.cpu().numpy().view(np.uint64)[:total], o_levels)
    assert np.array_equal(d_root.cpu().numpy().view(np.uint64), o_root)


@pytest.mark.gpu
@pytest.mark.parametrize("k_total", [40, 20000])  # the lane-group kernel (k <= 8,192) and the one-lane kernel
def test_incremental_update_skips_out_of_range_positions(gpu_ctx, oracle_mod, k_total):
    """a position >= n_leaves used to be undefined behaviour (a store outside the leaves).  It is skipped now;
    p252_merkle4_update_checked_device also counts the skipped positions; the Python wrapper's check=True raises before
    launching.  The tree afterwards equals a fresh build over the in-range updates only."""
    import ctypes
    import torch
    from poseidon252_amd import _lib
    n_leaves = 5003
    tag = oracle_mod.tag(0, [4], 1)
    leaves = oracle_mod.fill_random(11, n_leaves)
    total = oracle_mod.levels_total(n_leaves)
    # guard scalars behind the leaves: an out-of-range store would land there
    d_buf = torch.from_numpy(np.concatenate([leaves, np.full((64, 4), 0x5A5A5A5A5A5A5A5A, dtype=np.uint64)]).view(np.int64).copy()).cuda()
    d_levels = torch.zeros((total, 4), dtype=torch.int64, device="cuda")
    d_root = torch.zeros(4, dtype=torch.int64, device="cuda")
    gpu_ctx.merkle4_tree_device(tag, d_buf, n_leaves, d_root, d_levels)
    rng = np.random.default_rng(k_total)
    good = rng.permutation(n_leaves)[:min(k_total - 5, n_leaves // 2)].astype(np.int64)
    bad = np.array([n_leaves, n_leaves + 1, n_leaves + 40, 2 ** 31 + 5, 2 ** 32 - 1], dtype=np.int64)
    idx = np.concatenate([good[:3], bad[:2], good[3:], bad[2:]])
    k = idx.shape[0]
    new = oracle_mod.fill_random(500 + k, k)
    d_idx = torch.from_numpy(idx.astype(np.uint32).view(np.int32)).cuda()
    d_new = torch.from_numpy(new.view(np.int64).copy()).cuda()
    with pytest.raises(ValueError):
        gpu_ctx.merkle4_update_device(tag, d_buf, n_leaves, d_levels, d_idx, d_new, k, d_root, check=True)
    with pytest.raises(ValueError):  # a repeated position
        gpu_ctx.merkle4_update_device(tag, d_buf, n_leaves, d_levels, torch.zeros(2, dtype=torch.int32, device="cuda"), d_new, 2, d_root, check=True)
    d_bad = torch.zeros(1, dtype=torch.int32, device="cuda")
    t = np.ascontiguousarray(tag, dtype=np.uint64)
    rc = _lib.lib().p252_merkle4_update_checked_device(gpu_ctx._h, t.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), d_buf.data_ptr(), n_leaves,
                                                       d_levels.data_ptr(), d_idx.data_ptr(), d_new.data_ptr(), k, d_root.data_ptr(),
                                                       d_bad.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    assert int(d_bad.item()) == bad.shape[0]
    updated = leaves.copy()
    in_range = idx < n_leaves
    updated[idx[in_range]] = new[in_range]
    o_root, o_levels, _ = oracle_mod.merkle4_tree(tag, updated, want_levels=True)
    got = d_buf.cpu().numpy().view(np.uint64)
    assert np.array_equal(got[:n_leaves], updated) and (got[n_leaves:] == 0x5A5A5A5A5A5A5A5A).all()
    assert np.array_equal(d_levels.cpu().numpy().view(np.uint64), o_levels)
    assert np.array_equal(d_root.cpu().numpy().view(np.uint64), o_root)
