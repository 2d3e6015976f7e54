"""p252_merkle4_openings_device (csrc/openings.hip): openings of a stored tree extracted on the device — pure data movement.
Checked against the host-side bookkeeping of poseidon252_amd.merkle.merkle4_openings (numpy indexing, no hashing) element for
element, and end to end: re-hashing the extracted openings (p252_merkle4_path_batch_device) gives the tree's root, which the
oracle confirms.  Ragged trees (missing siblings = zero scalar, hash.rs:22-26), a single leaf, duplicates, out-of-range positions."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_leaves,k", [(1, 3), (2, 4), (5, 9), (16, 40), (1000, 3000), (4 ** 6, 5000), (70001, 20000), (4 ** 9, 100000)])
def test_device_openings_equal_host_bookkeeping_and_rehash_to_the_root(gpu_ctx, oracle_mod, n_leaves, k):
    import torch
    import poseidon252_amd as P
    from poseidon252_amd import merkle, _lib
    tag = P.merkle4_tag()
    lv = oracle_mod.fill_random(0x0E + n_leaves, n_leaves)
    d_lv = torch.from_numpy(lv.view(np.int64)).to("cuda:0")
    root, d_levels = P.merkle4_tree(d_lv, tag=tag, ctx=gpu_ctx, want_levels=True)
    rng = np.random.default_rng(n_leaves)
    idx = rng.integers(0, n_leaves, size=k).astype(np.int32)
    idx[:2] = (0, n_leaves - 1)
    d_idx = torch.from_numpy(idx).to("cuda:0")
    out, sib, pos, depth = gpu_ctx.merkle4_openings_device(d_lv, n_leaves, d_levels, d_idx, k, check=True)
    torch.cuda.synchronize()
    assert depth == _lib.lib().p252_merkle4_depth(n_leaves) == (0 if n_leaves == 1 else len(bin(n_leaves - 1)[2:]) + 1 >> 1)
    levels = d_levels.cpu().numpy().view(np.uint64)[:P.levels_len(n_leaves)]
    h_sib, h_pos = merkle.merkle4_openings(lv, levels, idx)
    assert np.array_equal(out.cpu().numpy().view(np.uint64), lv[idx])
    assert np.array_equal(sib.cpu().numpy().view(np.uint64).reshape(h_sib.shape), h_sib)
    assert np.array_equal(pos.cpu().numpy().reshape(h_pos.shape), h_pos)
    # end to end on the device: build -> extract -> re-hash = the root, for every opening
    roots = torch.empty((k, 4), dtype=torch.int64, device="cuda:0")
    gpu_ctx.merkle4_path_batch_device(tag, out, sib, pos, depth, roots, k)
    torch.cuda.synchronize()
    assert bool((roots == root.view(1, 4)).all())
    assert np.array_equal(root.cpu().numpy().view(np.uint64), oracle_mod.merkle4_tree(tag, lv)[0])


def test_out_of_range_positions_yield_zero_openings_and_are_counted(gpu_ctx, oracle_mod):
    import torch
    import poseidon252_amd as P
    n = 1000
    lv = oracle_mod.fill_random(77, n)
    d_lv = torch.from_numpy(lv.view(np.int64)).to("cuda:0")
    _, d_levels = P.merkle4_tree(d_lv, ctx=gpu_ctx, want_levels=True)
    idx = np.array([5, n, 17, 2 ** 31 - 1, n - 1], dtype=np.int32)
    d_idx = torch.from_numpy(idx).to("cuda:0")
    out, sib, pos, depth = gpu_ctx.merkle4_openings_device(d_lv, n, d_levels, d_idx, len(idx))
    torch.cuda.synchronize()
    o = out.cpu().numpy().view(np.uint64)
    assert np.array_equal(o[[0, 2, 4]], lv[[5, 17, n - 1]]) and not o[[1, 3]].any()
    assert not sib.cpu().numpy()[[1, 3]].any() and not pos.cpu().numpy()[[1, 3]].any()
    with pytest.raises(ValueError):
        gpu_ctx.merkle4_openings_device(d_lv, n, d_levels, d_idx, len(idx), check=True)
    assert gpu_ctx.merkle4_openings_device(d_lv, n, d_levels, d_idx, 0)[0].shape[0] == 0  # k = 0: nothing to do


@pytest.mark.parametrize("n_leaves,k", [(1, 2), (2, 5), (3, 7), (7, 20), (1000, 3000), (2 ** 12, 5000), (70001, 20000)])
def test_merkle2_openings_extract_and_rehash(gpu_ctx, oracle_mod, n_leaves, k):
    """the arity-2 twins (Domain::Merkle2): p252_merkle2_openings_device out of a tree p252_merkle2_tree built, against plain numpy
    bookkeeping; p252_merkle2_path_batch_device against a level-by-level re-hash with the ORACLE's Merkle2 digests; and end to end:
    every extracted opening re-hashes to the root the oracle's own tree builder gives"""
    import torch
    import poseidon252_amd as P
    from poseidon252_amd import _lib
    tag = P.compute_tag(P.Domain.Merkle2, [2], 1)
    lv = oracle_mod.fill_random(0x2E + n_leaves, n_leaves)
    root, levels = gpu_ctx.merkle2_tree(tag, lv, want_levels=True)
    o_root, o_levels, _ = oracle_mod.merkle2_tree(tag, lv, want_levels=True)
    assert np.array_equal(root, o_root) and np.array_equal(levels, o_levels)
    d_lv = torch.from_numpy(lv.view(np.int64)).to("cuda:0")
    d_levels = torch.from_numpy(np.ascontiguousarray(levels if levels.shape[0] else np.zeros((1, 4), dtype=np.uint64)).view(np.int64)).to("cuda:0")
    rng = np.random.default_rng(n_leaves)
    idx = rng.integers(0, n_leaves, size=k).astype(np.int32)
    idx[:2] = (0, n_leaves - 1)
    out, sib, pos, depth = gpu_ctx.merkle4_openings_device(d_lv, n_leaves, d_levels, torch.from_numpy(idx).to("cuda:0"), k, check=True, arity=2)
    torch.cuda.synchronize()
    assert depth == _lib.lib().p252_merkle2_depth(n_leaves) == (0 if n_leaves == 1 else len(bin(n_leaves - 1)[2:]))
    # numpy bookkeeping: level arrays, the sibling of node j is j ^ 1 (zero when it lies beyond a ragged level)
    per_level, cnt, off = [lv], n_leaves, 0
    while cnt > 1:
        cnt = (cnt + 1) // 2
        per_level.append(levels[off:off + cnt])
        off += cnt
    h_sib = np.zeros((k, depth, 4), dtype=np.uint64)
    h_pos = np.zeros((k, depth), dtype=np.uint8)
    cur = idx.astype(np.int64).copy()
    for l in range(depth):
        other = cur ^ 1
        ok = other < per_level[l].shape[0]
        h_sib[ok, l] = per_level[l][other[ok]]
        h_pos[:, l] = cur & 1
        cur >>= 1
    assert np.array_equal(out.cpu().numpy().view(np.uint64), lv[idx])
    assert np.array_equal(sib.cpu().numpy().view(np.uint64).reshape(k, depth, 4), h_sib) and np.array_equal(pos.cpu().numpy().reshape(k, depth), h_pos)
    roots = torch.empty((k, 4), dtype=torch.int64, device="cuda:0")
    gpu_ctx.merkle2_path_batch_device(tag, out, sib, pos, depth, roots, k)
    torch.cuda.synchronize()
    assert np.array_equal(roots.cpu().numpy().view(np.uint64), np.broadcast_to(o_root, (k, 4)))
    # the re-hash kernel on ARBITRARY siblings / positions (not a consistent tree), against the oracle's Merkle2 digests level by level
    m, d2 = min(k, 700), 9
    leaves = oracle_mod.fill_random(1 + n_leaves, m)
    sibs = oracle_mod.fill_random(2 + n_leaves, m * d2).reshape(m, d2, 4)
    poss = rng.integers(0, 2, size=(m, d2), dtype=np.uint8)
    cur = leaves.copy()
    for l in range(d2):
        right = poss[:, l].astype(bool)
        pairs = np.where(right[:, None, None], np.stack([sibs[:, l], cur], axis=1), np.stack([cur, sibs[:, l]], axis=1))
        cur = oracle_mod.hash_batch(tag, np.ascontiguousarray(pairs), 2, 1).reshape(m, 4)
    got = torch.empty((m, 4), dtype=torch.int64, device="cuda:0")
    gpu_ctx.merkle2_path_batch_device(tag, torch.from_numpy(leaves.view(np.int64)).to("cuda:0"), torch.from_numpy(sibs.view(np.int64)).to("cuda:0"),
                                      torch.from_numpy(poss).to("cuda:0"), d2, got, m)
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy().view(np.uint64), cur)
