"""Arity-4 Merkle trees over Hash::digest(Domain::Merkle4, [c0,c1,c2,c3]) nodes.

The reference removed its tree builder in 0.29.0 (CHANGELOG.md:164-168); only the node hash
remains (src/hash.rs:22-26: total input exactly 4 scalars, empty slots = zero scalar).  The tree is
the obvious composition: levels are hashed while more than one node remains (a single
leaf is its own root), a level whose length is not a multiple of 4 is zero-padded.

Openings (the `poseidon-merkle` use named in AGENTS.md:62-66): an opening of leaf
i is, per level, the 3 siblings of the node on the path and the node's position 0..3 among its
parent's children; `merkle4_path_roots` re-hashes n such branches in one kernel launch.
"""
import numpy as np

from .hash import Context, Domain, compute_tag, _as_scalars, _is_torch


def merkle4_tag():
    return compute_tag(Domain.Merkle4, [4], 1)


def levels_len(n_leaves):
    total, c = 0, n_leaves
    while c > 1:
        c = (c + 3) // 4
        total += c
    return total


def permutations(n_leaves):
    """number of Hades permutations a tree over n_leaves costs (= number of internal nodes)"""
    return levels_len(n_leaves)


def merkle4_tree(leaves, tag=None, ctx=None, want_levels=False):
    """leaves: (n,4) uint64 numpy, or a torch CUDA tensor holding n BlsScalars.  Returns the root
    (numpy (4,) / torch (4,) int64 on the same device) and optionally all upper levels bottom-up."""
    ctx = ctx or Context.default()
    tag = merkle4_tag() if tag is None else _as_scalars(tag).reshape(4)
    if _is_torch(leaves):
        import torch
        n = leaves.numel() * leaves.element_size() // 32
        root = torch.empty(4, dtype=torch.int64, device=leaves.device)
        levels = torch.empty((max(levels_len(n), 1), 4), dtype=torch.int64, device=leaves.device) if want_levels else None
        ctx.merkle4_tree_device(tag, leaves, n, root, levels)
        return (root, levels) if want_levels else root
    return ctx.merkle4_tree(tag, np.asarray(leaves), want_levels=want_levels)


def merkle4_forest(leaves, leaves_per_tree, tag=None, ctx=None, want_levels=False):
    """roots of the n_trees = n / leaves_per_tree independent complete trees stored tree-major in `leaves` (a torch CUDA tensor
    or an (n,4) uint64 numpy array): one kernel launch per level across all trees.  Returns roots (n_trees,4) — torch int64 on
    the device for device input, numpy uint64 otherwise — and optionally the level-major array of all levels."""
    ctx = ctx or Context.default()
    tag = merkle4_tag() if tag is None else _as_scalars(tag).reshape(4)
    dev_in = _is_torch(leaves)
    if not dev_in and not want_levels:  # host leaves, roots only: the library's staged pipeline (no torch needed)
        return ctx.merkle4_forest(tag, leaves, leaves_per_tree)
    import torch
    d = leaves if dev_in else torch.from_numpy(_as_scalars(leaves).reshape(-1, 4).view(np.int64)).to("cuda:%d" % ctx.device)
    n = d.numel() * d.element_size() // 32
    if leaves_per_tree < 1 or n % leaves_per_tree:
        raise ValueError("forest: %d leaves are not a whole number of %d-leaf trees" % (n, leaves_per_tree))
    n_trees = n // leaves_per_tree
    roots = torch.empty((n_trees, 4), dtype=torch.int64, device=d.device)
    levels = torch.empty((max(n_trees * levels_len(leaves_per_tree), 1), 4), dtype=torch.int64, device=d.device) if want_levels else None
    ctx.merkle4_forest_device(tag, d, n_trees, leaves_per_tree, roots, levels)
    if not dev_in:
        torch.cuda.synchronize(d.device)
        roots = roots.cpu().numpy().view(np.uint64)
        levels = levels.cpu().numpy().view(np.uint64) if want_levels else None
    return (roots, levels) if want_levels else roots


def merkle4_openings(leaves, levels, indices):
    """Host-side bookkeeping (no hashing): sibling paths of the leaves at `indices` out of a built tree.
    leaves (n,4), levels = concatenated upper levels as merkle4_tree(..., want_levels=True) returns.
    Returns (siblings (m,depth,3,4) uint64, positions (m,depth) uint8); missing siblings = zero scalar."""
    leaves = _as_scalars(leaves).reshape(-1, 4)
    levels = _as_scalars(levels).reshape(-1, 4)
    per_level, cnt, off = [leaves], leaves.shape[0], 0
    while cnt > 1:
        cnt = (cnt + 3) // 4
        per_level.append(levels[off:off + cnt])
        off += cnt
    depth = len(per_level) - 1
    idx = np.asarray(indices, dtype=np.int64).reshape(-1)
    sib = np.zeros((idx.shape[0], depth, 3, 4), dtype=np.uint64)
    pos = np.zeros((idx.shape[0], depth), dtype=np.uint8)
    cur = idx.copy()
    for l in range(depth):
        nodes = per_level[l]
        p = cur & 3
        pos[:, l] = p
        base = cur - p
        for m in range(idx.shape[0]):
            others = [base[m] + k for k in range(4) if k != p[m]]
            for s, j in enumerate(others):
                if j < nodes.shape[0]:
                    sib[m, l, s] = nodes[j]
        cur = cur >> 2
    return sib, pos


def merkle4_path_roots(leaves, siblings, positions, tag=None, ctx=None):
    """Roots recomputed from n openings (numpy host buffers); compare with the tree root to verify."""
    ctx = ctx or Context.default()
    tag = merkle4_tag() if tag is None else _as_scalars(tag).reshape(4)
    return ctx.merkle4_path_batch(tag, leaves, siblings, positions)
