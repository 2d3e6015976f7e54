"""Arity-4 Merkle trees over Hash::digest(Domain::Merkle4, [c0,c1,c2,c3]) nodes.

The reference removed its tree builder in 0.29.0 (CHANGELOG.md:164-168); only the node hash
remains (src/hash.rs:22-26: total input exactly 4 scalars, empty slots = zero scalar).  The tree is
the obvious composition: levels are hashed while more than one node remains (a single leaf is its own root), a level whose length
is not a multiple of 4 is zero-padded.
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
