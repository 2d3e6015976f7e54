"""Multi-GPU: one process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).

The hot path shards with NO data-path collective: digests are independent, so rank r hashes its
contiguous slice.  Two small collectives exist around it:
  * broadcast_tables(): rank 0's derived constant table (25 KB) is broadcast and imported on every
    rank (BASELINE.json north_star); it is byte-identical to what each rank derives locally, which
    is asserted.
  * merkle4_tree_sharded(): every rank reduces its complete 4^k-leaf subtree to a root, the W roots
    (32 B each) are all-gathered and the <= log4(W)+1 top levels are hashed on every rank.
Messages are tens of bytes to tens of KB: latency-bound, never xGMI-bandwidth-bound.
"""
import numpy as np


def shard_range(n, rank, world):
    """contiguous slice [lo, hi) of n items owned by `rank`; sizes differ by at most 1"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _dist():
    import torch.distributed as dist
    return dist


def broadcast_tables(ctx, device=None):
    """rank 0 -> all: the device constant table (BASELINE.json north_star: RCCL broadcast of constants).  The table is
    a pure function of the arc.bin / mds.bin compiled into the library, so the received bytes must equal the local
    derivation: a mismatch (corrupted broadcast, ranks running different builds) RAISES — p252_tables_import itself
    refuses such a table — instead of hashing with unchecked constants.  Returns True."""
    import torch
    dist = _dist()
    local = ctx.tables_export()
    t = torch.from_numpy(local.copy())
    if device is not None:
        t = t.to(device)
    if dist.is_initialized():
        dist.broadcast(t, src=0)
    got = t.cpu().numpy()
    if not np.array_equal(got, local):
        raise RuntimeError("constant table received from rank 0 differs from the locally derived one "
                           "(corrupted broadcast or mismatched library builds)")
    ctx.tables_import(got)  # validated again inside the library (byte-identical to its own derivation)
    return True


def is_power_of_4(n):
    return n > 0 and (n & (n - 1)) == 0 and (n.bit_length() - 1) % 2 == 0


def merkle4_tree_sharded(local_leaves, tag, subtree_fn, top_fn, device=None):
    """Root of the tree over the concatenation (in rank order) of every rank's leaves.

    Every rank must hold the same number 4^k of leaves, so that the rank roots are exactly the nodes
    of level k of the global tree.  subtree_fn(leaves) -> root (4 limbs) reduces the local subtree on
    this rank's GPU; top_fn(nodes (W,4)) -> root builds the top of the tree from the gathered roots
    (zero-padded per src/hash.rs:22-26).  Returns the global root as numpy uint64 (4,) on every rank."""
    import torch
    dist = _dist()
    world = dist.get_world_size() if dist.is_initialized() else 1
    n_local = (local_leaves.numel() * local_leaves.element_size() // 32) if hasattr(local_leaves, "numel") else np.asarray(local_leaves).reshape(-1, 4).shape[0]
    if world > 1 and not is_power_of_4(n_local):
        raise ValueError("sharded tree needs 4^k leaves per rank (got %d)" % n_local)
    root = subtree_fn(local_leaves)
    if hasattr(root, "cpu"):
        root_t = root.reshape(4).contiguous()
    else:
        root_t = torch.from_numpy(np.ascontiguousarray(root, dtype=np.uint64).view(np.int64).copy())
        if device is not None:
            root_t = root_t.to(device)
    if world == 1:
        return root_t.cpu().numpy().view(np.uint64).reshape(4)
    gathered = [torch.empty_like(root_t) for _ in range(world)]
    dist.all_gather(gathered, root_t)  # the path's only exchange step: W x 32 bytes
    nodes = torch.stack(gathered).cpu().numpy().view(np.uint64).reshape(world, 4)
    return np.asarray(top_fn(nodes), dtype=np.uint64).reshape(4)
