"""N>1 path on CPU: world_size-2 gloo processes exercise the sharding and the single exchange step
(all-gather of subtree roots) of poseidon252_amd/distributed.py.  No GPU here, so the per-rank
"device" work is stood in for by the oracle (test infrastructure acting as the checker's double);
the assertions are about partitioning, ordering, zero-padding of the top levels and the collective."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from poseidon252_amd import distributed as D


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 1000, (1 << 20) + 3):
        for world in (1, 2, 3, 4, 8):
            spans = [D.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_power_of_4():
    assert [D.is_power_of_4(n) for n in (0, 1, 2, 4, 8, 16, 64, 1 << 24, 1 << 25)] == [False, True, False, True, False, True, True, True, False]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_local, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    tag = oracle.tag(0, [4], 1)
    leaves = oracle.fill_random(0x5eed, n_local * world).reshape(world, n_local, 4)
    # --- sharded digests: rank hashes its slice, no collective on the data path ---
    n_msgs = 1000 + 3
    msgs = oracle.fill_random(77, 4 * n_msgs).reshape(n_msgs, 4, 4)
    lo, hi = D.shard_range(n_msgs, rank, world)
    mine = oracle.hash_batch(tag, msgs[lo:hi], 4, 1)
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, hi, mine))  # test-only gather to compare with the unsharded run
    # --- sharded tree ---
    root = D.merkle4_tree_sharded(
        torch.from_numpy(leaves[rank].view(np.int64).copy()), tag,
        subtree_fn=lambda lv: oracle.merkle4_tree(tag, lv.numpy().view(np.uint64).reshape(-1, 4))[0],
        top_fn=lambda nodes: oracle.merkle4_tree(tag, nodes)[0])
    if rank == 0:
        full = np.concatenate([g[2] for g in sorted(gathered, key=lambda g: g[0])])
        ok_digest = np.array_equal(full, oracle.hash_batch(tag, msgs, 4, 1))
        ok_tree = np.array_equal(root, oracle.merkle4_tree(tag, leaves.reshape(-1, 4))[0])
        out_q.put((ok_digest, ok_tree))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_local", [(2, 16), (2, 64), (2, 1)])
def test_gloo_world2_sharded_digests_and_tree(world, n_local):
    import oracle
    oracle.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_local, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok == (True, True)


def test_sharded_tree_rejects_incomplete_subtrees():
    # with world > 1 every rank must hold 4^k leaves; single process (world 1) accepts anything
    root = D.merkle4_tree_sharded(torch.zeros((5, 4), dtype=torch.int64), np.zeros(4, dtype=np.uint64),
                                  subtree_fn=lambda lv: np.arange(4, dtype=np.uint64), top_fn=None)
    assert np.array_equal(root, np.arange(4, dtype=np.uint64))


class _FakeCtx:
    """stands in for poseidon252_amd.Context where no GPU exists: only the two table calls broadcast_tables uses"""

    def __init__(self, table):
        self.table, self.imported = table, None

    def tables_export(self):
        return self.table.copy()

    def tables_import(self, t):
        self.imported = np.asarray(t).copy()


def _bcast_worker(rank, world, port, corrupt, out_q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    table = np.arange(5000, dtype=np.int32)
    if corrupt and rank == 1:
        table[1234] ^= 1  # this rank's library derived something else (mismatched build / corrupted memory)
    ctx = _FakeCtx(table)
    try:
        same = D.broadcast_tables(ctx)
        out_q.put((rank, "ok", bool(same), ctx.imported is not None and np.array_equal(ctx.imported, np.arange(5000, dtype=np.int32))))
    except RuntimeError as e:
        out_q.put((rank, "raised", "differs" in str(e), ctx.imported is None))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("corrupt", [False, True])
def test_broadcast_tables_raises_on_a_mismatching_rank(corrupt):
    """a table that differs from the local derivation must never be imported silently"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bcast_worker, args=(r, 2, port, corrupt, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if not corrupt:
        assert got == [(0, "ok", True, True), (1, "ok", True, True)]
    else:
        assert got[0] == (0, "ok", True, True)          # rank 0 is the source: its own table
        assert got[1] == (1, "raised", True, True)      # rank 1 refuses the import
