"""Driver of tests/test_comm_mock_ranks.py — runs in a subprocess whose P252_RCCL_PATH names tests/cpp/mock_rccl.cpp built as a
shared object: the SHIPPED library resolves it as its RCCL (csrc/rccl_dyn.hpp), several ranks on ONE device.  Prints one JSON line
with what it verified."""
import json
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import oracle
import poseidon252_amd as P
from poseidon252_amd import comm as C, multi

assert os.environ.get("P252_RCCL_PATH") and not os.environ.get("P252_LIB_PATH") and os.environ.get("P252_COMM_ALLOW_SHARED_DEVICE") == "1"
assert os.path.samefile(C.backend(), os.environ["P252_RCCL_PATH"]), C.backend()  # the resolver took the mock, nothing else
tag = P.merkle4_tag()
report = {}


def leaves_of(world, per, seed):
    return [oracle.fill_random(seed + r, per) for r in range(world)]


def ranks_in_threads(world, per, seed, repeats):
    """one thread per rank, each with its own context: p252_comm_unique_id on rank 0, the id handed round, p252_comm_create_rank on every
    rank (collective; broadcasts and validates the constants), then p252_merkle4_tree_sharded_device `repeats` times"""
    lv = leaves_of(world, per, seed)
    exp = oracle.merkle4_tree(tag, np.concatenate(lv))[0]
    box, got, errs = {}, [None] * world, []
    ready = threading.Barrier(world)

    def exchange_for(rank):
        def exchange(id_bytes):
            if rank == 0:
                box["id"] = id_bytes
            ready.wait()
            return box["id"]
        return exchange

    def run(rank):
        try:
            ctx = P.Context(0)
            c = C.Comm.create_rank(ctx, rank, world, exchange_for(rank))
            assert c.rank == rank and c.size == world
            d = torch.from_numpy(lv[rank].view(np.int64).copy()).to("cuda:0")
            d_root = torch.zeros(4, dtype=torch.int64, device="cuda:0")
            for _ in range(repeats):
                d_root.zero_()
                c.merkle4_tree_sharded_device(tag, d, per, d_root)
                torch.cuda.synchronize()
                assert np.array_equal(d_root.cpu().numpy().view(np.uint64), exp), "rank %d: root differs from the oracle's" % rank
            got[rank] = d_root.cpu().numpy().view(np.uint64).copy()
            ready.wait()  # every rank still here when the communicators go
            c.destroy()
            ctx.close()
        except Exception as e:  # noqa: BLE001
            errs.append("rank %d: %r" % (rank, e))
            try:
                ready.abort()
            except Exception:
                pass
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not errs and not any(t.is_alive() for t in ts), errs
    assert all(np.array_equal(g, exp) for g in got)


for world, per in ((8, 4 ** 5), (2, 4 ** 6), (3, 4 ** 3), (5, 16), (8, 1)):
    ranks_in_threads(world, per, 0xC10D + world, repeats=3)
report["one thread per rank (p252_comm_create_rank + p252_merkle4_tree_sharded_device)"] = "worlds 8, 2, 3, 5, 8 x 3 builds: every rank's root == the oracle's tree over the concatenation"



def a_rank_fails(world, per, bad_rank, seed):
    """one rank's LOCAL build fails (here: a misaligned leaf pointer, refused by the tree builder after the communicator
    bookkeeping) — it returns its error and still enters the all-gather with the all-ones sentinel.  Every healthy rank: the call
    itself has returned P252_OK (asynchronous), the root on the device is all-ones, p252_comm_check reports P252_ERR_COMM naming the
    failed rank — once; a following healthy build on the same communicator gives the oracle's root again."""
    lv = leaves_of(world, per, seed)
    exp = oracle.merkle4_tree(tag, np.concatenate(lv))[0]
    box, errs, seen = {}, [], [None] * world
    ready = threading.Barrier(world)

    def exchange_for(rank):
        def exchange(id_bytes):
            if rank == 0:
                box["id"] = id_bytes
            ready.wait()
            return box["id"]
        return exchange

    def run(rank):
        try:
            ctx = P.Context(0)
            c = C.Comm.create_rank(ctx, rank, world, exchange_for(rank))
            raw = torch.zeros(per * 4 + 1, dtype=torch.int64, device="cuda:0")
            raw[:per * 4] = torch.from_numpy(lv[rank].view(np.int64).copy()).reshape(-1).to("cuda:0")
            good = raw[:per * 4]
            d_root = torch.zeros(4, dtype=torch.int64, device="cuda:0")
            if rank == bad_rank:
                shifted = raw[1:per * 4 + 1]  # 8 bytes off a 16-byte boundary
                try:
                    c.merkle4_tree_sharded_device(tag, shifted, per, d_root)
                    raise AssertionError("the misaligned build was accepted")
                except ValueError:
                    seen[rank] = "error returned"
            else:
                c.merkle4_tree_sharded_device(tag, good, per, d_root)  # P252_OK: the failure is a peer's and the call is asynchronous
                torch.cuda.synchronize()
                assert (d_root.cpu().numpy().view(np.uint64) == np.uint64(0xffffffffffffffff)).all(), "rank %d: root not poisoned" % rank
                try:
                    c.check()
                    raise AssertionError("p252_comm_check did not report the failed peer")
                except P.DeviceError as e:
                    assert "rank %d " % bad_rank in str(e), str(e)
                c.check()  # reported once
                seen[rank] = "poisoned + reported"
            ready.wait()
            d_root.zero_()
            c.merkle4_tree_sharded_device(tag, good, per, d_root)  # the communicator is still usable
            c.check()
            assert np.array_equal(d_root.cpu().numpy().view(np.uint64), exp), "rank %d: root after the failure differs" % rank
            ready.wait()
            c.destroy()
            ctx.close()
        except Exception as e:  # noqa: BLE001
            errs.append("rank %d: %r" % (rank, e))
            try:
                ready.abort()
            except Exception:
                pass
    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not errs and not any(t.is_alive() for t in ts), errs
    assert seen[bad_rank] == "error returned" and all(s == "poisoned + reported" for r, s in enumerate(seen) if r != bad_rank), seen


for world, per, bad in ((4, 4 ** 3, 2), (8, 16, 0), (3, 4, 2)):
    a_rank_fails(world, per, bad, 0xBAD0 + world)
report["a rank's local build fails (sentinel in the all-gather)"] = "worlds 4, 8, 3: healthy ranks' root poisoned, p252_comm_check names the rank once, next build correct"

# one process, an array of contexts (p252_comm_create_all; one thread drives every rank inside ncclGroupStart / End)
for world, per in ((8, 4 ** 5), (4, 4 ** 4), (3, 4 ** 2), (6, 4)):
    ctxs = [P.Context(0) for _ in range(world)]
    comms = C.Comm.create_all(ctxs)
    assert [c.rank for c in comms] == list(range(world)) and all(c.size == world for c in comms)
    lv = leaves_of(world, per, 0xABC + world)
    exp = oracle.merkle4_tree(tag, np.concatenate(lv))[0]
    d = [torch.from_numpy(x.view(np.int64).copy()).to("cuda:0") for x in lv]
    for _ in range(2):
        assert np.array_equal(multi.merkle4_tree_multi_device(ctxs, tag, d, per), exp)
    d_roots = [torch.zeros(4, dtype=torch.int64, device="cuda:0") for _ in range(world)]
    C.merkle4_tree_multi_device_resident(ctxs, tag, d, per, d_roots)
    torch.cuda.synchronize()
    assert all(np.array_equal(r.cpu().numpy().view(np.uint64), exp) for r in d_roots)  # the root is resident on EVERY rank
    # the contexts in another order / a subset of them are not this clique (rank t must be ctxs[t]) and belong to a
    # communicator the CALLER made — ABI 5 accepted any context array, so the call must still work: the roots are gathered through
    # the host; only the resident variant, which has no host path, refuses (P252_ERR_COMM)
    assert np.array_equal(multi.merkle4_tree_multi_device(ctxs[::-1], tag, d, per), exp)
    if world > 2:
        sub = oracle.merkle4_tree(tag, np.concatenate(lv[:2]))[0]
        assert np.array_equal(multi.merkle4_tree_multi_device(ctxs[:2], tag, d[:2], per), sub)
    try:
        C.merkle4_tree_multi_device_resident(ctxs[::-1], tag, d, per, d_roots)
        raise SystemExit("the resident variant accepted contexts of a foreign communicator")
    except P.DeviceError:
        pass
    assert np.array_equal(multi.merkle4_tree_multi_device(ctxs, tag, d, per), exp)  # and the caller's clique is intact
    for c in comms:
        c.destroy()
    for c in ctxs:
        c.close()
report["one process, array of contexts (p252_comm_create_all, p252_merkle4_tree_multi_device[_resident])"] = "worlds 8, 4, 3, 6: root == oracle, resident on every rank"

# the multi-device entry point creating its communicator itself, and taking it along when the contexts are destroyed
ctxs = [P.Context(0) for _ in range(8)]
lv = leaves_of(8, 4 ** 4, 0x5151)
d = [torch.from_numpy(x.view(np.int64).copy()).to("cuda:0") for x in lv]
assert np.array_equal(multi.merkle4_tree_multi_device(ctxs, tag, d, 4 ** 4), oracle.merkle4_tree(tag, np.concatenate(lv))[0])
try:
    C.Comm.create_all(ctxs)
    raise SystemExit("contexts that already belong to the library-made communicator were accepted")
except ValueError:
    pass
# ADVICE r4 (the regression of ABI 6): another count / order / subset of contexts that sit in a LIBRARY-made communicator — the
# pattern of a shared `ctxs` fixture: ctxs[:8], then ctxs[:4], ctxs[:2], a permutation, all 8 again.  The library tears its own
# cliques down and makes one over the array at hand; every call returns the oracle's root.
for sel in (list(range(4)), [0, 1], [3, 2, 1, 0], [5, 6, 7], list(range(8))):
    exp_sel = oracle.merkle4_tree(tag, np.concatenate([lv[t] for t in sel]))[0]
    assert np.array_equal(multi.merkle4_tree_multi_device([ctxs[t] for t in sel], tag, [d[t] for t in sel], 4 ** 4), exp_sel), sel
d_res = [torch.zeros(4, dtype=torch.int64, device="cuda:0") for _ in range(3)]
C.merkle4_tree_multi_device_resident(ctxs[2:5], tag, d[2:5], 4 ** 4, d_res)  # the resident variant re-forms the clique as well
torch.cuda.synchronize()
assert all(np.array_equal(r.cpu().numpy().view(np.uint64), oracle.merkle4_tree(tag, np.concatenate(lv[2:5]))[0]) for r in d_res)
# one context: no communicator is made at all (nothing to exchange), and the context is not tied to a hidden one
solo = P.Context(0)
assert np.array_equal(multi.merkle4_tree_multi_device([solo], tag, d[:1], 4 ** 4), oracle.merkle4_tree(tag, lv[0])[0])
own = C.Comm.create_all([solo])
own[0].destroy()
solo.close()
for c in ctxs:
    c.close()
ctxs = [P.Context(0) for _ in range(8)]  # and again with fresh contexts: nothing of the first clique is left behind
assert np.array_equal(multi.merkle4_tree_multi_device(ctxs, tag, d, 4 ** 4), oracle.merkle4_tree(tag, np.concatenate(lv))[0])
for c in ctxs:
    c.close()
report["communicator created on first use by p252_merkle4_tree_multi_device, re-formed for other context arrays, destroyed with its contexts"] = "ok, twice"
print(json.dumps(report))
