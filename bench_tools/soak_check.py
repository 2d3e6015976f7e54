"""One-off heavier parity soak on the GPU box (developer tool, not part of the test-suite): 2^18 full-state
permutations and a few thousand random sponge shapes against the oracle."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import oracle
import poseidon252_amd as P

if "--long" in sys.argv:  # (torch must initialise HIP before the library's staging threads exist: initialised later it finds no device)
    import torch
    torch.cuda.init()

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

# --long [minutes]: a randomized differential run over EVERY entry point (both kernel families: batch sizes straddle the
# 8,192 / 16,384 switches), each call compared with the oracle limb for limb
if "--long" in sys.argv:
    from poseidon252_amd import encryption as E
    minutes = float(sys.argv[sys.argv.index("--long") + 1]) if len(sys.argv) > sys.argv.index("--long") + 1 else 5.0
    counts = {"permute": 0, "sponge": 0, "digest": 0, "tree leaves": 0, "openings": 0, "encrypt+decrypt": 0, "truncate": 0, "bytes": 0, "tree updates": 0,
              "forest trees": 0, "sharded-tree leaves (RCCL, one rank)": 0,
              "openings extracted on the device": 0, "fused truncated outputs": 0, "root-only builds on concurrent streams": 0, "openings verified in bulk": 0}
    from poseidon252_amd import comm as C
    comm_ctx = P.Context(0)
    comm1 = C.Comm.create_rank(comm_ctx, 0, 1, lambda b: b)  # the library's RCCL communicator on the real backend (round 4)
    tag2 = oracle.tag(1, [2], 1)
    P_ = oracle.P
    t0 = time.time()
    it = 0
    while time.time() - t0 < 60 * minutes:
        it += 1
        seed = int(rng.integers(1, 1 << 30))
        kind = it % 15
        if rng.integers(0, 23) == 0:  # round 6: the grow-only scratch given back at random points between calls (p252_trim) — the next
            import torch             # call of any kind must allocate what it needs again and still equal the oracle
            torch.cuda.synchronize()
            ctx.trim()
            assert ctx.scratch_residue() == 0
            counts["trims"] = counts.get("trims", 0) + 1
        n = int(rng.choice([int(rng.integers(1, 300)), int(rng.integers(300, 8193)), int(rng.integers(8193, 20000))]))
        if kind == 0:
            n = min(n, 12000)
            st = oracle.fill_random(seed, 5 * n).reshape(n, 5, 4)
            assert np.array_equal(ctx.permute_batch(st), oracle.permute_batch(st)), ("permute", n, seed)
            counts["permute"] += n
        elif kind == 1:
            in_len, out_len = int(rng.integers(1, 50)), int(rng.integers(1, 10))
            n = min(n, 40000 // in_len + 1)
            tag = oracle.fill_random(seed + 1, 1).reshape(4)
            msg = oracle.fill_random(seed, n * in_len).reshape(n, in_len, 4)
            assert np.array_equal(ctx.hash_batch(tag, msg, in_len, out_len), oracle.hash_batch(tag, msg, in_len, out_len, threads=8)), ("sponge", n, in_len, out_len, seed)
            counts["sponge"] += n
        elif kind == 2:
            x = oracle.fill_random(seed, 4 * n).reshape(n, 4, 4)
            assert np.array_equal(ctx.hash_batch(mtag, x, 4, 1), oracle.hash_batch(mtag, x, 4, 1, threads=8)), ("digest", n, seed)
            tag2 = oracle.tag(1, [2], 1)
            assert np.array_equal(ctx.hash_batch(tag2, np.ascontiguousarray(x[:, :2]), 2, 1), oracle.hash_batch(tag2, np.ascontiguousarray(x[:, :2]), 2, 1, threads=8))
            counts["digest"] += 2 * n
        elif kind == 3:
            leaves = int(rng.choice([n, 4 ** int(rng.integers(1, 9)), 4 ** int(rng.integers(1, 8)) + int(rng.integers(1, 50))]))
            lv = oracle.fill_random(seed, leaves)
            root, levels = ctx.merkle4_tree(mtag, lv, want_levels=True)
            o_root, o_levels, _ = oracle.merkle4_tree(mtag, lv, want_levels=True)
            assert np.array_equal(root, o_root) and np.array_equal(levels, o_levels), ("tree", leaves, seed)
            counts["tree leaves"] += leaves
        elif kind == 4:
            depth = int(rng.integers(0, 14))
            n = min(n, 9000, 30000 // max(depth, 1) + 1)  # (the oracle re-hashes n * depth nodes on one core)
            leaves = oracle.fill_random(seed, n)
            sib = oracle.fill_random(seed + 1, n * depth * 3).reshape(n, depth, 3, 4) if depth else np.zeros((n, 0, 3, 4), dtype=np.uint64)
            pos = rng.integers(0, 4, size=(n, depth), dtype=np.uint8)
            assert np.array_equal(ctx.merkle4_path_batch(mtag, leaves, sib, pos), oracle.merkle4_path_batch(mtag, leaves, sib, pos)), ("openings", n, depth, seed)
            counts["openings"] += n
        elif kind == 5:
            length = int(rng.choice([1, 2, 3, 4, 5, 8, 21, 42]))
            n = min(n, 12000 // length + 1)
            variant = int(rng.integers(0, 2))
            msg = oracle.fill_random(seed, n * length).reshape(n, length, 4)
            sec = oracle.fill_random(seed + 1, 2 * n).reshape(n, 2, 4)
            non = oracle.fill_random(seed + 2, n)
            c = E.encrypt_batch(msg, sec, non, ctx=ctx, variant=variant)
            assert np.array_equal(c, oracle.encrypt_batch(E.encryption_tag(length, variant), msg, sec, non, variant=variant)), ("encrypt", n, length, variant, seed)
            back, ok = E.decrypt_batch(c, sec, non, ctx=ctx, variant=variant)
            assert ok.all() and np.array_equal(back, msg), ("decrypt", n, length, variant, seed)
            counts["encrypt+decrypt"] += n
        elif kind == 6:
            x = oracle.fill_random(seed, n)
            import torch
            d = torch.from_numpy(x.view(np.int64)).cuda()
            o = torch.empty_like(d)
            ctx.truncate250_device(d, o, n)
            torch.cuda.synchronize()
            assert np.array_equal(o.cpu().numpy().view(np.uint64), P.truncate250(x)), ("truncate", n, seed)
            counts["truncate"] += n
        elif kind == 8:
            import torch
            leaves_n = int(rng.choice([int(rng.integers(1, 5000)), 4 ** int(rng.integers(1, 9)), int(rng.integers(5000, 80000))]))
            k = int(min(leaves_n, rng.choice([1, int(rng.integers(1, 200)), int(rng.integers(200, 12000))])))
            lv = oracle.fill_random(seed, leaves_n)
            total = oracle.levels_total(leaves_n)
            d_lv = torch.from_numpy(lv.view(np.int64).copy()).cuda()
            d_levels = torch.zeros((max(total, 1), 4), dtype=torch.int64, device="cuda")
            d_root = torch.zeros(4, dtype=torch.int64, device="cuda")
            ctx.merkle4_tree_device(mtag, d_lv, leaves_n, d_root, d_levels)
            idx = rng.permutation(leaves_n)[:k]
            new = oracle.fill_random(seed + 1, k)
            ctx.merkle4_update_device(mtag, d_lv, leaves_n, d_levels, torch.from_numpy(idx.astype(np.int32)).cuda(),
                                      torch.from_numpy(new.view(np.int64).copy()).cuda(), k, d_root)
            torch.cuda.synchronize()
            upd = lv.copy()
            upd[idx] = new
            o_root, o_levels, _ = oracle.merkle4_tree(mtag, upd, want_levels=True)
            assert np.array_equal(d_levels.cpu().numpy().view(np.uint64)[:total], o_levels) and np.array_equal(d_root.cpu().numpy().view(np.uint64), o_root), ("update", leaves_n, k, seed)
            counts["tree updates"] += k
        elif kind == 9:  # forests: one launch per level across all trees, both arities, levels layout included
            import torch
            arity = int(rng.choice([4, 2]))
            per = arity ** int(rng.integers(0, 7 if arity == 4 else 11))
            n_trees = int(min(rng.choice([1, int(rng.integers(1, 50)), int(rng.integers(50, 3000))]), max(1, 400000 // per)))
            lv = oracle.fill_random(seed, n_trees * per)
            d = torch.from_numpy(lv.view(np.int64).copy()).cuda()
            total = oracle.levels_total(per) if arity == 4 else per - 1
            d_roots = torch.zeros((n_trees, 4), dtype=torch.int64, device="cuda")
            d_levels = torch.zeros((max(n_trees * total, 1), 4), dtype=torch.int64, device="cuda")
            ftag = mtag if arity == 4 else tag2
            ctx.merkle4_forest_device(ftag, d, n_trees, per, d_roots, d_levels, arity=arity)
            torch.cuda.synchronize()
            roots, levels = d_roots.cpu().numpy().view(np.uint64), d_levels.cpu().numpy().view(np.uint64)
            if arity == 4:  # the host-buffer twin (staged through the lanes when the forest is several chunks)
                assert np.array_equal(ctx.merkle4_forest(ftag, lv, per), roots), ("host forest", per, n_trees, seed)
            for t in sorted(set(int(v) for v in rng.integers(0, n_trees, size=min(n_trees, 12)))):
                leaf = lv[t * per:(t + 1) * per]
                if per == 1:
                    assert np.array_equal(roots[t], leaf[0]), ("forest", arity, per, n_trees, seed)
                    continue
                o_root, o_levels, _ = (oracle.merkle4_tree if arity == 4 else oracle.merkle2_tree)(ftag, leaf, want_levels=True)
                assert np.array_equal(roots[t], o_root), ("forest root", arity, per, n_trees, t, seed)
                off_f, off_t, width = 0, 0, per // arity
                while width >= 1:
                    assert np.array_equal(levels[off_f + t * width:off_f + (t + 1) * width], o_levels[off_t:off_t + width]), ("forest levels", arity, per, n_trees, t, width, seed)
                    off_f, off_t, width = off_f + n_trees * width, off_t + width, width // arity
            counts["forest trees"] += n_trees
        elif kind == 11:  # openings of a stored tree extracted on the device: against the host bookkeeping, then re-hashed to the root
            import torch
            from poseidon252_amd import merkle
            leaves_n = int(rng.choice([int(rng.integers(1, 3000)), 4 ** int(rng.integers(0, 9)), int(rng.integers(3000, 90000))]))
            k = int(rng.integers(1, 4000))
            lv = oracle.fill_random(seed, leaves_n)
            d_lv = torch.from_numpy(lv.view(np.int64).copy()).cuda()
            d_root, d_levels = P.merkle4_tree(d_lv, tag=mtag, ctx=ctx, want_levels=True)
            idx = rng.integers(0, leaves_n, size=k).astype(np.int32)
            out, sib, pos, depth = ctx.merkle4_openings_device(d_lv, leaves_n, d_levels, torch.from_numpy(idx).cuda(), k, check=True)
            roots = torch.empty((k, 4), dtype=torch.int64, device="cuda")
            ctx.merkle4_path_batch_device(mtag, out, sib, pos, depth, roots, k)
            torch.cuda.synchronize()
            h_sib, h_pos = merkle.merkle4_openings(lv, d_levels.cpu().numpy().view(np.uint64)[:oracle.levels_total(leaves_n)], idx)
            assert np.array_equal(sib.cpu().numpy().view(np.uint64).reshape(h_sib.shape), h_sib) and np.array_equal(pos.cpu().numpy().reshape(h_pos.shape), h_pos), ("openings extract", leaves_n, k, seed)
            assert bool((roots == d_root.view(1, 4)).all()) and np.array_equal(d_root.cpu().numpy().view(np.uint64), oracle.merkle4_tree(mtag, lv)[0]), ("openings rehash", leaves_n, k, seed)
            counts["openings extracted on the device"] += k
        elif kind == 12:  # round 5: finalize_truncated in the digest kernels' output stage, random shapes, host and device buffers
            import torch
            in_len, out_len = int(rng.choice([4, 2, int(rng.integers(1, 50))])), int(rng.integers(1, 10))
            if in_len in (4, 2) and rng.integers(0, 2):
                out_len = 1  # the single-permutation digest kernels
            n = min(n, 40000 // in_len + 1)
            tag = oracle.fill_random(seed + 1, 1).reshape(4)
            msg = oracle.fill_random(seed, n * in_len).reshape(n, in_len, 4)
            exp = P.truncate250(oracle.hash_batch(tag, msg, in_len, out_len, threads=8).reshape(-1, 4)).reshape(n, out_len, 4)
            assert np.array_equal(ctx.hash_batch(tag, msg, in_len, out_len, truncated=True), exp), ("truncated host", n, in_len, out_len, seed)
            d_out = torch.empty((n, out_len, 4), dtype=torch.int64, device="cuda")
            ctx.hash_batch_device(tag, torch.from_numpy(msg.view(np.int64).copy()).cuda(), in_len, out_len, d_out, n, truncated=True)
            torch.cuda.synchronize()
            assert np.array_equal(d_out.cpu().numpy().view(np.uint64).reshape(exp.shape), exp), ("truncated device", n, in_len, out_len, seed)
            counts["fused truncated outputs"] += n * out_len
        elif kind == 13:  # round 5: root-only builds queued on 2 .. 6 streams of ONE context at once (per-stream level scratch)
            import torch
            k = int(rng.integers(2, 7))
            sizes = [int(rng.choice([4 ** int(rng.integers(1, 8)), int(rng.integers(1, 20000))])) for _ in range(k)]
            lvs = [oracle.fill_random(seed + j, m) for j, m in enumerate(sizes)]
            ds = [torch.from_numpy(lv.view(np.int64).copy()).cuda() for lv in lvs]
            d_roots = torch.zeros((k, 4), dtype=torch.int64, device="cuda")
            streams = [torch.cuda.Stream() for _ in range(k)]
            torch.cuda.synchronize()
            for rep in range(3):
                for j in range(k):
                    with torch.cuda.stream(streams[j]):
                        ctx.merkle4_tree_device(mtag, ds[j], sizes[j], d_roots[j], None)
            torch.cuda.synchronize()
            got = d_roots.cpu().numpy().view(np.uint64)
            for j in range(k):
                assert np.array_equal(got[j], oracle.merkle4_tree(mtag, lvs[j])[0]), ("streams", sizes, j, seed)
            counts["root-only builds on concurrent streams"] += 3 * k
        elif kind == 14:  # round 5: Opening::verify in bulk — extract, tamper a random subset, verify against the root, both arities
            import torch
            arity = int(rng.choice([4, 2]))
            leaves_n = int(rng.choice([int(rng.integers(1, 3000)), arity ** int(rng.integers(0, 9 if arity == 4 else 15)), int(rng.integers(3000, 60000))]))
            k = int(rng.integers(1, 6000))
            vtag = mtag if arity == 4 else tag2
            lv = oracle.fill_random(seed, leaves_n)
            root, levels = (ctx.merkle4_tree if arity == 4 else ctx.merkle2_tree)(vtag, lv, want_levels=True)
            assert np.array_equal(root, (oracle.merkle4_tree if arity == 4 else oracle.merkle2_tree)(vtag, lv)[0]), ("verify tree", arity, leaves_n, seed)
            d_lv = torch.from_numpy(lv.view(np.int64).copy()).cuda()
            d_levels = torch.from_numpy(np.ascontiguousarray(levels if levels.shape[0] else np.zeros((1, 4), dtype=np.uint64)).view(np.int64)).cuda()
            idx = rng.integers(0, leaves_n, size=k).astype(np.int32)
            out, sib, pos, depth = ctx.merkle4_openings_device(d_lv, leaves_n, d_levels, torch.from_numpy(idx).cuda(), k, check=True, arity=arity)
            bad = rng.random(k) < 0.3
            h_out = out.cpu().numpy().copy()
            h_out[bad, int(rng.integers(0, 4))] ^= 1 << int(rng.integers(0, 60))  # the leaf of every tampered opening
            ok = torch.zeros(k, dtype=torch.uint8, device="cuda")
            ctx.merkle_verify_batch_device(vtag, torch.from_numpy(h_out).cuda(), sib, pos, depth, torch.from_numpy(root.view(np.int64).copy()).cuda(), ok, k, arity=arity)
            torch.cuda.synchronize()
            assert np.array_equal(ok.cpu().numpy().astype(bool), ~bad), ("verify", arity, leaves_n, k, seed)
            counts["openings verified in bulk"] += k
        elif kind == 10:  # subtree -> ncclAllGather of the roots on the stream -> top levels, inside the library
            import torch
            leaves_n = 4 ** int(rng.integers(0, 10))
            lv = oracle.fill_random(seed, leaves_n)
            d = torch.from_numpy(lv.view(np.int64).copy()).cuda()
            d_root = torch.zeros(4, dtype=torch.int64, device="cuda")
            comm1.merkle4_tree_sharded_device(mtag, d, leaves_n, d_root)
            torch.cuda.synchronize()
            comm1.check()  # (round 6: no peer reported a failed build; waits for the stream itself)
            assert np.array_equal(d_root.cpu().numpy().view(np.uint64), oracle.merkle4_tree(mtag, lv)[0]), ("sharded", leaves_n, seed)
            counts["sharded-tree leaves (RCCL, one rank)"] += leaves_n
        else:
            import torch
            raw = np.frombuffer(np.random.default_rng(seed).bytes(32 * n), dtype=np.uint8).reshape(n, 32)
            d_b = torch.from_numpy(raw.copy()).cuda()
            d_s = torch.empty((n, 4), dtype=torch.int64, device="cuda")
            d_ok = torch.empty(n, dtype=torch.uint8, device="cuda")
            ctx.from_bytes_device(d_b, d_s, n, d_ok)
            d_back = torch.empty_like(d_b)
            ctx.to_bytes_device(d_s, d_back, n)
            torch.cuda.synchronize()
            h_s, h_ok = P.from_bytes(raw)
            assert np.array_equal(d_s.cpu().numpy().view(np.uint64), h_s) and np.array_equal(d_ok.cpu().numpy().astype(bool), h_ok), ("from_bytes", n, seed)
            vals = [int.from_bytes(r.tobytes(), "little") % P_ for r in raw[:64]]
            assert [int.from_bytes(r.tobytes(), "little") for r in d_back.cpu().numpy()[:64]] == vals, ("to_bytes", n, seed)
            assert np.array_equal(d_back.cpu().numpy(), P.to_bytes(h_s))
            counts["bytes"] += n
    print("long soak, %.1f minutes, %d calls, every result equal to the oracle's: %s" % ((time.time() - t0) / 60, it, ", ".join("%s %d" % kv for kv in counts.items())))
