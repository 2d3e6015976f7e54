"""BASELINE.json configs at FULL size on the GPU, checked through size-independent properties
(subtree composition, shard consistency, oracle spot checks), plus a randomized-shape sweep."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_config3_full_2pow24_leaf_tree(gpu_ctx, oracle_mod):
    """2^24-leaf arity-4 tree (512 MiB of leaves, 5,592,405 permutations):
       root == tree over the 16 roots of the 2^20-leaf subtrees == tree over 1024 roots of 2^14-leaf
       subtrees; one 2^14-leaf subtree root is recomputed by the oracle."""
    import torch
    import poseidon252_amd as P
    n = 1 << 24
    tag = oracle_mod.tag(0, [4], 1)
    g = torch.Generator(device="cuda")
    g.manual_seed(24)
    d = torch.randint(0, 2 ** 62, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    root = P.merkle4_tree(d, tag=tag, ctx=gpu_ctx)
    sub20 = torch.stack([P.merkle4_tree(d[i << 20:(i + 1) << 20], tag=tag, ctx=gpu_ctx) for i in range(16)])
    assert torch.equal(P.merkle4_tree(sub20.contiguous(), tag=tag, ctx=gpu_ctx), root)
    # level 7 of the global tree = roots of the 1024 subtrees of 4^7 leaves: one batched level-by-level pass
    lv = d
    for _ in range(7):
        nxt = torch.empty((lv.shape[0] // 4, 4), dtype=torch.int64, device="cuda")
        gpu_ctx.hash_batch_device(tag, lv, 4, 1, nxt, lv.shape[0] // 4)
        lv = nxt
    assert torch.equal(P.merkle4_tree(lv.contiguous(), tag=tag, ctx=gpu_ctx), root)
    torch.cuda.synchronize()
    k = 777  # subtree index
    leaves_k = d[k << 14:(k + 1) << 14].cpu().numpy().view(np.uint64)
    assert np.array_equal(lv[k].cpu().numpy().view(np.uint64), oracle_mod.merkle4_tree(tag, leaves_k)[0])
    assert P.levels_len(n) == 5592405


def test_config2_scaled_up_2pow24_digests(gpu_ctx, oracle_mod):
    """16 x configs[1]: 2^24 digests in one launch (2 GiB in): grid arithmetic, spot parity, shard consistency"""
    import torch
    n = 1 << 24
    tag = oracle_mod.tag(0, [4], 1)
    g = torch.Generator(device="cuda")
    g.manual_seed(2)
    d_in = torch.randint(0, 2 ** 62, (n * 4, 4), dtype=torch.int64, device="cuda", generator=g)
    d_out = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    gpu_ctx.hash_batch_device(tag, d_in, 4, 1, d_out, n)
    idx = torch.tensor([0, 1, 255, 256, n // 2 - 1, n // 2, n - 257, n - 1], device="cuda")
    sel_in = d_in.view(n, 4, 4)[idx].cpu().numpy().view(np.uint64)
    assert np.array_equal(d_out[idx].cpu().numpy().view(np.uint64), oracle_mod.hash_batch(tag, sel_in, 4, 1).reshape(-1, 4))
    lo = (1 << 23) + 12345
    d_out2 = torch.empty((4096, 4), dtype=torch.int64, device="cuda")
    gpu_ctx.hash_batch_device(tag, d_in[lo * 4:(lo + 4096) * 4], 4, 1, d_out2, 4096)
    torch.cuda.synchronize()
    assert torch.equal(d_out2, d_out[lo:lo + 4096])


def test_config2_every_digest_against_the_oracle(gpu_ctx, oracle_mod):
    """configs[1] in full: all 2^20 Merkle4 digests of the seeded batch, limb for limb against the CPU restatement of the
    reference schedule (multi-threaded: a few seconds) — not a sample"""
    import os
    n = 1 << 20
    tag = oracle_mod.tag(0, [4], 1)
    h = oracle_mod.fill_random(0xc10d, 4 * n).reshape(n, 4, 4)
    got = gpu_ctx.hash_batch(tag, h, 4, 1).reshape(n, 4)
    threads = max(1, min(64, len(os.sched_getaffinity(0))))
    exp = oracle_mod.hash_batch(tag, h, 4, 1, threads=threads).reshape(n, 4)
    assert np.array_equal(got, exp)


def test_config4_full_2pow20_sponges(gpu_ctx, oracle_mod):
    """Domain::Other, 2^20 messages x 42 scalars -> 5 outputs (1.3 GiB in, 12 permutations each)"""
    import torch
    import poseidon252_amd as P
    n = 1 << 20
    hb = P.HashBatch(P.Domain.Other, 42, output_len=5, ctx=gpu_ctx)
    g = torch.Generator(device="cuda")
    g.manual_seed(4)
    d_in = torch.randint(0, 2 ** 62, (n * 42, 4), dtype=torch.int64, device="cuda", generator=g)
    out = hb.digest(d_in)
    torch.cuda.synchronize()
    idx = torch.arange(0, n, 8191, device="cuda")
    sel = d_in.view(n, 42, 4)[idx].cpu().numpy().view(np.uint64)
    assert np.array_equal(out[idx].cpu().numpy().view(np.uint64), oracle_mod.hash_batch(hb.tag, sel, 42, 5))


def test_random_shapes_sweep(gpu_ctx, oracle_mod):
    """60 random (n, in_len, out_len) triples, ragged batch sizes, random tags"""
    rng = np.random.default_rng(2026)
    for case in range(60):
        n = int(rng.integers(1, 700))
        in_len = int(rng.integers(1, 23))
        out_len = int(rng.integers(1, 11))
        tag = oracle_mod.fill_random(9000 + case, 1)[0]
        m = oracle_mod.fill_random(10000 + case, n * in_len).reshape(n, in_len, 4)
        got = gpu_ctx.hash_batch(tag, m, in_len, out_len)
        assert np.array_equal(got, oracle_mod.hash_batch(tag, m, in_len, out_len)), (n, in_len, out_len)


def test_two_threads_two_contexts(gpu_ctx, oracle_mod):
    """ABI threading contract: distinct contexts are independent — two host threads hash concurrently
    (ctypes releases the GIL), host-buffer path incl. the pipelined large-batch branch"""
    import threading
    import poseidon252_amd as P
    tag = oracle_mod.tag(0, [4], 1)
    data = [oracle_mod.fill_random(500 + t, 4 * 300000).reshape(300000, 4, 4) for t in range(2)]
    results, errors = [None, None], []

    def work(t):
        try:
            ctx = P.Context(0)
            outs = [ctx.hash_batch(tag, data[t], 4, 1) for _ in range(3)]
            assert all(np.array_equal(outs[0], o) for o in outs[1:])
            results[t] = outs[0]
            ctx.close()
        except Exception as e:  # pragma: no cover
            errors.append(e)

    threads = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for t in range(2):
        idx = np.arange(0, 300000, 1501)
        assert np.array_equal(results[t][idx], oracle_mod.hash_batch(tag, data[t][idx], 4, 1))


def test_long_messages(gpu_ctx, oracle_mod):
    """few, very long messages: 1,025 and 4,099 scalars absorbed sequentially (257 / 1,025 permutations per lane)"""
    for n, in_len, out_len in [(3, 1025, 9), (2, 4099, 1)]:
        tag = oracle_mod.fill_random(31 + in_len, 1)[0]
        m = oracle_mod.fill_random(32 + in_len, n * in_len).reshape(n, in_len, 4)
        assert np.array_equal(gpu_ctx.hash_batch(tag, m, in_len, out_len), oracle_mod.hash_batch(tag, m, in_len, out_len))


def test_two_contexts_and_streams(gpu_ctx, oracle_mod):
    """distinct contexts are independent; work on a non-default torch stream is ordered on that stream"""
    import torch
    import poseidon252_amd as P
    ctx2 = P.Context(0)
    tag = oracle_mod.tag(0, [4], 1)
    x = oracle_mod.fill_random(77, 4 * 5000).reshape(5000, 4, 4)
    exp = oracle_mod.hash_batch(tag, x, 4, 1)
    assert np.array_equal(ctx2.hash_batch(tag, x, 4, 1), exp)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        d = torch.from_numpy(x.view(np.int64)).cuda(non_blocking=True)
        out = torch.empty((5000, 4), dtype=torch.int64, device="cuda")
        ctx2.hash_batch_device(tag, d, 4, 1, out, 5000)
        gpu_ctx.hash_batch_device(tag, d, 4, 1, out, 5000)  # second context, same stream, same answer
    s.synchronize()
    assert np.array_equal(out.cpu().numpy().view(np.uint64).reshape(5000, 1, 4), exp)
    ctx2.close()


def test_config5_per_gpu_share_2pow24_leaves_all_levels(gpu_ctx, oracle_mod):
    """BASELINE configs[4] = 8 x this: ONE GPU's full share of the 2^27-leaf sharded tree — 2^24 leaves (512 MiB), ALL
    levels written (5,592,405 nodes, 171 MiB) — generated on the device exactly as bench.py's rank 3 generates it
    (SURVEY §8d generator, seed 0xc10d + rank).  Every one of the 12 levels is checked against the oracle on a random
    sample of nodes (each node = Hash::digest(Merkle4, its 4 children in the level below)), the top 3 levels in full;
    then the 8-way composition of configs[4]: roots of 8 such-shaped subtrees (of 4^6 leaves, to stay short) -> 2 -> 1
    through the multi-context ABI equals the tree over the concatenation."""
    import torch
    import poseidon252_amd as P
    from poseidon252_amd import multi, synth
    n = 1 << 24
    tag = P.merkle4_tag()
    d = synth.splitmix_scalars(0xC10D + 3, n, "cuda")
    assert np.array_equal(d[:1000].cpu().numpy().view(np.uint64), oracle_mod.fill_random(0xC10D + 3, 1000))  # the bytes bench.py times
    total = P.levels_len(n)
    assert total == 5592405
    d_levels = torch.empty((total, 4), dtype=torch.int64, device="cuda")
    d_root = torch.empty(4, dtype=torch.int64, device="cuda")
    gpu_ctx.merkle4_tree_device(tag, d, n, d_root, d_levels)
    torch.cuda.synchronize()
    rng = np.random.default_rng(5)
    below, off, cnt = d, 0, n // 4
    while cnt >= 1:
        cur = d_levels[off:off + cnt]
        idx = np.unique(rng.integers(0, cnt, size=min(cnt, 96))) if cnt > 64 else np.arange(cnt)
        t_idx = torch.from_numpy(idx).cuda()
        children = below.view(-1, 4, 4)[t_idx].cpu().numpy().view(np.uint64)
        exp = oracle_mod.hash_batch(tag, children, 4, 1).reshape(-1, 4)
        assert np.array_equal(cur[t_idx].cpu().numpy().view(np.uint64), exp), "level with %d nodes" % cnt
        below, off, cnt = cur, off + cnt, cnt // 4
    assert off == total and torch.equal(d_root, d_levels[-1])
    # root-only build (ping-pong scratch, what bench.py --workload tree times) gives the same root
    d_root2 = torch.empty(4, dtype=torch.int64, device="cuda")
    gpu_ctx.merkle4_tree_device(tag, d, n, d_root2, None)
    torch.cuda.synchronize()
    assert torch.equal(d_root, d_root2)
    # configs[4] composition at 8 shards: gather of 8 roots, then [r0..r3], [r4..r7] -> 2 nodes -> [n0, n1, 0, 0] -> root
    per = 4 ** 6
    sub = d[:8 * per].cpu().numpy().view(np.uint64)
    ctxs = [P.Context(0) for _ in range(8)]
    root8 = multi.merkle4_tree_multi(ctxs, tag, sub)
    for c in ctxs:
        c.close()
    assert np.array_equal(root8, oracle_mod.merkle4_tree(tag, sub)[0])
    roots = np.stack([P.merkle4_tree(sub[i * per:(i + 1) * per], tag=tag, ctx=gpu_ctx) for i in range(8)])
    zero = np.zeros((2, 4), dtype=np.uint64)
    n01 = oracle_mod.hash_batch(tag, roots.reshape(2, 4, 4), 4, 1).reshape(2, 4)
    top = oracle_mod.hash_batch(tag, np.concatenate([n01, zero]).reshape(1, 4, 4), 4, 1).reshape(4)
    assert np.array_equal(root8, top)  # 8 subtrees + 3 top permutations: 44,739,243 in all at full size


@pytest.mark.parametrize("log2n", [28, 30, 32])
def test_beyond_4GiB_leaves_in_one_buffer(gpu_ctx, oracle_mod, log2n):
    """(The default set runs log2n = 28: 8 GiB of leaves in one buffer — byte offsets beyond 2^32 on every array of the path — in a few
    seconds.  log2n = 30 (32 GiB + 10 GiB of scratch; in the default set until round 5, 40 s of the driver's 1,200 s) and log2n = 32
    (2^32 leaves = 128 GiB in one buffer, ~180 GiB in all, more leaves than a uint32 counts) run with P252_TEST_HUGE=1: VERDICT r5 item 3;
    both ran on an MI355X, profiles/r05_huge_sizes.txt.)
    Maximum sizes (MI355X: 288 GB of HBM per GPU — shards are sized for it): 2^30 leaves = 32 GiB in ONE buffer, 64 x one GPU's
    share of BASELINE configs[4].  Every index on this path must be 64-bit: 2^28 digests in one launch (8 GiB out), the 15-level
    tree over the 2^30 leaves (357,913,941 permutations, root only: 10 GiB of per-stream level scratch) and the forest of 2^18 trees
    of 4^6 leaves over the same buffer.  Checked through size-independent properties and oracle spot checks at both ends of the
    buffer: tree(leaves) == tree(digests of the groups of four) == tree(forest roots); digests and forest roots against the oracle."""
    import torch
    import poseidon252_amd as P
    import os
    if log2n > 28 and os.environ.get("P252_TEST_HUGE") != "1":
        pytest.skip("2^%d leaves (%d GiB): set P252_TEST_HUGE=1" % (log2n, (32 << log2n) >> 30))
    n = 1 << log2n
    free, _ = torch.cuda.mem_get_info()
    if free < int(1.8 * n * 32):
        pytest.skip("needs %d GiB of free HBM" % (int(1.8 * n * 32) >> 30))
    tag = oracle_mod.tag(0, [4], 1)
    g = torch.Generator(device="cuda")
    g.manual_seed(30)
    d = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    for c in range(0, n, 1 << 27):  # (limbs below 2^62: every scalar is below p)
        d[c:c + (1 << 27)] = torch.randint(0, 2 ** 62, (1 << 27, 4), dtype=torch.int64, device="cuda", generator=g)
    root = P.merkle4_tree(d, tag=tag, ctx=gpu_ctx)
    # level 1 on its own: 2^28 digests in one launch, 8 GiB of output
    lvl1 = torch.empty((n // 4, 4), dtype=torch.int64, device="cuda")
    gpu_ctx.hash_batch_device(tag, d, 4, 1, lvl1, n // 4)
    assert torch.equal(P.merkle4_tree(lvl1, tag=tag, ctx=gpu_ctx), root)
    idx = torch.tensor([0, 1, 12345, (n >> 3) + 5, (n >> 2) - 2, (n >> 2) - 1], device="cuda")  # incl. the last items: byte offsets > 2^35
    got = lvl1[idx].cpu().numpy().view(np.uint64)
    inp = d.view(n // 4, 4, 4)[idx].cpu().numpy().view(np.uint64)
    assert np.array_equal(got, oracle_mod.hash_batch(tag, inp, 4, 1).reshape(-1, 4))
    del lvl1
    # the same leaves as a forest of 2^18 trees of 4^6 leaves (one launch per level across all trees)
    per = 4 ** 6
    roots = P.merkle4_forest(d, per, tag=tag, ctx=gpu_ctx)
    assert roots.shape[0] == n // per and torch.equal(P.merkle4_tree(roots.contiguous(), tag=tag, ctx=gpu_ctx), root)
    for t in (0, min(77777, n // per - 2), n // per - 1):
        leaves_t = d[t * per:(t + 1) * per].cpu().numpy().view(np.uint64)
        assert np.array_equal(roots[t].cpu().numpy().view(np.uint64), oracle_mod.merkle4_tree(tag, leaves_t)[0]), t
    assert P.levels_len(n) == (4 ** (log2n // 2) - 1) // 3
    del d, roots
    torch.cuda.empty_cache()
