"""Round 6: the grow-only scratch can be given back (p252_trim; the reference holds no state,
src/hash.rs:92-96); extraction of openings is exact beyond 2^24 openings per call (round 5's reciprocal wrapped there); RCCL is
resolved at run time and is the ONE copy the process holds."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dev(x):
    import torch
    return torch.from_numpy(np.ascontiguousarray(x).view(np.int64)).to("cuda:0")


def _host(t):
    return t.cpu().numpy().view(np.uint64)


def test_trim_returns_the_scratch_of_a_2pow26_leaf_root_only_build(oracle_mod):
    """a 2^26-leaf root-only build leaves 5/16 of 2 GiB = 640 MiB of level scratch in the context; after
    p252_trim the device's free memory is back to within 64 MiB of where it was before the build, the context still works
    (same root again, equal to the all-levels build's through caller-owned memory), and a trimmed context holds no residue"""
    import torch
    import poseidon252_amd as P
    tag = P.merkle4_tag()
    n = 1 << 26
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    # the leaves: 2 GiB of valid scalars made on the device from 2^20 oracle scalars (digests of digests are scalars < p)
    seed = _dev(oracle_mod.fill_random(0x26, 1 << 20))
    d = seed.repeat(64, 1).contiguous()
    d[:: 1 << 14] = seed[: d[:: 1 << 14].shape[0]].flip(0)  # (break the exact periodicity a little)
    ctx = P.Context(0)
    try:
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        r1 = torch.zeros(4, dtype=torch.int64, device="cuda:0")
        ctx.merkle4_tree_device(tag, d, n, r1, None)
        torch.cuda.synchronize()
        held = free0 - torch.cuda.mem_get_info()[0]
        assert held >= (n // 4 + n // 16) * 32 - (64 << 20), "the build did not use context-owned level scratch? held %d" % held
        assert ctx.scratch_residue() > 0
        ctx.trim()
        free1 = torch.cuda.mem_get_info()[0]
        assert abs(free0 - free1) <= 64 << 20, "p252_trim left %d bytes allocated" % (free0 - free1)
        assert ctx.scratch_residue() == 0
        # the context works as before: root-only again (scratch re-allocated), and through caller-owned levels
        r2 = torch.zeros(4, dtype=torch.int64, device="cuda:0")
        ctx.merkle4_tree_device(tag, d, n, r2, None)
        lv = torch.empty((P.levels_len(n), 4), dtype=torch.int64, device="cuda:0")
        r3 = torch.zeros(4, dtype=torch.int64, device="cuda:0")
        ctx.merkle4_tree_device(tag, d, n, r3, lv)
        torch.cuda.synchronize()
        assert torch.equal(r1, r2) and torch.equal(r1, r3) and bool(r1.any())
        # against the oracle where it finishes in seconds: the subtree over the first 4^8 leaves is the stored level-8 node 0
        sub = oracle_mod.merkle4_tree(tag, _host(d[: 4 ** 8]))[0]
        off = sum(n // 4 ** l for l in range(1, 8))
        assert np.array_equal(_host(lv[off]), sub)
        # host-buffer calls after a trim (staging lanes, device scratch, the encryption call table are all made again)
        from poseidon252_amd import encryption as Enc
        m = oracle_mod.fill_random(0x27, 300 * 3).reshape(300, 3, 4)
        s = oracle_mod.fill_random(0x28, 600).reshape(300, 2, 4)
        nn = oracle_mod.fill_random(0x29, 300)
        c1 = Enc.encrypt_batch(m, s, nn, ctx=ctx)
        ctx.trim()
        assert np.array_equal(Enc.encrypt_batch(m, s, nn, ctx=ctx), c1)
        big = oracle_mod.fill_random(0x2A, 4 * (1 << 19)).reshape(-1, 4, 4)
        h1 = P.HashBatch(P.Domain.Merkle4, 4, ctx=ctx).digest(big)
        ctx.trim()
        assert np.array_equal(P.HashBatch(P.Domain.Merkle4, 4, ctx=ctx).digest(big), h1)
        ctx.trim()
        ctx.trim()  # idempotent
        assert abs(free0 - torch.cuda.mem_get_info()[0] - lv.numel() * 8 - 3 * 32) <= 64 << 20
    finally:
        ctx.close()


def _expected_openings(torch, d_lv, d_levels, n_leaves, idx, arity):
    """plain torch indexing on the device (no library code): siblings (k, depth, arity - 1, 4), positions (k, depth)"""
    shift = 2 if arity == 4 else 1
    per, cnt, off = [d_lv.view(-1, 4)], n_leaves, 0
    while cnt > 1:
        cnt = (cnt + arity - 1) // arity
        per.append(d_levels.view(-1, 4)[off:off + cnt])
        off += cnt
    depth = len(per) - 1
    k = idx.shape[0]
    sib = torch.zeros((k, depth, arity - 1, 4), dtype=torch.int64, device=idx.device)
    pos = torch.zeros((k, depth), dtype=torch.uint8, device=idx.device)
    cur = idx.to(torch.int64)
    for l in range(depth):
        p = cur & (arity - 1)
        base = cur - p
        pos[:, l] = p.to(torch.uint8)
        for s in range(arity - 1):
            j = base + s + (s >= p).to(torch.int64)
            ok = j < per[l].shape[0]
            v = per[l][j.clamp(max=per[l].shape[0] - 1)]
            sib[:, l, s] = torch.where(ok[:, None], v, torch.zeros_like(v))
        cur = cur >> shift
    return sib, pos


@pytest.mark.parametrize("n_leaves,arity", [(64, 4), (4, 4), (50, 4), (8, 2)])
def test_extraction_is_exact_beyond_2pow24_openings_in_one_call(gpu_ctx, oracle_mod, n_leaves, arity):
    """the FAST extraction kernel mapped lane -> (opening, level) with a 40-bit reciprocal that wrapped from
    opening 2^24 on: every opening beyond it was written with another opening's data, silently, and nothing tested such a k.
    2^24 + 5,000 openings of small trees (depth 3, 1 and a ragged 3; arity 2 depth 3), every record compared ON THE DEVICE with plain
    torch indexing; the tail re-hashed to the root."""
    import torch
    import poseidon252_amd as P
    tag = P.merkle4_tag() if arity == 4 else P.compute_tag(P.Domain.Merkle2, [2], 1)
    lv = oracle_mod.fill_random(0xA0 + n_leaves, n_leaves)
    d_lv = _dev(lv)
    if arity == 4:
        root, d_levels = P.merkle4_tree(d_lv, tag=tag, ctx=gpu_ctx, want_levels=True)
        assert np.array_equal(_host(root), oracle_mod.merkle4_tree(tag, lv)[0])
    else:
        o_root, o_levels, _ = oracle_mod.merkle2_tree(tag, lv, want_levels=True)
        h_root, h_levels = gpu_ctx.merkle2_tree(tag, lv, want_levels=True)
        assert np.array_equal(h_root, o_root) and np.array_equal(h_levels, o_levels)
        root, d_levels = _dev(h_root), _dev(h_levels)
    k = (1 << 24) + 5000
    g = torch.Generator(device="cuda:0")
    g.manual_seed(n_leaves)
    idx = torch.randint(0, n_leaves, (k,), dtype=torch.int32, device="cuda:0", generator=g)
    idx[(1 << 24) - 2:(1 << 24) + 2] = torch.tensor([0, n_leaves - 1, 1 % n_leaves, n_leaves - 1], dtype=torch.int32, device="cuda:0")
    out, sib, pos, depth = gpu_ctx.merkle4_openings_device(d_lv, n_leaves, d_levels, idx, k, check=True, arity=arity)
    torch.cuda.synchronize()
    e_sib, e_pos = _expected_openings(torch, d_lv, d_levels, n_leaves, idx, arity)
    assert depth == e_pos.shape[1]
    assert torch.equal(out.view(k, 4), d_lv.view(-1, 4)[idx.to(torch.int64)])
    bad = (sib.view(e_sib.shape) != e_sib).flatten(1).any(1) | (pos.view(e_pos.shape) != e_pos).any(1)
    n_bad = int(bad.sum())
    assert n_bad == 0, "%d openings differ, the first at opening %d" % (n_bad, int(torch.nonzero(bad)[0]))
    del e_sib, e_pos, bad
    # the last 2^16 openings (all beyond the old wrap point) re-hash to the root
    m = 1 << 16
    ok = torch.zeros(m, dtype=torch.uint8, device="cuda:0")
    gpu_ctx.merkle_verify_batch_device(tag, out.view(k, 4)[k - m:].contiguous(), sib.view(k, -1)[k - m:].contiguous(), pos.view(k, -1)[k - m:].contiguous(),
                                       depth, root, ok, m, arity=arity)
    torch.cuda.synchronize()
    assert bool(ok.all())


def test_real_rccl_is_the_copy_the_process_holds(gpu_ctx, oracle_mod):
    """on the GPU box, with torch imported: the library's communicator calls go to torch's bundled librccl (found already mapped,
    csrc/rccl_dyn.hpp) and the process maps exactly one RCCL — before and after a communicator has been made and used"""
    import torch
    import poseidon252_amd as P
    from poseidon252_amd import comm as C
    b = C.backend()
    ctx = P.Context(0)
    c = C.Comm.create_rank(ctx, 0, 1, lambda i: i)
    tag = P.merkle4_tag()
    lv = oracle_mod.fill_random(0xC6, 4 ** 5)
    d_root = torch.zeros(4, dtype=torch.int64, device="cuda:0")
    c.merkle4_tree_sharded_device(tag, _dev(lv), 4 ** 5, d_root)
    c.check()  # a healthy build: nothing to report
    assert np.array_equal(_host(d_root), oracle_mod.merkle4_tree(tag, lv)[0])
    ctx.sync()  # p252_sync reports peers' failures too: none
    c.destroy()
    ctx.close()
    maps = sorted(set(l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l))
    assert len(maps) == 1 and os.path.realpath(maps[0]) == os.path.realpath(b), (maps, b)
    assert os.sep + "torch" + os.sep in b, b
