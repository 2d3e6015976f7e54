"""Host-buffer entry points on large batches take a pipelined path: pageable caller memory goes through the library's
page-locked staging lanes (worker threads: memcpy in / H2D / kernel / D2H / memcpy out), memory that is page-locked on
both sides (p252_host_alloc / p252_host_register) is DMA'd in place over 3 streams.  Results must equal the serial
path and the oracle, for pageable and pinned memory, ragged sizes and multi-output sponges."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_pipelined_host_path_parity(gpu_ctx, oracle_mod):
    import poseidon252_amd as P
    from poseidon252_amd.hash import PinnedScalars
    hb = P.HashBatch(P.Domain.Merkle4, 4, ctx=gpu_ctx)
    n = (1 << 18) + 777  # > 2 chunks of 2^17 digests, ragged tail
    x = oracle_mod.fill_random(0x51e, 4 * n).reshape(n, 4, 4)
    out = hb.digest(x)  # pageable
    idx = np.concatenate([np.arange(0, n, 997), [n - 1, (1 << 17) - 1, 1 << 17, (1 << 18) - 1, 1 << 18]])
    assert np.array_equal(out[idx], oracle_mod.hash_batch(hb.tag, x[idx], 4, 1))
    pin_in, pin_out = PinnedScalars(4 * n), PinnedScalars(n)
    pin_in.array[:] = x.reshape(-1, 4)
    out2 = gpu_ctx.hash_batch(hb.tag, pin_in.array, 4, 1, out=pin_out.array)
    assert np.array_equal(out2, out)
    assert np.array_equal(pin_in.array, x.reshape(-1, 4))  # inputs are borrowed, never modified
    pin_in.free()
    pin_out.free()


def test_pipelined_sponge_multi_output(gpu_ctx, oracle_mod):
    import poseidon252_amd as P
    hb = P.HashBatch(P.Domain.Other, 42, output_len=5, ctx=gpu_ctx)
    n = 30000  # 42*32 B per message -> chunks of ~12.4k messages: 3 chunks
    m = oracle_mod.fill_random(0x5b0, 42 * n).reshape(n, 42, 4)
    out = hb.digest(m)
    idx = np.arange(0, n, 131)
    assert np.array_equal(out[idx], oracle_mod.hash_batch(hb.tag, m[idx], 42, 5))


def test_caller_registered_buffers(gpu_ctx, oracle_mod):
    """p252_host_register / p252_host_unregister: a caller-owned (numpy) buffer page-locked once is treated like
    p252_host_alloc memory by the host-buffer entry points (zero-copy DMA); results unchanged"""
    import poseidon252_amd as P
    from poseidon252_amd import _lib
    from poseidon252_amd.hash import registered
    hb = P.HashBatch(P.Domain.Merkle4, 4, ctx=gpu_ctx)
    n = (1 << 19) + 5
    x = oracle_mod.fill_random(0x7e9, 4 * n).reshape(n, 4, 4)
    ref = hb.digest(x)
    out = np.empty((n, 1, 4), dtype=np.uint64)
    with registered(x), registered(out):
        got = hb.digest(x, out=out).copy()
    assert np.array_equal(got.reshape(ref.shape), ref)
    out[:] = 0
    assert np.array_equal(hb.digest(x, out=out).reshape(ref.shape), ref)  # unregistered again: staged path, same bytes
    assert _lib.lib().p252_host_register(None, 16) != 0 and _lib.lib().p252_host_unregister(None) != 0  # argument checks


def test_staged_lanes_mixed_memory_and_lane_counts(gpu_ctx, oracle_mod):
    """pageable on one side only, and 1 / 3 / default staging lanes (P252_HOST_LANES is read once per process: the lane
    count is varied through batch sizes that need fewer chunks than lanes)"""
    import poseidon252_amd as P
    from poseidon252_amd.hash import PinnedScalars
    hb = P.HashBatch(P.Domain.Merkle4, 4, ctx=gpu_ctx)
    for n in ((1 << 18) + 1, 3 * (1 << 17) + 100, (1 << 21) + 12345):  # 3, 4 and 17 chunks of 2^17 digests
        x = oracle_mod.fill_random(0x600 + n % 97, 4 * n).reshape(n, 4, 4)
        ref = hb.digest(x)
        idx = np.concatenate([np.arange(0, n, 4999), [n - 1, (1 << 17) - 1, 1 << 17]])
        assert np.array_equal(ref[idx], oracle_mod.hash_batch(hb.tag, x[idx], 4, 1))
        pin_in = PinnedScalars(4 * n)
        pin_in.array[:] = x.reshape(-1, 4)
        got = gpu_ctx.hash_batch(hb.tag, pin_in.array, 4, 1)  # pinned in, pageable out -> staged
        assert np.array_equal(got.reshape(ref.shape), ref)
        pin_in.free()
    # a second context staging concurrently must not disturb the first (lanes are per context)
    ctx2 = P.Context(0)
    assert np.array_equal(ctx2.hash_batch(hb.tag, x, 4, 1).reshape(ref.shape), ref)
    ctx2.close()


def test_tree_from_pageable_host_leaves_streams_the_first_level(gpu_ctx, oracle_mod):
    """p252_merkle4_tree / p252_merkle2_tree on big pageable leaf arrays: the first level is hashed chunk by chunk while
    the leaves stream in through the staging lanes; root and all levels must equal the device-resident build"""
    import torch
    import poseidon252_amd as P
    tag = P.merkle4_tag()
    for n in (1 << 20, (1 << 20) + 4 * 70000, 1 << 21):  # 4 / 5+ / 8 chunks of 2^16 level-1 nodes, ragged last chunk
        lv = oracle_mod.fill_random(0xabc0 + n % 1000, n)
        root, levels = gpu_ctx.merkle4_tree(tag, lv, want_levels=True)
        d = torch.from_numpy(lv.view(np.int64)).cuda()
        d_root, d_levels = P.merkle4_tree(d, tag=tag, ctx=gpu_ctx, want_levels=True)
        assert np.array_equal(root, d_root.cpu().numpy().view(np.uint64))
        assert np.array_equal(levels, d_levels.cpu().numpy().view(np.uint64))
        assert np.array_equal(gpu_ctx.merkle4_tree(tag, lv), root)  # root-only variant (level 1 in scratch)
        idx = np.arange(0, n // 4, 9973)
        assert np.array_equal(levels[idx], oracle_mod.hash_batch(tag, lv.reshape(-1, 4, 4)[idx], 4, 1).reshape(-1, 4))
    # arity 2 through the same pipeline.  (Until round 5 the full 2^20-leaf tree was rebuilt by the ORACLE: 1,048,575 permutations on one
    # core, 47 s of the suite's 440.  Now: every level of the streamed build is the device kernels' digest of the level below — the
    # Merkle2 digest kernel is compared with the oracle in tests/test_gpu_parity.py and again here on a strided sample of every
    # level — and a 2^14-leaf tree, four chunks' worth of pairs at P252_HOST_CHUNK_MB's floor, equals the oracle's node for node.)
    tag2 = oracle_mod.tag(1, [2], 1)
    lv = oracle_mod.fill_random(0xabc9, 1 << 20)
    r2, l2 = gpu_ctx.merkle2_tree(tag2, lv, want_levels=True)
    assert np.array_equal(gpu_ctx.merkle2_tree(tag2, lv), r2)
    below, off, cnt = lv, 0, 1 << 20
    while cnt > 1:
        cnt //= 2
        level = l2[off:off + cnt]
        assert np.array_equal(level, gpu_ctx.hash_batch(tag2, below.reshape(cnt, 2, 4), 2, 1).reshape(cnt, 4)), cnt
        idx = np.arange(0, cnt, max(1, cnt // 97))
        assert np.array_equal(level[idx], oracle_mod.hash_batch(tag2, np.ascontiguousarray(below.reshape(cnt, 2, 4)[idx]), 2, 1).reshape(-1, 4)), cnt
        below, off = level, off + cnt
    assert np.array_equal(r2, l2[-1])
    small = oracle_mod.fill_random(0xabca, 1 << 14)
    rs, ls = gpu_ctx.merkle2_tree(tag2, small, want_levels=True)
    o2 = oracle_mod.merkle2_tree(tag2, small, want_levels=True)
    assert np.array_equal(rs, o2[0]) and np.array_equal(ls, o2[1])


def test_other_host_entry_points_stream_large_batches(gpu_ctx, oracle_mod):
    """p252_permute_batch, p252_merkle4_path_batch, p252_encrypt_batch / p252_decrypt_batch on batches large enough to go
    through the staging lanes (several chunks, ragged tail): same bytes as the oracle on a sample and as small calls"""
    import poseidon252_amd as P
    from poseidon252_amd.encryption import decrypt_batch, encrypt_batch, encryption_tag
    from poseidon252_amd.merkle import merkle4_path_roots
    # permutations: 160 B in / out per item, chunks of 52,224 items
    n = 3 * 52224 + 1234
    st = oracle_mod.fill_random(0x9e1, 5 * n).reshape(n, 5, 4)
    got = gpu_ctx.permute_batch(st)
    idx = np.concatenate([np.arange(0, n, 7919), [52223, 52224, n - 1]])
    assert np.array_equal(got[idx], oracle_mod.permute_batch(st[idx]))
    assert np.array_equal(got[:1000], gpu_ctx.permute_batch(st[:1000]))
    # openings of depth 12: leaves + 36 siblings + 12 position bytes per item, chunks of 6,656 items
    n, depth = 4 * 6656 + 77, 12
    tag = P.merkle4_tag()
    leaves = oracle_mod.fill_random(0x9e2, n)
    sibs = oracle_mod.fill_random(0x9e3, n * depth * 3).reshape(n, depth, 3, 4)
    pos = np.random.default_rng(5).integers(0, 4, size=(n, depth), dtype=np.uint8)
    roots = merkle4_path_roots(leaves, sibs, pos, tag=tag, ctx=gpu_ctx)
    idx = np.concatenate([np.arange(0, n, 997), [6655, 6656, n - 1]])
    assert np.array_equal(roots[idx], oracle_mod.merkle4_path_batch(tag, leaves[idx], sibs[idx], pos[idx]))
    # encryption of 2-scalar messages (both variants coincide at len 2) and of 21-scalar messages (STREAM)
    for ln, n in ((2, 2 * 32768 + 999), (21, 3 * 5376 + 5)):
        msgs = oracle_mod.fill_random(0x9e4 + ln, n * ln).reshape(n, ln, 4)
        secrets = oracle_mod.fill_random(0x9e5 + ln, 2 * n).reshape(n, 2, 4)
        nonces = oracle_mod.fill_random(0x9e6 + ln, n)
        cph = encrypt_batch(msgs, secrets, nonces, ctx=gpu_ctx)
        idx = np.concatenate([np.arange(0, n, 1009), [n - 1]])
        assert np.array_equal(cph[idx], oracle_mod.encrypt_batch(encryption_tag(ln), msgs[idx], secrets[idx], nonces[idx]))
        bad = cph.copy()
        bad[::5000, 0, 0] ^= np.uint64(1)
        dec, ok = decrypt_batch(bad, secrets, nonces, ctx=gpu_ctx)
        assert not ok[::5000].any() and ok.sum() == n - len(ok[::5000])
        good = np.ones(n, dtype=bool)
        good[::5000] = False
        assert np.array_equal(dec[good], msgs[good])
