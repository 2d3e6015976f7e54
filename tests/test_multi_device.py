"""p252_*_multi: the in-library multi-device entry points (SURVEY §8b sketch, §8e).  One GPU here, so the contexts of
an array all sit on device 0 — the code path (one host thread + context per shard, contiguous shards, 32-byte roots
gathered on the host, top levels on ctxs[0]) is the one an 8-GPU node runs with one context per device."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctxs(gpu_ctx):
    import poseidon252_amd as P
    cs = [P.Context(0) for _ in range(8)]
    yield cs
    for c in cs:
        c.close()


def test_hash_batch_multi_equals_single_context(gpu_ctx, ctxs, oracle_mod):
    import poseidon252_amd as P
    from poseidon252_amd import multi
    tag = P.HashBatch(P.Domain.Merkle4, 4, ctx=gpu_ctx).tag
    for k, n in ((2, 100003), (3, 7), (8, 8), (8, 5), (2, (1 << 19) + 3)):  # ragged, fewer items than devices, staged path
        x = oracle_mod.fill_random(4000 + n % 1000, 4 * n).reshape(n, 4, 4)
        got = multi.hash_batch_multi(ctxs[:k], tag, x, 4, 1)
        assert np.array_equal(got, gpu_ctx.hash_batch(tag, x, 4, 1))
        idx = np.arange(0, n, max(1, n // 300))
        assert np.array_equal(got[idx], oracle_mod.hash_batch(tag, x[idx], 4, 1))
    hb = P.HashBatch(P.Domain.Other, 42, output_len=5, ctx=gpu_ctx)
    m = oracle_mod.fill_random(4242, 42 * 999).reshape(999, 42, 4)
    assert np.array_equal(multi.hash_batch_multi(ctxs[:4], hb.tag, m, 42, 5), oracle_mod.hash_batch(hb.tag, m, 42, 5))


def test_sharded_tree_root_equals_tree_over_concatenation(gpu_ctx, ctxs, oracle_mod):
    """BASELINE configs[4] structure: n_ctx complete subtrees, roots gathered, top levels zero-padded"""
    import poseidon252_amd as P
    from poseidon252_amd import multi
    tag = P.merkle4_tag()
    for k, per in ((2, 4 ** 5), (8, 4 ** 4), (4, 4 ** 3), (3, 4 ** 2), (1, 4 ** 4), (8, 1)):
        lv = oracle_mod.fill_random(700 + k + per, k * per)
        root = multi.merkle4_tree_multi(ctxs[:k], tag, lv)
        assert np.array_equal(root, P.merkle4_tree(lv, tag=tag, ctx=gpu_ctx))
        assert np.array_equal(root, oracle_mod.merkle4_tree(tag, lv)[0])


def test_multi_device_resident_variants(gpu_ctx, ctxs, oracle_mod):
    import torch
    import poseidon252_amd as P
    from poseidon252_amd import multi
    tag = P.merkle4_tag()
    k, per = 4, 4 ** 6
    lv = oracle_mod.fill_random(31337, k * per)
    d = [torch.from_numpy(lv[t * per:(t + 1) * per].view(np.int64)).cuda() for t in range(k)]
    assert np.array_equal(multi.merkle4_tree_multi_device(ctxs[:k], tag, d, per), oracle_mod.merkle4_tree(tag, lv)[0])
    outs = [torch.empty((per // 4, 4), dtype=torch.int64, device="cuda") for _ in range(k)]
    multi.hash_batch_multi_device(ctxs[:k], tag, d, 4, 1, outs, [per // 4] * k)
    torch.cuda.synchronize()
    got = torch.cat(outs).cpu().numpy().view(np.uint64)
    assert np.array_equal(got, oracle_mod.hash_batch(tag, lv.reshape(-1, 4, 4), 4, 1).reshape(-1, 4))


def test_multi_argument_errors(gpu_ctx, ctxs, oracle_mod):
    import poseidon252_amd as P
    from poseidon252_amd import multi
    tag = P.merkle4_tag()
    lv = oracle_mod.fill_random(1, 48)
    with pytest.raises(ValueError):
        multi.merkle4_tree_multi(ctxs[:4], tag, lv)           # 12 leaves per device: not 4^k
    with pytest.raises(ValueError):
        multi.merkle4_tree_multi(ctxs[:5], tag, lv)           # 48 % 5 != 0
    with pytest.raises(ValueError):
        multi.merkle4_tree_multi(ctxs[:2], tag, lv)           # 24 leaves per device: not 4^k
    with pytest.raises(ValueError):
        multi.hash_batch_multi([ctxs[0], ctxs[0]], tag, lv[:8].reshape(2, 4, 4), 4, 1)  # a context twice
    with pytest.raises(P.InvalidIOPattern):
        multi.hash_batch_multi(ctxs[:2], tag, lv[:8].reshape(2, 4, 4), 4, 0)
