"""Host-buffer entry points on large batches take the pipelined path (3 streams, chunked H2D / kernel /
D2H, buffers page-locked for the call or allocated with p252_host_alloc): results must equal the
serial path and the oracle, for pageable and pinned memory, ragged sizes and multi-output sponges."""
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
    p252_host_alloc memory by the host-buffer entry points; results unchanged, no per-call page-locking"""
    import time
    import poseidon252_amd as P
    from poseidon252_amd import _lib
    from poseidon252_amd.hash import registered
    hb = P.HashBatch(P.Domain.Merkle4, 4, ctx=gpu_ctx)
    n = (1 << 19) + 5
    x = oracle_mod.fill_random(0x7e9, 4 * n).reshape(n, 4, 4)
    ref = hb.digest(x)
    out = np.empty((n, 1, 4), dtype=np.uint64)
    with registered(x), registered(out):
        hb.digest(x, out=out)  # warm
        t0 = time.perf_counter()
        got = hb.digest(x, out=out)
        t_reg = time.perf_counter() - t0
    assert np.array_equal(got.reshape(ref.shape), ref)
    t0 = time.perf_counter()
    hb.digest(x, out=out)
    t_unreg = time.perf_counter() - t0
    assert t_reg < t_unreg  # no per-call page-locking while registered
    assert _lib.lib().p252_host_register(None, 16) != 0 and _lib.lib().p252_host_unregister(None) != 0  # argument checks
