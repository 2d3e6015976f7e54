"""batched encryption / decryption (src/encryption.rs:62-95).  Mirrors the reference's
own tests (tests/encryption.rs:30-115): round trip, wrong shared secret, wrong nonce, tampered cipher ->
DecryptionFailed.  Those tests hold no fixed values and dusk-safe is un-vendored, so the byte-level recipe
is UNPINNED; what IS checked: GPU == oracle bit-exact, and the sponge mechanics against the big-int model."""
import numpy as np
import pytest

import pymodel


def _inputs(oracle_mod, n, ln, seed):
    msgs = oracle_mod.fill_random(seed, n * ln).reshape(n, ln, 4)
    secrets = oracle_mod.fill_random(seed + 1, 2 * n).reshape(n, 2, 4)
    nonces = oracle_mod.fill_random(seed + 2, n)
    return msgs, secrets, nonces


# ------------------------------------------------------------------ CPU
@pytest.mark.parametrize("ln", [1, 2, 4, 5, 8, 10])
def test_oracle_round_trip_and_failures(oracle_mod, ln):
    tag = oracle_mod.encryption_tag(ln)
    msgs, secrets, nonces = _inputs(oracle_mod, 6, ln, 100 + ln)
    cph = oracle_mod.encrypt_batch(tag, msgs, secrets, nonces)
    assert cph.shape == (6, ln + 1, 4)
    dec, ok = oracle_mod.decrypt_batch(tag, cph, secrets, nonces)
    assert ok.all() and np.array_equal(dec, msgs)
    assert not oracle_mod.decrypt_batch(tag, cph, secrets[::-1].copy(), nonces)[1].any()  # wrong secret (encryption.rs test 2)
    assert not oracle_mod.decrypt_batch(tag, cph, secrets, nonces[::-1].copy())[1].any()   # wrong nonce
    bad = cph.copy()
    bad[:, 0, 0] ^= np.uint64(1)
    assert not oracle_mod.decrypt_batch(tag, bad, secrets, nonces)[1].any()                # tampered cipher


def test_oracle_encryption_matches_bigint_sponge(oracle_mod):
    """duplex use of the sponge state machine, recomputed with Python big ints"""
    P = pymodel.P
    C, M = pymodel.load_constants()
    perm = lambda s: pymodel.perm_reference(s, C, M)
    for ln in (3, 4, 6):
        tag = oracle_mod.encryption_tag(ln)
        msgs, secrets, nonces = _inputs(oracle_mod, 1, ln, 900 + ln)
        got = [oracle_mod.int_from_mont(v) for v in oracle_mod.encrypt_batch(tag, msgs, secrets, nonces)[0]]
        m = [oracle_mod.int_from_mont(v) for v in msgs[0]]
        st = [oracle_mod.int_from_mont(tag), oracle_mod.int_from_mont(secrets[0, 0]), oracle_mod.int_from_mont(secrets[0, 1]),
              oracle_mod.int_from_mont(nonces[0]), 0]
        exp = []
        for off in range(0, ln, 4):
            st = perm(st)
            for k in range(min(4, ln - off)):
                st[1 + k] = (st[1 + k] + m[off + k]) % P
                exp.append(st[1 + k])
        st = perm(st)
        exp.append(st[1])
        assert got == exp


def test_encryption_tag_helpers_agree(oracle_mod):
    from poseidon252_amd.encryption import encryption_tag
    for ln in (1, 2, 4, 5, 9, 100):
        assert np.array_equal(encryption_tag(ln), oracle_mod.encryption_tag(ln))
    import poseidon252_amd as P
    with pytest.raises(P.InvalidIOPattern):
        encryption_tag(0)


# ------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("ln", [1, 2, 3, 4, 5, 7, 8, 13])
def test_gpu_encrypt_decrypt_match_oracle(gpu_ctx, oracle_mod, ln):
    from poseidon252_amd.encryption import decrypt_batch, encrypt_batch, encryption_tag
    n = 333
    tag = encryption_tag(ln)
    msgs, secrets, nonces = _inputs(oracle_mod, n, ln, 3000 + ln)
    cph = encrypt_batch(msgs, secrets, nonces, ctx=gpu_ctx)
    assert np.array_equal(cph, oracle_mod.encrypt_batch(tag, msgs, secrets, nonces))
    dec, ok = decrypt_batch(cph, secrets, nonces, ctx=gpu_ctx)
    assert ok.all() and np.array_equal(dec, msgs)
    # negative cases of tests/encryption.rs, batched: only the tampered items fail
    bad = cph.copy()
    bad[::3, ln // 2, 1] ^= np.uint64(4)
    dec2, ok2 = decrypt_batch(bad, secrets, nonces, ctx=gpu_ctx)
    assert not ok2[::3].any() and ok2[1::3].all() and ok2[2::3].all()
    o_dec, o_ok = oracle_mod.decrypt_batch(tag, bad, secrets, nonces)
    assert np.array_equal(ok2, o_ok) and np.array_equal(dec2[ok2], o_dec[o_ok])
    assert not decrypt_batch(cph, np.roll(secrets, 1, axis=0), nonces, ctx=gpu_ctx)[1].any()
    assert not decrypt_batch(cph, secrets, np.roll(nonces, 1, axis=0), ctx=gpu_ctx)[1].any()
    bad_mac = cph.copy()
    bad_mac[:, ln, 3] ^= np.uint64(1)
    assert not decrypt_batch(bad_mac, secrets, nonces, ctx=gpu_ctx)[1].any()


@pytest.mark.gpu
def test_gpu_single_message_api_like_reference(gpu_ctx, oracle_mod):
    """src/encryption.rs doctest shape: encrypt(&message, &shared_secret, &nonce) -> cipher; decrypt -> message"""
    import poseidon252_amd as P
    from poseidon252_amd.encryption import DecryptionFailed, decrypt, encrypt
    msgs, secrets, nonces = _inputs(oracle_mod, 1, 2, 42)  # benches/encrypt.rs: 2 BlsScalar
    cipher = encrypt(msgs[0], secrets[0], nonces[0], ctx=gpu_ctx)
    assert cipher.shape == (3, 4)
    assert np.array_equal(decrypt(cipher, secrets[0], nonces[0], ctx=gpu_ctx), msgs[0])
    with pytest.raises(DecryptionFailed):
        decrypt(cipher, secrets[0][::-1].copy(), nonces[0], ctx=gpu_ctx)
    with pytest.raises(P.InvalidIOPattern):
        encrypt(np.zeros((0, 4), dtype=np.uint64), secrets[0], nonces[0], ctx=gpu_ctx)


@pytest.mark.gpu
def test_gpu_encrypt_device_buffers(gpu_ctx, oracle_mod):
    import ctypes
    import torch
    from poseidon252_amd import _lib
    from poseidon252_amd.encryption import encryption_tag
    n, ln = 1000, 6
    tag = encryption_tag(ln)
    msgs, secrets, nonces = _inputs(oracle_mod, n, ln, 777)
    d_m = torch.from_numpy(msgs.view(np.int64)).cuda()
    d_s = torch.from_numpy(secrets.view(np.int64)).cuda()
    d_n = torch.from_numpy(nonces.view(np.int64)).cuda()
    d_c = torch.empty((n, ln + 1, 4), dtype=torch.int64, device="cuda")
    d_back = torch.empty((n, ln, 4), dtype=torch.int64, device="cuda")
    d_ok = torch.zeros(n, dtype=torch.uint8, device="cuda")
    u64p = ctypes.POINTER(ctypes.c_uint64)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    L = _lib.lib()
    gpu_ctx._check(L.p252_encrypt_batch_device(gpu_ctx._h, tag.ctypes.data_as(u64p), d_m.data_ptr(), d_s.data_ptr(), d_n.data_ptr(), ln, d_c.data_ptr(), n, st))
    gpu_ctx._check(L.p252_decrypt_batch_device(gpu_ctx._h, tag.ctypes.data_as(u64p), d_c.data_ptr(), d_s.data_ptr(), d_n.data_ptr(), ln, d_back.data_ptr(), d_ok.data_ptr(), n, st))
    torch.cuda.synchronize()
    assert np.array_equal(d_c.cpu().numpy().view(np.uint64), oracle_mod.encrypt_batch(tag, msgs, secrets, nonces))
    assert bool(d_ok.all()) and torch.equal(d_back, d_m)
