"""Host-side mirror of `dusk_poseidon::{encrypt, decrypt}` (src/encryption.rs:62-95), single and batched.

The reference wraps `dusk_safe::encrypt/decrypt` with `ScalarPermutation`, `Domain::Encryption` and the
two coordinates of a JubJub shared point as the secret.  dusk-safe is not vendored in the reference and
its tests only check round trips and failures (tests/encryption.rs:30-115), so the construction is UNPINNED at
the byte level.  The library interprets the literal sponge-call sequence; `variant` picks it:
  STREAM (default)  [Absorb(2), Absorb(1), Squeeze(len), Absorb(len), Squeeze(1)] — dusk-safe's encrypt as recollected
  DUPLEX            [Absorb(2), Absorb(1), {Squeeze(c), Absorb(c)}*, Squeeze(1)], c = min(4, remaining)
(identical for len <= 4).  cipher[i] = message[i] + mask[i], cipher[len] = MAC.  All hashing happens on the GPU.

shared_secret: (2,4) uint64 — the (u, v) coordinates of the shared JubJubAffine as BlsScalars
               (encryption.rs:66-69: `shared_secret.get_u(), shared_secret.get_v()`); nonce: (4,) uint64.
"""
import ctypes

import numpy as np

from . import _lib
from .hash import Context, Error, _as_scalars, _raise

_u64p = ctypes.POINTER(ctypes.c_uint64)


STREAM, DUPLEX = 0, 1  # P252_CRYPT_STREAM / P252_CRYPT_DUPLEX (include/poseidon252_hip.h)


class DecryptionFailed(Error):
    """Error::DecryptionFailed (src/error.rs:27-29)"""


def encryption_tag(message_len, variant=STREAM):
    """Safe::tag of the encryption io-pattern for a message of `message_len` scalars.  UNPINNED recipe."""
    out = np.empty(4, dtype=np.uint64)
    rc = _lib.lib().p252_encryption_tag(int(variant), int(message_len), out.ctypes.data_as(_u64p))
    if rc:
        _raise(rc)
    return out


def encrypt_batch(messages, shared_secrets, nonces, ctx=None, tag=None, variant=STREAM):
    """messages (n,len,4), shared_secrets (n,2,4), nonces (n,4) -> ciphers (n,len+1,4)"""
    secrets = _as_scalars(shared_secrets).reshape(-1, 2, 4)
    n = secrets.shape[0]
    msgs = _as_scalars(messages).reshape(n, -1, 4) if n else _as_scalars(messages).reshape(0, 1, 4)
    non = _as_scalars(nonces).reshape(n, 4)
    ln = msgs.shape[1]
    ctx = ctx or Context.default()
    tag = encryption_tag(ln, variant) if tag is None else _as_scalars(tag).reshape(4)
    out = np.empty((n, ln + 1, 4), dtype=np.uint64)
    ctx._check(_lib.lib().p252_encrypt_batch(ctx._h, int(variant), tag.ctypes.data_as(_u64p), msgs.ctypes.data_as(_u64p),
                                             secrets.ctypes.data_as(_u64p), non.ctypes.data_as(_u64p), ln,
                                             out.ctypes.data_as(_u64p), n))
    return out


def decrypt_batch(ciphers, shared_secrets, nonces, ctx=None, tag=None, variant=STREAM):
    """ciphers (n,len+1,4) -> (messages (n,len,4), ok (n,) bool); ok[i] False = DecryptionFailed for item i"""
    secrets = _as_scalars(shared_secrets).reshape(-1, 2, 4)
    n = secrets.shape[0]
    cph = _as_scalars(ciphers).reshape(n, -1, 4)
    non = _as_scalars(nonces).reshape(n, 4)
    ln = cph.shape[1] - 1
    ctx = ctx or Context.default()
    if ln < 1:
        ctx._check(_lib.ERR_INVALID_IO_PATTERN)
    tag = encryption_tag(ln, variant) if tag is None else _as_scalars(tag).reshape(4)
    out = np.empty((n, ln, 4), dtype=np.uint64)
    ok = np.zeros(n, dtype=np.uint8)
    ctx._check(_lib.lib().p252_decrypt_batch(ctx._h, int(variant), tag.ctypes.data_as(_u64p), cph.ctypes.data_as(_u64p),
                                             secrets.ctypes.data_as(_u64p), non.ctypes.data_as(_u64p), ln,
                                             out.ctypes.data_as(_u64p), ok.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), n))
    return out, ok.astype(bool)


def encrypt(message, shared_secret, nonce, ctx=None, tag=None, variant=STREAM):
    """`dusk_poseidon::encrypt` (encryption.rs:62-76): Vec<BlsScalar> of message.len() + 1"""
    m = _as_scalars(message).reshape(-1, 4)
    return encrypt_batch(m[None], _as_scalars(shared_secret).reshape(1, 2, 4), _as_scalars(nonce).reshape(1, 4), ctx=ctx, tag=tag, variant=variant)[0]


def decrypt(cipher, shared_secret, nonce, ctx=None, tag=None, variant=STREAM):
    """`dusk_poseidon::decrypt` (encryption.rs:81-95): the message, or raises DecryptionFailed"""
    c = _as_scalars(cipher).reshape(-1, 4)
    msg, ok = decrypt_batch(c[None], _as_scalars(shared_secret).reshape(1, 2, 4), _as_scalars(nonce).reshape(1, 4), ctx=ctx, tag=tag, variant=variant)
    if not ok[0]:
        raise DecryptionFailed("DecryptionFailed")
    return msg[0]


def encrypt_batch_device(d_messages, d_secrets, d_nonces, message_len, d_ciphers, n, ctx=None, tag=None, variant=STREAM):
    """device-resident variant (torch CUDA tensors of int64 limbs): messages n*len, secrets n*2, nonces n scalars in,
    ciphers n*(len+1) scalars out; asynchronous on torch's current stream"""
    import torch
    ctx = ctx or Context.default()
    tag = encryption_tag(message_len, variant) if tag is None else _as_scalars(tag).reshape(4)
    assert all(t.is_cuda for t in (d_messages, d_secrets, d_nonces, d_ciphers))
    assert d_ciphers.numel() * d_ciphers.element_size() >= n * (message_len + 1) * 32
    ctx._check(_lib.lib().p252_encrypt_batch_device(ctx._h, int(variant), tag.ctypes.data_as(_u64p), d_messages.data_ptr(), d_secrets.data_ptr(),
                                                    d_nonces.data_ptr(), message_len, d_ciphers.data_ptr(), n,
                                                    torch.cuda.current_stream().cuda_stream))


def decrypt_batch_device(d_ciphers, d_secrets, d_nonces, message_len, d_messages, d_ok, n, ctx=None, tag=None, variant=STREAM):
    """device-resident variant: d_ok is a uint8 tensor of n flags (0 = DecryptionFailed for that item)"""
    import torch
    ctx = ctx or Context.default()
    tag = encryption_tag(message_len, variant) if tag is None else _as_scalars(tag).reshape(4)
    assert all(t.is_cuda for t in (d_ciphers, d_secrets, d_nonces, d_messages, d_ok)) and d_ok.numel() >= n
    ctx._check(_lib.lib().p252_decrypt_batch_device(ctx._h, int(variant), tag.ctypes.data_as(_u64p), d_ciphers.data_ptr(), d_secrets.data_ptr(),
                                                    d_nonces.data_ptr(), message_len, d_messages.data_ptr(), d_ok.data_ptr(), n,
                                                    torch.cuda.current_stream().cuda_stream))
