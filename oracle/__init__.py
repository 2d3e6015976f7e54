"""ctypes binding of the CPU oracle (oracle/p252_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package;
the product package `poseidon252_amd` never does.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libp252_oracle.so")


def build(force=False):
    """Compile the C restatement (gcc).  Building the checker is not using it."""
    stale = force or not os.path.exists(_LIB_PATH)
    if not stale:
        t = os.path.getmtime(_LIB_PATH)
        stale = any(os.path.getmtime(os.path.join(_HERE, f)) > t for f in ("p252_oracle.c", "p252_oracle.h"))
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_lib = None
_u64p = ctypes.POINTER(ctypes.c_uint64)
_szp = ctypes.POINTER(ctypes.c_size_t)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        L.p252o_permute_batch.argtypes = [_u64p, _u64p, ctypes.c_size_t]
        L.p252o_hash_batch.argtypes = [_u64p, _u64p, ctypes.c_size_t, ctypes.c_size_t, _u64p, ctypes.c_size_t]
        L.p252o_hash_batch.restype = ctypes.c_int
        L.p252o_hash_batch_mt.argtypes = [_u64p, _u64p, ctypes.c_size_t, ctypes.c_size_t, _u64p, ctypes.c_size_t, ctypes.c_int]
        L.p252o_hash_batch_mt.restype = ctypes.c_int
        L.p252o_kat_hash.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
        L.p252o_merkle4_tree.argtypes = [_u64p, _u64p, ctypes.c_size_t, _u64p, _u64p]
        L.p252o_merkle4_tree.restype = ctypes.c_longlong
        L.p252o_domain_separator.argtypes = [ctypes.c_int]
        L.p252o_domain_separator.restype = ctypes.c_uint64
        L.p252o_check_io.argtypes = [ctypes.c_int, _szp, ctypes.c_size_t, ctypes.c_size_t]
        L.p252o_check_io.restype = ctypes.c_int
        L.p252o_tag.argtypes = [ctypes.c_int, _szp, ctypes.c_size_t, ctypes.c_size_t, _u64p]
        L.p252o_tag.restype = ctypes.c_int
        L.p252o_blake2b512.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
        L.p252o_fill_random.argtypes = [ctypes.c_uint64, _u64p, ctypes.c_size_t]
        for name in ("p252o_from_raw", "p252o_to_canonical", "p252o_truncate250"):
            getattr(L, name).argtypes = [_u64p, _u64p]
        for name in ("p252o_add", "p252o_mul"):
            getattr(L, name).argtypes = [_u64p, _u64p, _u64p]
        L.p252o_round_constant.argtypes = [ctypes.c_int, ctypes.c_int, _u64p]
        L.p252o_mds.argtypes = [ctypes.c_int, ctypes.c_int, _u64p]
        L.p252o_is_reduced.argtypes = [_u64p]
        L.p252o_is_reduced.restype = ctypes.c_int
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(_u64p)


def _c(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
R = (1 << 256) % P


def limbs_to_int(a):
    a = np.asarray(a, dtype=np.uint64).reshape(-1)
    return sum(int(a[i]) << (64 * i) for i in range(4))


def int_to_limbs(v):
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def mont_from_int(v):
    """integer -> Montgomery limbs (what BlsScalar::from(v) / from_raw holds in memory)"""
    return int_to_limbs((v % P) * R % P)


def int_from_mont(a):
    return limbs_to_int(a) * pow(R, -1, P) % P


def fill_random(seed, n_scalars):
    out = np.empty((n_scalars, 4), dtype=np.uint64)
    lib().p252o_fill_random(seed, _p(out), n_scalars)
    return out


def permute_batch(states):
    states = _c(states).reshape(-1, 5, 4)
    out = np.empty_like(states)
    lib().p252o_permute_batch(_p(states), _p(out), states.shape[0])
    return out


def hash_batch(tag, inp, in_len, out_len, threads=1):
    tag = _c(tag).reshape(4)
    inp = _c(inp).reshape(-1, in_len, 4)
    n = inp.shape[0]
    out = np.empty((n, out_len, 4), dtype=np.uint64)
    if threads > 1:
        rc = lib().p252o_hash_batch_mt(_p(tag), _p(inp), in_len, out_len, _p(out), n, threads)
    else:
        rc = lib().p252o_hash_batch(_p(tag), _p(inp), in_len, out_len, _p(out), n)
    if rc:
        raise ValueError("invalid io-pattern")
    return out


def levels_total(n_leaves):
    total, c = 0, n_leaves
    while c > 1:
        c = (c + 3) // 4
        total += c
    return total


def merkle4_tree(tag, leaves, want_levels=False):
    tag = _c(tag).reshape(4)
    leaves = _c(leaves).reshape(-1, 4)
    n = leaves.shape[0]
    root = np.empty(4, dtype=np.uint64)
    levels = np.empty((levels_total(n), 4), dtype=np.uint64) if want_levels else None
    perms = lib().p252o_merkle4_tree(_p(tag), _p(leaves), n, _p(root), _p(levels) if want_levels else None)
    if perms < 0:
        raise ValueError("empty tree")
    return (root, levels, perms) if want_levels else (root, perms)


def merkle2_tree(tag, leaves, want_levels=False):
    tag = _c(tag).reshape(4)
    leaves = _c(leaves).reshape(-1, 4)
    n = leaves.shape[0]
    total, c = 0, n
    while c > 1:
        c = (c + 1) // 2
        total += c
    root = np.empty(4, dtype=np.uint64)
    levels = np.empty((total, 4), dtype=np.uint64) if want_levels else None
    f = lib().p252o_merkle2_tree
    f.argtypes = [_u64p, _u64p, ctypes.c_size_t, _u64p, _u64p]
    f.restype = ctypes.c_longlong
    perms = f(_p(tag), _p(leaves), n, _p(root), _p(levels) if want_levels else None)
    if perms < 0:
        raise ValueError("empty tree")
    return (root, levels, perms) if want_levels else (root, perms)


def merkle4_path_batch(tag, leaves, siblings, positions):
    tag = _c(tag).reshape(4)
    leaves = _c(leaves).reshape(-1, 4)
    n = leaves.shape[0]
    positions = np.ascontiguousarray(positions, dtype=np.uint8).reshape(n, -1)
    depth = positions.shape[1]
    siblings = _c(siblings).reshape(n, depth, 3, 4) if depth else np.zeros((n, 0, 3, 4), dtype=np.uint64)
    roots = np.empty((n, 4), dtype=np.uint64)
    f = lib().p252o_merkle4_path_batch
    f.argtypes = [_u64p, _u64p, _u64p, ctypes.POINTER(ctypes.c_uint8), ctypes.c_size_t, _u64p, ctypes.c_size_t]
    f.restype = ctypes.c_int
    if f(_p(tag), _p(leaves), _p(siblings), positions.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), depth, _p(roots), n):
        raise ValueError("position outside 0..3")
    return roots


CRYPT_STREAM, CRYPT_DUPLEX = 0, 1  # p252_oracle.h: the two candidate dusk_safe::encrypt call sequences


def encryption_tag(message_len, variant=CRYPT_STREAM):
    """UNPINNED (see p252_oracle.h)"""
    out = np.empty(4, dtype=np.uint64)
    f = lib().p252o_encryption_tag_v
    f.argtypes = [ctypes.c_int, ctypes.c_size_t, _u64p]
    f.restype = ctypes.c_int
    if f(variant, message_len, _p(out)):
        raise ValueError("invalid message length / variant")
    return out


def encrypt_batch(tag, messages, secrets, nonces, variant=CRYPT_STREAM):
    """messages (n,len,4), secrets (n,2,4), nonces (n,4) -> ciphers (n,len+1,4)"""
    tag = _c(tag).reshape(4)
    secrets = _c(secrets).reshape(-1, 2, 4)
    n = secrets.shape[0]
    messages = _c(messages).reshape(n, -1, 4)
    nonces = _c(nonces).reshape(n, 4)
    ln = messages.shape[1]
    out = np.empty((n, ln + 1, 4), dtype=np.uint64)
    f = lib().p252o_encrypt_v
    f.argtypes = [ctypes.c_int, _u64p, _u64p, ctypes.c_size_t, _u64p, _u64p, _u64p]
    f.restype = ctypes.c_int
    for i in range(n):
        if f(variant, _p(tag), _p(messages[i]), ln, _p(secrets[i]), _p(nonces[i]), _p(out[i])):
            raise ValueError("empty message / bad variant")
    return out


def decrypt_batch(tag, ciphers, secrets, nonces, variant=CRYPT_STREAM):
    """ciphers (n,len+1,4) -> (messages (n,len,4), ok (n,) bool)"""
    tag = _c(tag).reshape(4)
    secrets = _c(secrets).reshape(-1, 2, 4)
    n = secrets.shape[0]
    ciphers = _c(ciphers).reshape(n, -1, 4)
    nonces = _c(nonces).reshape(n, 4)
    ln = ciphers.shape[1] - 1
    out = np.empty((n, ln, 4), dtype=np.uint64)
    ok = np.zeros(n, dtype=bool)
    f = lib().p252o_decrypt_v
    f.argtypes = [ctypes.c_int, _u64p, _u64p, ctypes.c_size_t, _u64p, _u64p, _u64p]
    f.restype = ctypes.c_int
    for i in range(n):
        ok[i] = f(variant, _p(tag), _p(ciphers[i]), ln, _p(secrets[i]), _p(nonces[i]), _p(out[i])) == 0
    return out, ok


def kat_hash(inputs_le32):
    """inputs: list of 32-byte little-endian canonical strings -> 32-byte LE canonical digest"""
    buf = b"".join(inputs_le32)
    out = ctypes.create_string_buffer(32)
    lib().p252o_kat_hash(buf, len(inputs_le32), out)
    return out.raw


def tag(domain, absorb_lens, out_len):
    """UNPINNED convenience (see p252_oracle.h)"""
    lens = (ctypes.c_size_t * max(1, len(absorb_lens)))(*absorb_lens)
    out = np.empty(4, dtype=np.uint64)
    rc = lib().p252o_tag(int(domain), lens, len(absorb_lens), out_len, _p(out))
    if rc:
        raise ValueError("io-pattern rejected (%d)" % rc)
    return out


def check_io(domain, absorb_lens, out_len):
    lens = (ctypes.c_size_t * max(1, len(absorb_lens)))(*absorb_lens)
    return lib().p252o_check_io(int(domain), lens, len(absorb_lens), out_len)


def blake2b512(msg):
    out = ctypes.create_string_buffer(64)
    lib().p252o_blake2b512(msg, len(msg), out)
    return out.raw


def truncate250(mont):
    out = np.empty(4, dtype=np.uint64)
    lib().p252o_truncate250(_p(_c(mont).reshape(4)), _p(out))
    return out
