"""Synthetic inputs: SURVEY.md §8(d)'s generator, vectorised with torch so that it runs ON the device.

splitmix64 stream from the seed -> 4 limbs per candidate, top bit of limb 3 cleared, candidates >= p rejected;
the k-th accepted candidate is scalar k.  The 4 limbs are used directly as BlsScalar memory (every residue in
[0, p) is a valid Montgomery representative), so the result is a uniform field element and the SAME bytes the
CPU oracle's `fill_random(seed, n)` produces (tests/test_synth.py pins that) — bench.py times exactly the batch
the full-size parity test verifies.
"""
import torch

_GAMMA = 0x9E3779B97F4A7C15
_M1 = 0xBF58476D1CE4E5B9
_M2 = 0x94D049BB133111EB
# p = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001, little-endian u64 limbs (src/hades.rs:12)
_P_LIMBS = (0xFFFFFFFF00000001, 0x53BDA402FFFE5BFE, 0x3339D80809A1D805, 0x73EDA753299D7D48)


def _i64(v):
    """two's-complement int64 view of an unsigned 64-bit constant"""
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >= (1 << 63) else v


def _lsr(z, k):
    """logical shift right of int64 lanes"""
    return (z >> k) & ((1 << (64 - k)) - 1)


def _splitmix_outputs(seed, first, count, device):
    """outputs number first+1 .. first+count of the splitmix64 stream started at `seed` (int64 tensor)"""
    idx = torch.arange(first + 1, first + count + 1, dtype=torch.int64, device=device)
    z = idx * _i64(_GAMMA) + _i64(seed)  # wraps mod 2^64
    z = (z ^ _lsr(z, 30)) * _i64(_M1)
    z = (z ^ _lsr(z, 27)) * _i64(_M2)
    return z ^ _lsr(z, 31)


def _lt_p(c):
    """c: (m, 4) int64 limbs (unsigned semantics) -> bool mask of candidates < p"""
    flip = _i64(1 << 63)
    lt = torch.zeros(c.shape[0], dtype=torch.bool, device=c.device)
    eq = torch.ones(c.shape[0], dtype=torch.bool, device=c.device)
    for k in (3, 2, 1, 0):
        a = c[:, k] ^ flip  # unsigned order == signed order after flipping the top bit
        b = _i64(_P_LIMBS[k]) ^ flip
        lt |= eq & (a < b)
        eq &= a == b
    return lt


def splitmix_scalars(seed, n, device="cpu", chunk=1 << 22):
    """(n, 4) int64 tensor on `device`: the first n scalars of the §8(d) generator for `seed`"""
    out = torch.empty((n, 4), dtype=torch.int64, device=device)
    have, cand = 0, 0
    while have < n:
        m = min(chunk, max(1024, int((n - have) * 1.12) + 64))  # acceptance rate p / 2^255 = 0.9056
        c = _splitmix_outputs(seed, 4 * cand, 4 * m, device).view(m, 4).clone()
        c[:, 3] &= (1 << 63) - 1
        ok = c[_lt_p(c)]
        take = min(ok.shape[0], n - have)
        out[have:have + take] = ok[:take]
        cand += m  # every candidate of the chunk is consumed before the next chunk starts (or we are done)
        have += take
    return out
