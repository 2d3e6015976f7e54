// fastdiv.hpp — division of a 32-bit lane index by a small wave-uniform divisor (the tree depth, <= 64) as the high half of a
// product with a host-computed 64-bit reciprocal.  Used by openings.hip's FAST extraction kernel; compiled for the host too
// (hosttest.cpp) so that the boundary indices are checked by a CPU test (tests/test_host_arith.py) — round 5's 40-bit
// multiply-shift wrapped from opening 2^24 on and no test saw it.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define P252_FASTDIV_HD __host__ __device__ __forceinline__
#else
#define P252_FASTDIV_HD inline __attribute__((always_inline))
#endif

namespace p252 {

// M = floor((2^64 - 1) / d) + 1 = (2^64 + e) / d with 0 <= e < d, for 2 <= d.  d == 1 has no 64-bit reciprocal (M = 2^64): 0 stands
// for "the quotient is the dividend".
P252_FASTDIV_HD unsigned long long fast_div_reciprocal(unsigned d) { return d >= 2 ? ~0ull / d + 1 : 0ull; }

// rec / d for every rec < 2^32 and 1 <= d <= 64 (exact far beyond 64; that is what the callers need and the test sweeps):
// rec * M / 2^64 = rec / d + rec * e / (d * 2^64), and rec * e < 2^38 < 2^64 / d leaves the floor alone.  The high 64 bits of the
// 96-bit product rec * M are (rec * M_hi + mulhi32(rec, M_lo)) >> 32: one v_mul_hi_u32 and one v_mad_u64_u32; the sum cannot carry
// out of 64 bits (M_hi <= 2^31).
P252_FASTDIV_HD unsigned fast_div(unsigned rec, unsigned long long m) {
    const unsigned hi_lo = (unsigned)(((unsigned long long)rec * (unsigned)m) >> 32);
    const unsigned long long t = (unsigned long long)rec * (unsigned)(m >> 32) + hi_lo;
    return m ? (unsigned)(t >> 32) : rec;
}

}  // namespace p252
