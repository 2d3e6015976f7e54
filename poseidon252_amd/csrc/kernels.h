// kernels.h — launch interface between api.cpp (host, C ABI) and kernels.hip (device).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#ifndef P252_BLOCK
#define P252_BLOCK 256
#endif

namespace p252 {

// the sponge tag (state[0]) travels as a kernel argument: 8 x u32 = one BlsScalar, plus (for the single-permutation
// digest kernels) the nine digits of its first-round S-box output, computed once per launch on the host (hades_pre0)
struct TagArg {
    uint32_t w[8];
    int32_t x0[9];
};

hipError_t launch_permute(const int32_t* tab, const void* in, void* out, size_t n, hipStream_t st);
// pad_lanes > n: the n nodes are computed redundantly by pad_lanes lanes (k_merkle4_pad: narrow levels of a large tree)
// trunc250: the output stage stores finalize_truncated's raw limbs (canonical value & (2^250 - 1), hash.rs:164-183) instead of BlsScalars
hipError_t launch_merkle4(const int32_t* tab, const TagArg& tag, const void* children, size_t n_children,
                          void* out, size_t n, hipStream_t st, unsigned arity = 4, size_t pad_lanes = 0, bool trunc250 = false);
// incremental tree update: index[k] = leaf positions (u32); one level: node = index[i] >> shift, re-hashed from `children`
// (index[i] >= n_dst: skipped, *n_bad incremented when n_bad != nullptr)
hipError_t launch_scatter_scalars(const void* index, const void* values, void* dst, size_t k, size_t n_dst, void* n_bad, hipStream_t st);
hipError_t launch_merkle4_update(const int32_t* tab, const TagArg& tag, const void* index, unsigned shift, const void* children,
                                 size_t n_children, void* out, size_t k, hipStream_t st);
hipError_t launch_sponge(const int32_t* tab, const TagArg& tag, const void* in, unsigned in_len,
                         unsigned out_len, void* out, size_t n, hipStream_t st, bool trunc250 = false);

// prog: device array of n_calls sponge-call words (kind << 29 | len), see k_crypt
hipError_t launch_crypt(bool decrypt, const int32_t* tab, const TagArg& tag, const void* in, const void* secrets,
                        const void* nonces, unsigned len, void* out, void* flags, size_t n, const uint32_t* prog,
                        unsigned n_calls, hipStream_t st);
hipError_t launch_truncate250(const void* in, void* out, size_t n, hipStream_t st);
// canonical byte format (BlsScalar::to_bytes / from_bytes): r2 = the balanced digits of 2^517 mod p; ok may be null
struct Digits9 {
    int32_t d[9];
};
hipError_t launch_to_canonical(const void* in, void* out, size_t n, hipStream_t st);
hipError_t launch_from_canonical(const void* in, void* out, void* ok, size_t n, const Digits9& r2, hipStream_t st);
hipError_t launch_merkle4_path(const int32_t* tab, const TagArg& tag, const void* leaves, const void* siblings,
                               const void* positions, unsigned depth, void* roots, size_t n, hipStream_t st);

// measurement aid (bench.py): one wave samples the shader-clock and real-time counters around a sleep of spin_ticks (100 MHz ticks)
hipError_t launch_clock_probe(void* out6, unsigned spin_ticks, hipStream_t st);

}  // namespace p252
