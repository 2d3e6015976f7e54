// merkle2.hip — branch re-hash of arity-2 Merkle openings ("batched path recomputation ... and Domain::Merkle2").
// node = Hash::digest(Domain::Merkle2, [left, right]) = perm([tag, left, right, 0, 0])[1] (hash.rs:27-31 with io-pattern
// [Absorb(2), Squeeze(1)]); a lane walks its depth levels sequentially, the sibling of each level on the side its position bit says.
// The permutation is the library's (hades29.hpp, the digest specialisation of k_merkle4: tag S-box hoisted to the host, only the
// squeezed row of the last layer); this file only adds the arity-2 walk.  Its own translation unit: kernels.hip — whose source digest
// the committed counter passes and ISA counts are keyed to — is untouched.
#include <hip/hip_runtime.h>

#include "hades29.hpp"
#include "kernels.h"
#include "openings.h"

namespace p252 {

namespace {

struct alignas(16) Rec {
    uint32_t w[8];
};

__device__ __forceinline__ E29 load_rec(const Rec* __restrict__ p) {
    const uint4 lo = *reinterpret_cast<const uint4*>(p);
    const uint4 hi = *(reinterpret_cast<const uint4*>(p) + 1);
    const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return from_mont4(w);
}

}  // namespace

__global__ void __launch_bounds__(P252_BLOCK) __attribute__((amdgpu_waves_per_eu(3, 3)))
k_merkle2_path(const int32_t* __restrict__ tab, TagArg tag, const Rec* __restrict__ leaves, const Rec* __restrict__ siblings,
               const uint8_t* __restrict__ positions, unsigned depth, Rec* __restrict__ roots, size_t n) {
    const size_t idx = (size_t)blockIdx.x * P252_BLOCK + threadIdx.x;
    if (idx >= n) return;
    E29 cur = load_rec(leaves + idx);
    const Rec* sib = siblings + idx * depth;
    const uint8_t* pos = positions + idx * depth;
#pragma unroll 1
    for (unsigned l = 0; l < depth; ++l) {
        const bool right = (pos[l] & 1u) != 0;  // the path's node is the RIGHT child
        const E29 other = load_rec(sib + l);
        E29 s[WIDTH];
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            s[0].d[k] = tag.x0[k];  // lane 0 enters after its first S-box (hades_permute PRE0)
            s[1].d[k] = right ? other.d[k] : cur.d[k];
            s[2].d[k] = right ? cur.d[k] : other.d[k];
        }
        s[3] = e29_zero();
        s[4] = e29_zero();
        hades_permute<0x02u, true>(s, tab);
        cur = s[1];
    }
    uint32_t w[8];
    to_mont4(cur, w);
    Rec* out = roots + idx;
    *reinterpret_cast<uint4*>(out) = make_uint4(w[0], w[1], w[2], w[3]);
    *(reinterpret_cast<uint4*>(out) + 1) = make_uint4(w[4], w[5], w[6], w[7]);
}

hipError_t launch_merkle2_path(const int32_t* tab, const TagArg& tag, const void* leaves, const void* siblings, const void* positions, unsigned depth,
                               void* roots, size_t n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_merkle2_path, dim3((unsigned)((n + P252_BLOCK - 1) / P252_BLOCK)), dim3(P252_BLOCK), 0, st, tab, tag,
                       static_cast<const Rec*>(leaves), static_cast<const Rec*>(siblings), static_cast<const uint8_t*>(positions), depth,
                       static_cast<Rec*>(roots), n);
    return hipGetLastError();
}

// ---- Opening::verify in bulk (the downstream poseidon-merkle verifier, AGENTS.md:62-66): the re-hashed roots of n openings against
// the ONE expected root.  ok[i] = 1 iff equal — n bytes leave the device instead of n x 32.  Data movement only (32 B in, 1 B out per
// opening), behind either arity's re-hash kernel. ----
__global__ void __launch_bounds__(256) k_compare_roots(const uint4* __restrict__ roots, const uint4* __restrict__ expected, uint8_t* __restrict__ ok, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint4 e0 = expected[0], e1 = expected[1], a = roots[2 * i], b = roots[2 * i + 1];
    const bool same = a.x == e0.x && a.y == e0.y && a.z == e0.z && a.w == e0.w && b.x == e1.x && b.y == e1.y && b.z == e1.z && b.w == e1.w;
    ok[i] = same ? (uint8_t)1 : (uint8_t)0;
}

hipError_t launch_compare_roots(const void* roots, const void* expected, void* ok, size_t n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_compare_roots, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, static_cast<const uint4*>(roots),
                       static_cast<const uint4*>(expected), static_cast<uint8_t*>(ok), n);
    return hipGetLastError();
}

// ---- a failed peer in a sharded build (comm.cpp): a rank whose local subtree build fails still enters the all-gather, with
// an all-ones "root" (no BlsScalar: >= p), so that its peers are not left blocked.  This kernel, behind the top levels on every healthy
// rank, looks for that sentinel among the gathered roots: found -> the final root is overwritten with all-ones as well (nothing
// downstream can take it for a digest) and 1 + the failing rank goes to *fail_flag, a word of host-mapped memory the host reads at its
// next synchronisation point (p252_comm_check, p252_sync).  One wave; world x 32 bytes read. ----
__global__ void __launch_bounds__(64) k_poison_if_peer_failed(const uint4* __restrict__ roots, unsigned world, uint4* __restrict__ root_out,
                                                              unsigned* __restrict__ fail_flag) {
    unsigned first = 0xffffffffu;
    for (unsigned r = threadIdx.x; r < world; r += 64) {
        const uint4 a = roots[2 * r], b = roots[2 * r + 1];
        if ((a.x & a.y & a.z & a.w & b.x & b.y & b.z & b.w) == 0xffffffffu && r < first) first = r;
    }
    for (int off = 32; off; off >>= 1) {
        const unsigned o = __shfl_xor(first, off);
        first = o < first ? o : first;
    }
    if (threadIdx.x == 0 && first != 0xffffffffu) {
        const uint4 ones = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
        root_out[0] = ones;
        root_out[1] = ones;
        __hip_atomic_store(fail_flag, first + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

hipError_t launch_poison_if_peer_failed(const void* roots, unsigned world, void* root_out, unsigned* fail_flag, hipStream_t st) {
    hipLaunchKernelGGL(k_poison_if_peer_failed, dim3(1), dim3(64), 0, st, static_cast<const uint4*>(roots), world, static_cast<uint4*>(root_out), fail_flag);
    return hipGetLastError();
}

}  // namespace p252
