// openings.hip — openings of a stored arity-4 Merkle tree, extracted ON THE DEVICE (the downstream poseidon-merkle
// `tree.branch(pos)`, AGENTS.md:62-66).  No hashing: for opening i and level l the node on the path is index_i >> 2l, its position
// among its parent's children the low two bits of that, and its three siblings the other nodes of the same group of four — one
// 128-byte cache line of the level array.  HBM-bound byte movement, laid out for it:
//   * one LANE per (opening, level): it reads its group's line once (three 32-byte records out of four) and writes one contiguous
//     96-byte record; consecutive lanes write consecutive records (siblings[i][l] is the lane's linear id), so a wave stores 6 KB
//     contiguously;
//   * 16-byte vector accesses throughout; the reads are gathers by nature (k random paths), a line per lane;
//   * lane l == 0 of an opening also copies the leaf and, every lane, its position byte.
// Algorithmic bytes per (opening, level): 96 read + 96 written + 1 position byte (+ 64 per opening for the leaf).
#include <hip/hip_runtime.h>

#include "openings.h"

namespace p252 {

namespace {

struct alignas(16) Rec32 {
    uint4 lo, hi;
};

__global__ void __launch_bounds__(256) k_merkle4_openings(const Rec32* __restrict__ leaves, size_t n_leaves, const Rec32* __restrict__ levels,
                                                          const uint32_t* __restrict__ index, size_t k, unsigned depth,
                                                          Rec32* __restrict__ leaves_out, Rec32* __restrict__ siblings,
                                                          uint8_t* __restrict__ positions, unsigned* __restrict__ n_bad) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t d1 = depth ? depth : 1;  // (a single-leaf tree has no levels: one lane per opening copies the leaf)
    if (t >= k * d1) return;
    const size_t i = t / d1;
    const unsigned l = (unsigned)(t % d1);
    const size_t leaf = index[i];
    const bool bad = leaf >= n_leaves;
    const Rec32 zero = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
    if (l == 0) {
        leaves_out[i] = bad ? zero : leaves[leaf];
        if (bad && n_bad) atomicAdd(n_bad, 1u);
    }
    if (depth == 0) return;
    // the array of level l (level 0 = the leaves) and its length: n_0 = n_leaves, n_{l+1} = ceil(n_l / 4)
    const Rec32* nodes = leaves;
    size_t cnt = n_leaves;
    for (unsigned j = 0; j < l; ++j) {  // (wave-divergent by at most `depth` trips of two integer ops: nothing beside the line fetch)
        nodes = j == 0 ? levels : nodes + cnt;
        cnt = (cnt + 3) >> 2;
    }
    const size_t node = leaf >> (2 * l);
    const unsigned p = (unsigned)(node & 3u);
    const size_t base = node - p;
    Rec32* out = siblings + t * 3;
#pragma unroll
    for (unsigned s = 0; s < 3; ++s) {
        const size_t j = base + s + (s >= p ? 1u : 0u);  // the group's nodes in order, the path's own node left out
        out[s] = (!bad && j < cnt) ? nodes[j] : zero;    // a ragged level's missing siblings are the zero scalar (hash.rs:22-26)
    }
    positions[t] = bad ? (uint8_t)0 : (uint8_t)p;
}

}  // namespace

hipError_t launch_merkle4_openings(const void* leaves, size_t n_leaves, const void* levels, const void* index, size_t k, unsigned depth,
                                   void* leaves_out, void* siblings, void* positions, void* n_bad, hipStream_t st) {
    if (k == 0) return hipSuccess;
    const size_t lanes = k * (depth ? depth : 1);
    hipLaunchKernelGGL(k_merkle4_openings, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, st, static_cast<const Rec32*>(leaves), n_leaves,
                       static_cast<const Rec32*>(levels), static_cast<const uint32_t*>(index), k, depth, static_cast<Rec32*>(leaves_out),
                       static_cast<Rec32*>(siblings), static_cast<uint8_t*>(positions), static_cast<unsigned*>(n_bad));
    return hipGetLastError();
}

}  // namespace p252
