// openings.hip — openings of a stored arity-4 Merkle tree, extracted ON THE DEVICE (the downstream poseidon-merkle
// `tree.branch(pos)`, AGENTS.md:62-66).  No hashing: for opening i and level l the node on the path is index_i >> 2l, its position
// among its parent's children the low two bits of that, and its three siblings the other nodes of the same group of four — one
// 128-byte cache line of the level array.  HBM-bound byte movement, laid out for it:
//   * SIX lanes per (opening, level), one per 16-byte piece of the 96-byte sibling record: lane t of the launch stores the t-th
//     16-byte word of the siblings array — every store instruction of a wave writes 1 KB contiguously — and loads the matching
//     piece of the group's line (three of its four 32-byte records; the six lanes of a record read one line between them);
//   * the reads are gathers by nature (k random paths): one 128-byte line per record at the leaf levels, cache hits higher up;
//   * piece 0 of a record also stores the position byte; pieces 0 and 1 of level 0 copy the two halves of the leaf.
// (A first version with one lane per record — six loads and six stores at a 96-byte lane stride — moved 3.19 TB/s algorithmic on
// random positions, 0.70 of a device-to-device copy counting fetched lines: profiles/r04_openings_extract.txt has the steps.)
// Algorithmic bytes per (opening, level): 96 read + 96 written + 1 position byte (+ 64 per opening for the leaf).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "fastdiv.hpp"
#include "openings.h"

namespace p252 {

namespace {

// a value select, not a pointer select: `cond ? array[i] : zero` makes the compiler park `zero` in SCRATCH and load through a selected
// pointer — 16 bytes of scratch written per lane, 1.2 GB per 2^20 openings of depth 12, which the first counter pass showed as
// HBM write traffic 1.83 x the algorithmic bytes (profiles/r04_pmc_k_merkle4_openings.txt)
__device__ __forceinline__ uint4 load_or_zero(const uint4* __restrict__ base, size_t word, bool ok) {
    uint4 v = base[ok ? word : 0];  // (word 0 always exists: the tree has at least one leaf)
    v.x = ok ? v.x : 0u;
    v.y = ok ? v.y : 0u;
    v.z = ok ? v.z : 0u;
    v.w = ok ? v.w : 0u;
    return v;
}

// IDX = uint32_t whenever the launch has fewer than 2^32 lanes (6 k depth < 2^32: every realistic batch) — the lane's record and
// level then come from 32-bit divisions; size_t otherwise
// ARITY 4: three siblings per level (six 16-byte pieces per record); ARITY 2 (Domain::Merkle2 trees): one sibling (two pieces)
// FAST (round 5; 32-bit lane indices only): what the counters of round 4's kernel pointed at was not memory (0.41-0.43 of HBM peak by
// FETCH_SIZE / WRITE_SIZE) but per-lane integer work around ONE 16-byte load and ONE 16-byte store — a 32-bit division by `depth`
// (~30 instructions with a quarter-rate reciprocal) and a DIVERGENT loop of up to `depth` trips for the level's offset (lanes of a
// wave sit on ~11 different levels).  FAST: the division is the high half of a product with a host-computed 64-bit reciprocal
// (fastdiv.hpp: exact for every record index < 2^32 and depth <= 64; round 5's 40-bit multiply-shift wrapped from opening
// 2^24 on — ADVICE r5), and the `depth` (offset, count) pairs are computed ONCE per block by its first `depth` lanes into LDS
// (512 B) and read back with one ds_read per lane.
struct LevelRef {
    unsigned long long first;  // index of the level's first 16-byte word, relative to `levels` (level 0: the leaves, unused)
    unsigned long long cnt;    // nodes in the level
};
template <class IDX, unsigned ARITY, bool FAST = false>
__global__ void __launch_bounds__(256) k_merkle4_openings(const uint4* __restrict__ leaves, size_t n_leaves, const uint4* __restrict__ levels,
                                                          const uint32_t* __restrict__ index, size_t k, unsigned depth,
                                                          uint4* __restrict__ leaves_out, uint4* __restrict__ siblings,
                                                          uint8_t* __restrict__ positions, unsigned* __restrict__ n_bad,
                                                          unsigned long long inv_depth) {
    __shared__ LevelRef lvl[FAST ? 64 : 1];
    if (FAST) {  // (before any lane leaves: every lane of the block reaches the barrier)
        if (threadIdx.x < depth && threadIdx.x < 64) {
            unsigned long long first = 0, cnt = n_leaves;
            for (unsigned j = 0; j < threadIdx.x; ++j) {
                first = j == 0 ? 0 : first + 2 * cnt;
                cnt = (cnt + ARITY - 1) >> (ARITY == 4 ? 2 : 1);
            }
            lvl[threadIdx.x].first = first;
            lvl[threadIdx.x].cnt = cnt;
        }
        __syncthreads();
    }
    const IDX t = (IDX)blockIdx.x * 256 + threadIdx.x;  // = the index of the 16-byte word of `siblings` this lane stores
    if (depth == 0) {  // a single-leaf tree has no levels: two lanes per opening copy the leaf
        if (t >= 2 * k) return;
        const size_t i = t >> 1, leaf = index[i];
        const bool bad = leaf >= n_leaves;
        leaves_out[t] = load_or_zero(leaves, 2 * leaf + (t & 1), !bad);
        if (bad && n_bad && !(t & 1)) atomicAdd(n_bad, 1u);
        return;
    }
    constexpr unsigned PIECES = 2 * (ARITY - 1), SHIFT = ARITY == 4 ? 2 : 1;
    if (t >= (IDX)(k * depth * PIECES)) return;
    const IDX rec = t / PIECES;
    const unsigned piece = (unsigned)(t - rec * PIECES), sib = piece >> 1, half = piece & 1u;
    const IDX i = FAST ? (IDX)fast_div((unsigned)rec, inv_depth) : rec / (IDX)depth;
    const unsigned l = (unsigned)(rec - i * depth);
    const size_t leaf = index[i];
    const bool bad = leaf >= n_leaves;
    if (l == 0 && piece < 2) {
        leaves_out[2 * (size_t)i + piece] = load_or_zero(leaves, 2 * leaf + piece, !bad);
        if (bad && n_bad && piece == 0) atomicAdd(n_bad, 1u);
    }
    // the array of level l (level 0 = the leaves) and its length: n_0 = n_leaves, n_{l+1} = ceil(n_l / 4)
    const uint4* nodes = leaves;
    size_t cnt = n_leaves;
    if (FAST) {
        const LevelRef r = lvl[l];
        nodes = l == 0 ? leaves : levels + r.first;
        cnt = r.cnt;
    } else {
        for (unsigned j = 0; j < l; ++j) {  // (at most `depth` trips of two integer operations, divergent within a wave)
            nodes = j == 0 ? levels : nodes + 2 * cnt;
            cnt = (cnt + ARITY - 1) >> SHIFT;
        }
    }
    const size_t node = leaf >> (SHIFT * l);
    const unsigned p = (unsigned)(node & (ARITY - 1));
    const size_t j = node - p + sib + (sib >= p ? 1u : 0u);  // the group's nodes in order, the path's own node left out
    siblings[t] = load_or_zero(nodes, 2 * j + half, !bad && j < cnt);  // a ragged level's missing siblings are the zero scalar (hash.rs:22-26)
    if (piece == 0) positions[rec] = bad ? (uint8_t)0 : (uint8_t)p;
}

}  // namespace

template <unsigned ARITY>
static hipError_t launch_openings(const void* leaves, size_t n_leaves, const void* levels, const void* index, size_t k, unsigned depth,
                                  void* leaves_out, void* siblings, void* positions, void* n_bad, hipStream_t st) {
    if (k == 0) return hipSuccess;
    const size_t lanes = depth ? k * depth * 2 * (ARITY - 1) : 2 * k;
    const dim3 grid((unsigned)((lanes + 255) / 256));
    // P252_OPENINGS_FAST=0: round 4's kernel (division + per-lane level loop) — kept for the A/B of profiles/r05_openings_extract.txt
    static const bool fast = [] {
        const char* e = std::getenv("P252_OPENINGS_FAST");
        return !(e && e[0] == '0');
    }();
    const unsigned long long inv_depth = fast_div_reciprocal(depth);
    if (lanes + 256 <= 0xffffffffull && fast && depth >= 1 && depth <= 64)
        hipLaunchKernelGGL((k_merkle4_openings<uint32_t, ARITY, true>), grid, dim3(256), 0, st, static_cast<const uint4*>(leaves), n_leaves,
                           static_cast<const uint4*>(levels), static_cast<const uint32_t*>(index), k, depth, static_cast<uint4*>(leaves_out),
                           static_cast<uint4*>(siblings), static_cast<uint8_t*>(positions), static_cast<unsigned*>(n_bad), inv_depth);
    else if (lanes + 256 <= 0xffffffffull)
        hipLaunchKernelGGL((k_merkle4_openings<uint32_t, ARITY>), grid, dim3(256), 0, st, static_cast<const uint4*>(leaves), n_leaves,
                           static_cast<const uint4*>(levels), static_cast<const uint32_t*>(index), k, depth, static_cast<uint4*>(leaves_out),
                           static_cast<uint4*>(siblings), static_cast<uint8_t*>(positions), static_cast<unsigned*>(n_bad), inv_depth);
    else
        hipLaunchKernelGGL((k_merkle4_openings<size_t, ARITY>), grid, dim3(256), 0, st, static_cast<const uint4*>(leaves), n_leaves,
                           static_cast<const uint4*>(levels), static_cast<const uint32_t*>(index), k, depth, static_cast<uint4*>(leaves_out),
                           static_cast<uint4*>(siblings), static_cast<uint8_t*>(positions), static_cast<unsigned*>(n_bad), inv_depth);
    return hipGetLastError();
}

hipError_t launch_merkle4_openings(const void* leaves, size_t n_leaves, const void* levels, const void* index, size_t k, unsigned depth,
                                   void* leaves_out, void* siblings, void* positions, void* n_bad, hipStream_t st) {
    return launch_openings<4>(leaves, n_leaves, levels, index, k, depth, leaves_out, siblings, positions, n_bad, st);
}
hipError_t launch_merkle2_openings(const void* leaves, size_t n_leaves, const void* levels, const void* index, size_t k, unsigned depth,
                                   void* leaves_out, void* siblings, void* positions, void* n_bad, hipStream_t st) {
    return launch_openings<2>(leaves, n_leaves, levels, index, k, depth, leaves_out, siblings, positions, n_bad, st);
}

}  // namespace p252
