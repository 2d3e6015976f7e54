// kernels.hip — gfx950 kernels of libposeidon252_hip.so.  One sponge state per lane, the 5 x 9
// 29-bit limbs of the state in VGPRs, 64-bit column accumulators in VGPR pairs, constants read
// through a wave-uniform pointer (scalar loads -> SGPR operands of v_mad_i64_i32).  No LDS, no
// cross-lane traffic, no MFMA: the path is VALU integer-multiply bound.
//
// Data layout in HBM: arrays of BlsScalar exactly as the reference holds them (AoS, 32 B per scalar,
// message-major).  Each lane reads its own contiguous message with 16-byte loads; at ~1e8 perm/s the
// whole stream is < 20 GB/s (0.3 % of HBM peak), so no staging/transposition is warranted.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "coop29.hpp"
#include "hades29.hpp"
#include "kernels.h"

// Occupancy target of the single-digest kernels: 3 waves per SIMD (512 / 3 = 170 VGPRs).  Left alone the
// register allocator lands one register above that (169: two waves); told the target it schedules the same
// instruction stream into 146 registers with no spills and no extra moves.
// (A/B experiments, bench_tools/ab_variants.sh: -DP252_WAVES_ATTR='__attribute__((amdgpu_waves_per_eu(2,2)))'.)
#ifndef P252_WAVES_ATTR
#define P252_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(3, 3)))
#endif

namespace p252 {

struct alignas(16) Scalar32 {
    uint32_t w[8];
};

__device__ __forceinline__ E29 load_scalar(const Scalar32* __restrict__ p) {
    const uint4 lo = *reinterpret_cast<const uint4*>(p);
    const uint4 hi = *(reinterpret_cast<const uint4*>(p) + 1);
    const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    return from_mont4(w);
}

// a record as it lies in memory (two 16-byte halves): fetched whole cache lines at a time by the sponge and opening
// kernels, kept raw across a permutation and converted where it is consumed
struct Raw32 {
    uint4 lo, hi;
};
__device__ __forceinline__ Raw32 load_raw(const Scalar32* __restrict__ p) {
    Raw32 r;
    r.lo = *reinterpret_cast<const uint4*>(p);
    r.hi = *(reinterpret_cast<const uint4*>(p) + 1);
    return r;
}
__device__ __forceinline__ Raw32 raw_zero() {
    Raw32 r;
    r.lo = make_uint4(0u, 0u, 0u, 0u);
    r.hi = r.lo;
    return r;
}
__device__ __forceinline__ Raw32 raw_select(bool first, const Raw32& a, const Raw32& b) {  // per-lane: first ? a : b
    Raw32 r;
    r.lo = make_uint4(first ? a.lo.x : b.lo.x, first ? a.lo.y : b.lo.y, first ? a.lo.z : b.lo.z, first ? a.lo.w : b.lo.w);
    r.hi = make_uint4(first ? a.hi.x : b.hi.x, first ? a.hi.y : b.hi.y, first ? a.hi.z : b.hi.z, first ? a.hi.w : b.hi.w);
    return r;
}
__device__ __forceinline__ E29 raw_to_e29(const Raw32& r) {
    const uint32_t w[8] = {r.lo.x, r.lo.y, r.lo.z, r.lo.w, r.hi.x, r.hi.y, r.hi.z, r.hi.w};
    return from_mont4(w);
}

__device__ __forceinline__ void store_scalar(Scalar32* __restrict__ p, const E29& e) {
    uint32_t w[8];
    to_mont4(e, w);
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    *(reinterpret_cast<uint4*>(p) + 1) = make_uint4(w[4], w[5], w[6], w[7]);
}

// Hash::finalize_truncated's output (hash.rs:164-183) from the kernel that already holds the value: the canonical
// value (Montgomery form dropped: redc(V * 2^5) = V * 2^5 / 2^261 = V / 2^256) & (2^250 - 1), stored as the raw limbs
// JubJubScalar::from_raw receives.  e is a tight residue (|V| < 2p < 2^256), so the reduction lands in [-p, 0]: two conditional
// subtractions canonicalise.  One product + one reduction per OUTPUT scalar (~110 of a digest's 77,000 instructions) instead of
// a second launch and a 64 B / scalar round trip through HBM (k_to_canonical<true>, kept for scalars that are already stored).
__device__ __forceinline__ void store_truncated(Scalar32* __restrict__ p, const E29& e) {
    const int32_t c32[NL] = {32, 0, 0, 0, 0, 0, 0, 0, 0};
    A29 t;
    acc_zero(t);
    acc_mul(t, e, c32);
    uint32_t w[8];
    to_mont4<2>(redc(t), w);
    w[7] &= 0x03ffffffu;  // TRUNCATION_MASK: keep the low 250 bits
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    *(reinterpret_cast<uint4*>(p) + 1) = make_uint4(w[4], w[5], w[6], w[7]);
}
template <bool TRUNC>
__device__ __forceinline__ void store_output(Scalar32* __restrict__ p, const E29& e) {
    if (TRUNC)
        store_truncated(p, e);
    else
        store_scalar(p, e);
}

// ---- n independent permutations (Safe::permute, scalar.rs:25-27) ----
// (k_permute, k_sponge and k_crypt keep all five lanes: 205-225 VGPRs = 2 waves per SIMD.  Held at 3 waves they spill
// 15-19 registers to scratch and gain 1 % on config 4 (same-box A/B, profiles/r02_ab_occupancy.txt) — at the price of
// tripling the kernel's HBM write traffic; not taken.)
__global__ void __launch_bounds__(P252_BLOCK) k_permute(const int32_t* __restrict__ tab,
                                                        const Scalar32* __restrict__ in,
                                                        Scalar32* __restrict__ out, size_t n) {
    const size_t idx = (size_t)blockIdx.x * P252_BLOCK + threadIdx.x;
    if (idx >= n) return;
    E29 s[WIDTH];
#pragma unroll
    for (int k = 0; k < WIDTH; ++k) s[k] = load_scalar(in + idx * WIDTH + k);
    hades_permute<0x1fu>(s, tab);
#pragma unroll
    for (int k = 0; k < WIDTH; ++k) store_scalar(out + idx * WIDTH + k, s[k]);
}

// ---- Merkle4 digest: Hash::digest(Domain::Merkle4, [c0..c3]) = perm([tag, c0, c1, c2, c3])[1]
// (hash.rs:128-155 with io-pattern [Absorb(4), Squeeze(1)]).  `n_children` may be short of 4*n:
// missing children are the zero scalar (hash.rs:22-26).
// Two builds of the same body: k_merkle4 (3 waves per SIMD: what a full machine wants, +1.2 % on 2^20 digests) and
// k_merkle4_lat for launches of at most one wave per SIMD — a tree's narrow levels, small batches — where occupancy
// is irrelevant and the unconstrained register allocation's schedule runs a lone wave 3.6 % faster (same-box A/B,
// profiles/r02_ab_occupancy.txt). ----
template <bool TRUNC = false>
__device__ __forceinline__ void merkle4_body(const int32_t* __restrict__ tab, const TagArg& tag,
                                             const Scalar32* __restrict__ children, size_t n_children,
                                             Scalar32* __restrict__ out, size_t n, unsigned arity) {
    // arity 4: Domain::Merkle4 node; arity 2: Domain::Merkle2 node (hash.rs:27-31) — the same sponge with two
    // absorbed elements, i.e. state [tag, c0, c1, 0, 0]; the caller passes the matching tag
    const size_t idx = (size_t)blockIdx.x * P252_BLOCK + threadIdx.x;
    if (idx >= n) return;
    E29 s[WIDTH];
#pragma unroll
    for (int k = 0; k < NL; ++k) s[0].d[k] = tag.x0[k];  // lane 0 enters after its first S-box (hades_permute PRE0)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const size_t c = idx * arity + k;
        if ((unsigned)k < arity && c < n_children)
            s[1 + k] = load_scalar(children + c);
        else
            s[1 + k] = e29_zero();
    }
    hades_permute<0x02u, true>(s, tab);  // only lane 1 is squeezed
    store_output<TRUNC>(out + idx, s[1]);
}
__global__ void __launch_bounds__(P252_BLOCK) P252_WAVES_ATTR k_merkle4(const int32_t* __restrict__ tab, TagArg tag,
                                                        const Scalar32* __restrict__ children,
                                                        size_t n_children, Scalar32* __restrict__ out,
                                                        size_t n, unsigned arity) {
    merkle4_body(tab, tag, children, n_children, out, n, arity);
}
// Hash::digest_truncated for a batch (hash.rs:203-210): the same kernel with the truncating output stage — ONE launch
__global__ void __launch_bounds__(P252_BLOCK) P252_WAVES_ATTR k_merkle4_trunc(const int32_t* __restrict__ tab, TagArg tag,
                                                        const Scalar32* __restrict__ children,
                                                        size_t n_children, Scalar32* __restrict__ out,
                                                        size_t n, unsigned arity) {
    merkle4_body<true>(tab, tag, children, n_children, out, n, arity);
}
__global__ void __launch_bounds__(P252_BLOCK) k_merkle4_lat(const int32_t* __restrict__ tab, TagArg tag,
                                                            const Scalar32* __restrict__ children,
                                                            size_t n_children, Scalar32* __restrict__ out,
                                                            size_t n, unsigned arity) {
    merkle4_body(tab, tag, children, n_children, out, n, arity);
}
// The narrow levels of a LARGE tree, computed redundantly by `lanes` > n lanes (lane i hashes node i mod n; duplicates
// store identical bytes): one wave on every SIMD of the chip instead of a handful of busy CUs.  Why: after ~1.4 ms of a
// nearly idle chip (eight latency-bound levels) the next wide launch runs at a dipped shader clock — GRBM_GUI_ACTIVE
// shows 2.14-2.25 GHz for a tree's 4M-node level against 2.36-2.40 GHz for the same kernel in a steady stream
// (profiles/r02_bench_kernel_trace_tree.txt).  Holding the load level through the narrow phase costs those levels
// nothing (they are bound by one wave's latency either way: 171 vs 174 us) and takes 3 % off a 2^24-leaf build in
// steady state (same-box A/B, profiles/r02_ab_tree_pad.txt).  Results are bit-identical; P252_TREE_PAD_LANES=0 turns it off.
__global__ void __launch_bounds__(P252_BLOCK) k_merkle4_pad(const int32_t* __restrict__ tab, TagArg tag,
                                                            const Scalar32* __restrict__ children,
                                                            size_t n_children, Scalar32* __restrict__ out,
                                                            size_t n, unsigned arity, size_t lanes) {
    const size_t lane = (size_t)blockIdx.x * P252_BLOCK + threadIdx.x;
    if (lane >= lanes) return;
    const size_t idx = (unsigned)lane % (unsigned)n;  // (lanes < 2^31: the launcher sees to it — no 64-bit division)
    E29 s[WIDTH];
#pragma unroll
    for (int k = 0; k < NL; ++k) s[0].d[k] = tag.x0[k];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const size_t c = idx * arity + k;
        if ((unsigned)k < arity && c < n_children)
            s[1 + k] = load_scalar(children + c);
        else
            s[1 + k] = e29_zero();
    }
    hades_permute<0x02u, true>(s, tab);
    store_scalar(out + idx, s[1]);
}

// ---- the low-latency digest: ONE node per group of 8 (or 4) lanes (coop29.hpp) — for launches that cannot fill the
// chip anyway (a tree's levels of <= 16,384 nodes, small batches).  209 (233) sequential field products per digest instead
// of 365: a lone wave finishes in 0.12 ms instead of 0.17.  Exchange, 8 lanes: the five S-box outputs of a full round
// travel by ds_bpermute_b32 (the LDS crossbar, no LDS memory: 45 per round, issued back to back, one wait); 4 lanes: by
// DPP quad_perm broadcast on v_mov_b32 (VALU, no wait).  The one element a partial round swaps inside a lane pair:
// DPP quad_perm [1,0,3,2].  `lanes` >= LANES * n: as in k_merkle4_pad, group g hashes node g mod n so that a large
// tree's narrow levels keep every SIMD busy. ----
struct WaveComm8 {
    int j;      // my index within the group
    int base4;  // ds_bpermute byte address of my group's lane 0 (within the wave)
    __device__ __forceinline__ int lane() const { return j; }
    template <int M>
    __device__ __forceinline__ E29 get(const E29& v) const {
        E29 r;
        const int addr = base4 + 4 * M;
#pragma unroll
        for (int k = 0; k < NL; ++k) r.d[k] = __builtin_amdgcn_ds_bpermute(addr, v.d[k]);
        return r;
    }
    __device__ __forceinline__ E29 swap1(const E29& v) const {
        E29 r;
#pragma unroll
        for (int k = 0; k < NL; ++k) r.d[k] = __builtin_amdgcn_mov_dpp(v.d[k], 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true);
        return r;
    }
};
struct WaveComm4 {  // a group = a quad
    int j;
    __device__ __forceinline__ int lane() const { return j; }
    template <int M>
    __device__ __forceinline__ E29 get(const E29& v) const {
        E29 r;
#pragma unroll
        for (int k = 0; k < NL; ++k) r.d[k] = __builtin_amdgcn_mov_dpp(v.d[k], M * 0x55 /* quad_perm [M,M,M,M] */, 0xf, 0xf, true);
        return r;
    }
    __device__ __forceinline__ E29 swap1(const E29& v) const {
        E29 r;
#pragma unroll
        for (int k = 0; k < NL; ++k) r.d[k] = __builtin_amdgcn_mov_dpp(v.d[k], 0xB1, 0xf, 0xf, true);
        return r;
    }
};
__device__ __forceinline__ E29 load_child_or_zero(const Scalar32* __restrict__ children, size_t n_children, size_t idx,
                                                  unsigned arity, int k /* child number, < 0: none */) {
    const size_t c = idx * arity + (size_t)(k > 0 ? k : 0);
    const bool present = k >= 0 && (unsigned)k < arity && c < n_children;
    return present ? load_scalar(children + c) : e29_zero();
}
template <int LANES, bool TRUNC>
__device__ __forceinline__ void merkle4_coop_body(const int32_t* __restrict__ tab, const TagArg& tag,
                                                  const Scalar32* __restrict__ children,
                                                  size_t n_children, Scalar32* __restrict__ out,
                                                  size_t n, unsigned arity, size_t lanes) {
    const size_t lane = (size_t)blockIdx.x * P252_BLOCK + threadIdx.x;
    if (lane >= lanes) return;  // (lanes is a multiple of LANES — launch_merkle4 rounds it up: whole groups only)
    const size_t idx = (unsigned)(lane / LANES) % (unsigned)n;  // (n <= 16,384 here: no 64-bit division)
    const int j = (int)(threadIdx.x & (LANES - 1));
    const int el = LANES == 8 ? (j < WIDTH ? j : WIDTH - 1) : j;  // the state element this lane brings: 0 = tag, 1..4 = children
    E29 mine = from_mont4(tag.w);
    {
        const E29 child = load_child_or_zero(children, n_children, idx, arity, el - 1);
#pragma unroll
        for (int k = 0; k < NL; ++k) mine.d[k] = el > 0 ? child.d[k] : mine.d[k];
    }
    E29 last = mine;
    if (LANES == 4) last = load_child_or_zero(children, n_children, idx, arity, 3);  // element 4 travels on every lane
    if (LANES == 8) {
        WaveComm8 cm{j, (int)(((threadIdx.x & 63u) & ~7u) * 4u)};
        CoopLane<8> L = coop_lane<8>(tab, cm);
        hades_permute_coop<8, false>(mine, last, tab, cm, L);
    } else {
        WaveComm4 cm{j};
        CoopLane<4> L = coop_lane<4>(tab, cm);
        hades_permute_coop<4, false>(mine, last, tab, cm, L);
    }
    if (j == 1) store_output<TRUNC>(out + idx, mine);  // the digest is element 1 of the permuted state: lane 1's
}
template <int LANES>
__global__ void __launch_bounds__(P252_BLOCK) k_merkle4_coop(const int32_t* __restrict__ tab, TagArg tag,
                                                             const Scalar32* __restrict__ children,
                                                             size_t n_children, Scalar32* __restrict__ out,
                                                             size_t n, unsigned arity, size_t lanes) {
    merkle4_coop_body<LANES, false>(tab, tag, children, n_children, out, n, arity, lanes);
}
__global__ void __launch_bounds__(P252_BLOCK) k_merkle4_coop8_trunc(const int32_t* __restrict__ tab, TagArg tag,
                                                                    const Scalar32* __restrict__ children,
                                                                    size_t n_children, Scalar32* __restrict__ out,
                                                                    size_t n, unsigned arity, size_t lanes) {
    merkle4_coop_body<8, true>(tab, tag, children, n_children, out, n, arity, lanes);
}

// ---- incremental update of a stored tree: k leaves changed; per level, update i re-hashes the node above
// its leaf (node = leaf_index[i] >> shift) from the node's children in the level below.  Several updates under one node
// compute it redundantly and store identical bytes.  One lane per update, or a lane group for k <= 8,192. ----
__global__ void __launch_bounds__(P252_BLOCK) k_scatter_scalars(const uint32_t* __restrict__ index, const Scalar32* __restrict__ values,
                                                                Scalar32* __restrict__ dst, size_t k, size_t n_dst,
                                                                unsigned* __restrict__ n_bad) {
    const size_t i = (size_t)blockIdx.x * P252_BLOCK + threadIdx.x;
    if (i >= k) return;
    if (index[i] >= n_dst) {  // not a leaf of this tree: skipped (and counted, when the caller asked)
        if (n_bad) atomicAdd(n_bad, 1u);
        return;
    }
    const uint4 lo = *reinterpret_cast<const uint4*>(values + i);
    const uint4 hi = *(reinterpret_cast<const uint4*>(values + i) + 1);
    *reinterpret_cast<uint4*>(dst + index[i]) = lo;
    *(reinterpret_cast<uint4*>(dst + index[i]) + 1) = hi;
}
__global__ void __launch_bounds__(P252_BLOCK) k_merkle4_update(const int32_t* __restrict__ tab, TagArg tag,
                                                               const uint32_t* __restrict__ index, unsigned shift,
                                                               const Scalar32* __restrict__ children, size_t n_children,
                                                               Scalar32* __restrict__ out, size_t k) {
    const size_t i = (size_t)blockIdx.x * P252_BLOCK + threadIdx.x;
    if (i >= k) return;
    const size_t node = index[i] >> shift;
    if (node * 4 >= n_children) return;  // (an out-of-range leaf index: nothing above it belongs to this tree)
    E29 s[WIDTH];
#pragma unroll
    for (int d = 0; d < NL; ++d) s[0].d[d] = tag.x0[d];
#pragma unroll
    for (int c = 0; c < 4; ++c) s[1 + c] = load_child_or_zero(children, n_children, node, 4, c);
    hades_permute<0x02u, true>(s, tab);
    store_scalar(out + node, s[1]);
}
__global__ void __launch_bounds__(P252_BLOCK) k_merkle4_update_coop(const int32_t* __restrict__ tab, TagArg tag,
                                                                    const uint32_t* __restrict__ index, unsigned shift,
                                                                    const Scalar32* __restrict__ children, size_t n_children,
                                                                    Scalar32* __restrict__ out, size_t k) {
    const size_t lane = (size_t)blockIdx.x * P252_BLOCK + threadIdx.x;
    const size_t i = lane / 8;
    if (i >= k) return;
    const size_t node = index[i] >> shift;
    if (node * 4 >= n_children) return;  // (whole group: the index is the group's)
    const int j = (int)(threadIdx.x & 7u);
    const int el = j < WIDTH ? j : WIDTH - 1;
    E29 mine = from_mont4(tag.w), unused;
    {
        const E29 child = load_child_or_zero(children, n_children, node, 4, el - 1);
#pragma unroll
        for (int d = 0; d < NL; ++d) mine.d[d] = el > 0 ? child.d[d] : mine.d[d];
    }
    unused = mine;
    WaveComm8 cm{j, (int)(((threadIdx.x & 63u) & ~7u) * 4u)};
    CoopLane<8> L = coop_lane<8>(tab, cm);
    hades_permute_coop<8, false>(mine, unused, tab, cm, L);
    if (j == 1) store_scalar(out + node, mine);
}

// ---- the other entry points for batches that cannot fill the chip (n <= 8,192): the same 8-lane groups, lane i holding
// state element i throughout.  Bit-identical results to the one-lane kernels below (P252_COOP_MAX_NODES=0 selects those). ----
__global__ void __launch_bounds__(P252_BLOCK) k_permute_coop(const int32_t* __restrict__ tab, const Scalar32* __restrict__ in,
                                                             Scalar32* __restrict__ out, size_t n) {
    const size_t lane = (size_t)blockIdx.x * P252_BLOCK + threadIdx.x;
    const size_t idx = lane / 8;
    if (idx >= n) return;
    const int j = (int)(threadIdx.x & 7u);
    WaveComm8 cm{j, (int)(((threadIdx.x & 63u) & ~7u) * 4u)};
    CoopLane<8> L = coop_lane<8>(tab, cm);
    E29 s = load_scalar(in + idx * WIDTH + L.row), unused = s;
    // (a loop of one iteration, as k_sponge_coop's loop of several: with the call in the kernel's entry block the register
    // allocator ends at 256 VGPRs + 4 AGPRs instead of 212)
#pragma unroll 1
    for (int once = 0; once < 1; ++once) hades_permute_coop<8>(s, unused, tab, cm, L);
    if (j < WIDTH) store_scalar(out + idx * WIDTH + j, s);
}

// the sponge of k_sponge with the state spread over a group: lane 0 the capacity element, lanes 1..4 the rate
template <bool TRUNC>
__device__ __forceinline__ void sponge_coop_body(const int32_t* __restrict__ tab, const TagArg& tag,
                                                 const Scalar32* __restrict__ in, unsigned in_len,
                                                 unsigned out_len, Scalar32* __restrict__ out, size_t n) {
    const size_t lane = (size_t)blockIdx.x * P252_BLOCK + threadIdx.x;
    const size_t idx = lane / 8;
    if (idx >= n) return;
    const int j = (int)(threadIdx.x & 7u);
    WaveComm8 cm{j, (int)(((threadIdx.x & 63u) & ~7u) * 4u)};
    CoopLane<8> L = coop_lane<8>(tab, cm);
    const Scalar32* my_in = in + idx * in_len;
    Scalar32* my_out = out + idx * out_len;
    const unsigned slot = (unsigned)L.row - 1u;  // my position in the rate (lane 0: none; lanes 5..7 shadow lane 4)
    const bool rate = L.row > 0;
    E29 s = from_mont4(tag.w), unused;
    if (rate) s = slot < in_len ? load_scalar(my_in + slot) : e29_zero();  // block 0 of the message
    unused = s;
    const unsigned absorb_blocks = (in_len + 3) / 4;
    const unsigned squeeze_blocks = (out_len + 3) / 4;
#pragma unroll 1
    for (unsigned it = 1; it < absorb_blocks + squeeze_blocks; ++it) {
        hades_permute_coop<8>(s, unused, tab, cm, L);
        if (it < absorb_blocks) {
            const unsigned e = it * 4 + slot;
            if (rate && e < in_len) add_e(s, load_scalar(my_in + e));  // Safe::add, scalar.rs:33-35
        } else {
            const unsigned o = (it - absorb_blocks) * 4 + slot;
            if (rate && j < WIDTH && o < out_len) store_output<TRUNC>(my_out + o, s);
        }
    }
}
__global__ void __launch_bounds__(P252_BLOCK) k_sponge_coop(const int32_t* __restrict__ tab, TagArg tag,
                                                            const Scalar32* __restrict__ in, unsigned in_len,
                                                            unsigned out_len, Scalar32* __restrict__ out, size_t n) {
    sponge_coop_body<false>(tab, tag, in, in_len, out_len, out, n);
}
__global__ void __launch_bounds__(P252_BLOCK) k_sponge_coop_trunc(const int32_t* __restrict__ tab, TagArg tag,
                                                                  const Scalar32* __restrict__ in, unsigned in_len,
                                                                  unsigned out_len, Scalar32* __restrict__ out, size_t n) {
    sponge_coop_body<true>(tab, tag, in, in_len, out_len, out, n);
}

// the opening of k_merkle4_path on a group: lane 0 the tag, lane 1 + slot the child in that slot of the node
__global__ void __launch_bounds__(P252_BLOCK) k_merkle4_path_coop(const int32_t* __restrict__ tab, TagArg tag,
                                                                  const Scalar32* __restrict__ leaves,
                                                                  const Scalar32* __restrict__ siblings,
                                                                  const uint8_t* __restrict__ positions, unsigned depth,
                                                                  Scalar32* __restrict__ roots, size_t n) {
    const size_t lane = (size_t)blockIdx.x * P252_BLOCK + threadIdx.x;
    const size_t idx = lane / 8;
    if (idx >= n) return;
    const int j = (int)(threadIdx.x & 7u);
    WaveComm8 cm{j, (int)(((threadIdx.x & 63u) & ~7u) * 4u)};
    CoopLane<8> L = coop_lane<8>(tab, cm);
    const E29 t = from_mont4(tag.w);
    E29 cur = load_scalar(leaves + idx);
    const Scalar32* sib = siblings + idx * depth * 3;
    const uint8_t* pos = positions + idx * depth;
    const unsigned slot = (unsigned)L.row - 1u;
#pragma unroll 1
    for (unsigned l = 0; l < depth; ++l) {
        const unsigned p = pos[l] & 3u;
        // children = the three siblings in order with `cur` inserted at slot p
        E29 s = t, unused;
        if (L.row > 0 && slot != p) s = load_scalar(sib + l * 3 + (slot < p ? slot : slot - 1));
#pragma unroll
        for (int k = 0; k < NL; ++k) s.d[k] = (L.row > 0 && slot == p) ? cur.d[k] : s.d[k];
        unused = s;
        hades_permute_coop<8, false>(s, unused, tab, cm, L);
        cur = cm.get<1>(s);  // the digest, to every lane
    }
    if (j == 0) store_scalar(roots + idx, cur);
}

// ---- generic sponge: n messages, same (in_len, out_len).  dusk-safe mechanics:
// absorb 4 elements per permutation into state[1..4]; first squeeze always permutes; 4 outputs per
// permutation.  One inlined permutation call site.
// HBM reads ("coalesced HBM loads of batched input scalars", north_star): a lane's message is in_len x 32 contiguous bytes,
// so an absorb block (4 scalars = 128 B) is exactly one cache line — when the message starts on a line boundary.  With an even
// in_len (config 4: 42 scalars = 10.5 lines) every other message starts in mid-line; fetched block by block, each of its
// blocks straddles two lines and every line is fetched twice, a permutation (0.18 ms) apart — by then it has left the
// caches: 1.19 x the algorithmic traffic (profiles/r02_pmc_k_sponge.txt).  LINES = true: every lane fetches WHOLE LINES —
// a lane whose message starts sh = 2 scalars into a line reads scalars 4 it + 2 .. 4 it + 5 at block it, absorbs the two
// carried over from the previous fetch plus the first two of this one and carries the other two (16 VGPRs, raw) across the
// permutation.  A message that ENDS in mid-line shares that line with the head of the next lane's message, which fetches it
// at its first block: the lane fetches its two tail scalars then as well (the same line at the same time: one HBM fetch)
// and parks them in LDS (64 B per lane, private: no barrier) until its last block.  Every line is touched once.  Taken
// when in_len is even and the array is 64-byte aligned (kernel-uniform); otherwise block by block as before. ----
template <bool LINES, bool TRUNC = false>
__device__ __forceinline__ void sponge_body(const int32_t* __restrict__ tab, const TagArg& tag, const Scalar32* __restrict__ in,
                                            unsigned in_len, unsigned out_len, Scalar32* __restrict__ out, size_t n) {
    __shared__ uint4 tail[LINES ? 4 : 1][LINES ? P252_BLOCK : 1];  // [half record][lane]: the two tail scalars of a message ending in mid-line
    const size_t idx = (size_t)blockIdx.x * P252_BLOCK + threadIdx.x;
    if (idx >= n) return;
    const Scalar32* my_in = in + idx * in_len;
    Scalar32* my_out = out + idx * out_len;
    E29 s[WIDTH];
    s[0] = from_mont4(tag.w);
#pragma unroll
    for (int k = 0; k < 4; ++k) s[1 + k] = e29_zero();
    const unsigned absorb_blocks = (in_len + 3) / 4;
    const unsigned squeeze_blocks = (out_len + 3) / 4;
    // my message starts `sh` scalars into its 128-byte line: 0, or 2 for every other message of an even in_len
    const unsigned sh = LINES ? (unsigned)((reinterpret_cast<uintptr_t>(my_in) >> 5) & 3u) : 0u;
    const bool shifted = sh != 0;
    const bool tail_half = LINES && ((sh + in_len) & 3u) == 2u && in_len >= 2;  // my message ends two scalars into a line
    const unsigned tl = LINES ? threadIdx.x : 0u;
    Raw32 c0 = raw_zero(), c1 = raw_zero();
    if (LINES) {  // the half line at the head of a shifted message (unshifted lanes: the same line block 0 fetches next)
        if (0 < in_len) c0 = load_raw(my_in);
        if (1 < in_len) c1 = load_raw(my_in + 1);
        if (tail_half) {  // (the next lane fetches this very line now, as the head of its message)
            const Raw32 t0 = load_raw(my_in + in_len - 2), t1 = load_raw(my_in + in_len - 1);
            tail[0][tl] = t0.lo;
            tail[1][tl] = t0.hi;
            tail[2][tl] = t1.lo;
            tail[3][tl] = t1.hi;
        }
    }
#pragma unroll 1
    for (unsigned it = 0; it < absorb_blocks + squeeze_blocks; ++it) {
        if (it > 0) hades_permute<0x1fu>(s, tab);
        if (it < absorb_blocks) {
            if (LINES) {
                Raw32 L[4];
                // the line this fetch addresses reaches past the end of my message: what is mine of it are the two tail scalars
                const bool parked = tail_half && it * 4 + sh + 4 > in_len;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned pos = it * 4 + sh + j;
                    L[j] = (pos < in_len && !parked) ? load_raw(my_in + pos) : raw_zero();
                }
                if (parked) {
                    L[0].lo = tail[0][tl];
                    L[0].hi = tail[1][tl];
                    L[1].lo = tail[2][tl];
                    L[1].hi = tail[3][tl];
                }
                const Raw32 e[4] = {raw_select(shifted, c0, L[0]), raw_select(shifted, c1, L[1]), raw_select(shifted, L[0], L[2]),
                                    raw_select(shifted, L[1], L[3])};
                c0 = L[2];
                c1 = L[3];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (it * 4 + k < in_len) add_e(s[1 + k], raw_to_e29(e[k]));  // Safe::add, scalar.rs:33-35 (block 0: 0 + e)
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned e = it * 4 + k;
                    if (e < in_len) add_e(s[1 + k], load_scalar(my_in + e));  // Safe::add, scalar.rs:33-35
                }
            }
        } else {
            const unsigned ob = (it - absorb_blocks) * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (ob + k < out_len) store_output<TRUNC>(my_out + ob + k, s[1 + k]);
        }
    }
}
__global__ void __launch_bounds__(P252_BLOCK) k_sponge(const int32_t* __restrict__ tab, TagArg tag,
                                                       const Scalar32* __restrict__ in, unsigned in_len,
                                                       unsigned out_len, Scalar32* __restrict__ out,
                                                       size_t n) {
    sponge_body<false>(tab, tag, in, in_len, out_len, out, n);
}
__global__ void __launch_bounds__(P252_BLOCK) k_sponge_lines(const int32_t* __restrict__ tab, TagArg tag,
                                                             const Scalar32* __restrict__ in, unsigned in_len,
                                                             unsigned out_len, Scalar32* __restrict__ out,
                                                             size_t n) {
    sponge_body<true>(tab, tag, in, in_len, out_len, out, n);
}
// Hash::finalize_truncated for a batch (hash.rs:164-183): the truncating output stage, every squeezed scalar — ONE launch
__global__ void __launch_bounds__(P252_BLOCK) k_sponge_trunc(const int32_t* __restrict__ tab, TagArg tag,
                                                             const Scalar32* __restrict__ in, unsigned in_len,
                                                             unsigned out_len, Scalar32* __restrict__ out,
                                                             size_t n) {
    sponge_body<false, true>(tab, tag, in, in_len, out_len, out, n);
}
__global__ void __launch_bounds__(P252_BLOCK) k_sponge_lines_trunc(const int32_t* __restrict__ tab, TagArg tag,
                                                                   const Scalar32* __restrict__ in, unsigned in_len,
                                                                   unsigned out_len, Scalar32* __restrict__ out,
                                                                   size_t n) {
    sponge_body<true, true>(tab, tag, in, in_len, out_len, out, n);
}

// ---- batched encryption / decryption (src/encryption.rs:62-95 -> dusk_safe::encrypt / decrypt) ----
// dusk-safe is not vendored in the reference and its encrypt() is pinned by no reference value, so the
// kernel does not hard-wire a construction: it INTERPRETS the sponge-call sequence the host hands it (api.cpp
// crypt_program(): one word per call, kind << 29 | len) on dusk-safe's sponge state machine (pinned
// by the reference KAT): absorb(e): pos_absorb == 4 -> permute, pos_absorb = 0; state[1 + pos_absorb] += e;
// after an absorb call pos_squeeze = 4.  squeeze: pos_squeeze == 4 -> permute, both positions = 0; emit
// state[1 + pos_squeeze].  Every position is wave-uniform, so all of this is scalar control flow around ONE inlined
// permutation.  Call kinds: 0 absorb the 2 secret scalars, 1 absorb the nonce, 2 squeeze `len` masks and apply them to
// the next `len` message elements (encrypt: cipher = message + mask; decrypt: message = cipher - mask, scalar.rs:67-74),
// 3 absorb the next `len` PLAINTEXT elements (decrypt reads back what it wrote), 4 squeeze the MAC (encrypt: stored
// as cipher[len]; decrypt: compared with it, scalar.rs:76-79 -> flags[i], 0 = Error::DecryptionFailed).
// Encrypt: in = messages[n][len], out = ciphers[n][len+1].  Decrypt: in = ciphers[n][len+1], out = messages[n][len].
__device__ __forceinline__ E29 lane_get(const E29 s[WIDTH], unsigned pos) {
    switch (pos) {  // pos is wave-uniform: a scalar branch, never a per-lane select
        case 0: return s[1];
        case 1: return s[2];
        case 2: return s[3];
        default: return s[4];
    }
}
__device__ __forceinline__ void lane_add(E29 s[WIDTH], unsigned pos, const E29& x) {
    switch (pos) {
        case 0: add_e(s[1], x); break;
        case 1: add_e(s[2], x); break;
        case 2: add_e(s[3], x); break;
        default: add_e(s[4], x); break;
    }
}

template <bool DECRYPT>
__global__ void __launch_bounds__(P252_BLOCK) k_crypt(const int32_t* __restrict__ tab, TagArg tag,
                                                      const Scalar32* __restrict__ in,
                                                      const Scalar32* __restrict__ secrets,
                                                      const Scalar32* __restrict__ nonces, unsigned len,
                                                      Scalar32* out, uint8_t* __restrict__ flags, size_t n,
                                                      const uint32_t* __restrict__ prog, unsigned n_calls) {
    const size_t idx = (size_t)blockIdx.x * P252_BLOCK + threadIdx.x;
    if (idx >= n) return;
    const Scalar32* my_in = in + idx * (DECRYPT ? len + 1 : len);
    Scalar32* my_out = out + idx * (DECRYPT ? len : len + 1);
    E29 s[WIDTH];
    s[0] = from_mont4(tag.w);
#pragma unroll
    for (int k = 1; k < WIDTH; ++k) s[k] = e29_zero();
    unsigned pos_absorb = 0, pos_squeeze = 0, masked = 0, absorbed = 0;
#pragma unroll 1
    for (unsigned ci = 0; ci < n_calls; ++ci) {
        const unsigned kind = prog[ci] >> 29, cnt = prog[ci] & 0x1fffffffu;
        const bool is_absorb = kind == 0 || kind == 1 || kind == 3;
#pragma unroll 1
        for (unsigned e = 0; e < cnt; ++e) {
            if ((is_absorb ? pos_absorb : pos_squeeze) == 4) {
                hades_permute<0x1fu>(s, tab);
                pos_absorb = 0;
                if (!is_absorb) pos_squeeze = 0;
            }
            if (is_absorb) {
                const Scalar32* src = kind == 0 ? secrets + 2 * idx + e
                                      : kind == 1 ? nonces + idx
                                                  : (DECRYPT ? my_out : my_in) + absorbed + e;
                lane_add(s, pos_absorb, load_scalar(src));  // Safe::add, scalar.rs:33-35
                ++pos_absorb;
            } else {
                const E29 v = lane_get(s, pos_squeeze);
                ++pos_squeeze;
                if (kind == 2) {
                    E29 x = load_scalar(my_in + masked + e);
                    if (!DECRYPT)
                        add_e(x, v);  // cipher = message + mask
                    else
                        sub_e(x, v);  // message = cipher - mask
                    store_scalar(my_out + masked + e, x);
                } else if (!DECRYPT) {
                    store_scalar(my_out + len, v);
                } else {
                    uint32_t w[8];
                    to_mont4(v, w);
                    const uint4 lo = *reinterpret_cast<const uint4*>(my_in + len);
                    const uint4 hi = *(reinterpret_cast<const uint4*>(my_in + len) + 1);
                    const bool same = w[0] == lo.x && w[1] == lo.y && w[2] == lo.z && w[3] == lo.w && w[4] == hi.x &&
                                      w[5] == hi.y && w[6] == hi.z && w[7] == hi.w;
                    flags[idx] = same ? 1 : 0;
                }
            }
        }
        if (is_absorb) pos_squeeze = 4;
        if (kind == 2) masked += cnt;
        if (kind == 3) absorbed += cnt;
    }
}

// k_crypt on lane groups (batches of <= 8,192 messages): lane 0 the capacity element, lane 1 + pos the rate position pos.
// Every position is wave-uniform, so "the lane at this position" is one compare; a message element decrypted by one
// lane and absorbed later by another travels through `out` (same wave, program order: the pointer is not __restrict__).
template <bool DECRYPT>
__global__ void __launch_bounds__(P252_BLOCK) k_crypt_coop(const int32_t* __restrict__ tab, TagArg tag,
                                                           const Scalar32* __restrict__ in,
                                                           const Scalar32* __restrict__ secrets,
                                                           const Scalar32* __restrict__ nonces, unsigned len,
                                                           Scalar32* out, uint8_t* __restrict__ flags, size_t n,
                                                           const uint32_t* __restrict__ prog, unsigned n_calls) {
    const size_t lane = (size_t)blockIdx.x * P252_BLOCK + threadIdx.x;
    const size_t idx = lane / 8;
    if (idx >= n) return;
    const int j = (int)(threadIdx.x & 7u);
    WaveComm8 cm{j, (int)(((threadIdx.x & 63u) & ~7u) * 4u)};
    CoopLane<8> L = coop_lane<8>(tab, cm);
    const Scalar32* my_in = in + idx * (DECRYPT ? len + 1 : len);
    Scalar32* my_out = out + idx * (DECRYPT ? len : len + 1);
    E29 s = from_mont4(tag.w), unused;
    if (L.row > 0) s = e29_zero();
    unused = s;
    unsigned pos_absorb = 0, pos_squeeze = 0, masked = 0, absorbed = 0;
#pragma unroll 1
    for (unsigned ci = 0; ci < n_calls; ++ci) {
        const unsigned kind = prog[ci] >> 29, cnt = prog[ci] & 0x1fffffffu;
        const bool is_absorb = kind == 0 || kind == 1 || kind == 3;
#pragma unroll 1
        for (unsigned e = 0; e < cnt; ++e) {
            if ((is_absorb ? pos_absorb : pos_squeeze) == 4) {
                hades_permute_coop<8>(s, unused, tab, cm, L);
                pos_absorb = 0;
                if (!is_absorb) pos_squeeze = 0;
            }
            if (is_absorb) {
                const Scalar32* src = kind == 0 ? secrets + 2 * idx + e
                                      : kind == 1 ? nonces + idx
                                                  : (DECRYPT ? my_out : my_in) + absorbed + e;
                if (L.row == (int)(1 + pos_absorb)) add_e(s, load_scalar(src));  // Safe::add, scalar.rs:33-35
                ++pos_absorb;
            } else {
                const bool mine = j == (int)(1 + pos_squeeze);  // (the lane itself, not its shadows: one store per element)
                ++pos_squeeze;
                if (!mine) continue;
                if (kind == 2) {
                    E29 x = load_scalar(my_in + masked + e);
                    if (!DECRYPT)
                        add_e(x, s);  // cipher = message + mask
                    else
                        sub_e(x, s);  // message = cipher - mask
                    store_scalar(my_out + masked + e, x);
                } else if (!DECRYPT) {
                    store_scalar(my_out + len, s);
                } else {
                    uint32_t w[8];
                    to_mont4(s, w);
                    const uint4 lo = *reinterpret_cast<const uint4*>(my_in + len);
                    const uint4 hi = *(reinterpret_cast<const uint4*>(my_in + len) + 1);
                    const bool same = w[0] == lo.x && w[1] == lo.y && w[2] == lo.z && w[3] == lo.w && w[4] == hi.x &&
                                      w[5] == hi.y && w[6] == hi.z && w[7] == hi.w;
                    flags[idx] = same ? 1 : 0;
                }
            }
        }
        if (is_absorb) pos_squeeze = 4;
        if (kind == 2) masked += cnt;
        if (kind == 3) absorbed += cnt;
    }
}

// ---- canonical-value outputs on device-resident scalars.  MASK250: finalize_truncated's post-processing (hash.rs:164-183):
// canonical value (Montgomery form dropped) & (2^250 - 1), written as the raw limbs that JubJubScalar::from_raw receives.
// Without the mask: BlsScalar::to_bytes (the 32 little-endian bytes of the canonical value; the reference uses the pair
// to_bytes / from_bytes at src/hades/round_constants.rs:66-67 and from_hex_str at src/hades.rs:131).
// redc(V * 2^5) = V * 2^5 / 2^261 = V / 2^256 = the canonical value.
// These two are the library's only HBM-bound kernels (64 B of traffic and one field product per scalar).  Measured
// (bench_tools/byte_format_bench.py, profiles/r02_byte_format.txt): what held them at 4.5 / 4.2 TB/s was VALU work, not the
// access pattern — one record per lane as two 16-byte loads runs exactly as fast as fully contiguous 1-KB wave
// instructions with a DPP pair swap, and non-temporal hints change nothing; canonicalising with the 2 conditional
// subtractions a tight reduction of a sub-modulus value needs instead of the general 5 gives 5.5 / 4.7 TB/s
// (HBM achievable: ~6.3). ----
template <bool MASK250>
__global__ void __launch_bounds__(P252_BLOCK) k_to_canonical(const Scalar32* in, Scalar32* out, size_t n) {  // (in place allowed)
    const size_t idx = (size_t)blockIdx.x * P252_BLOCK + threadIdx.x;
    if (idx >= n) return;
    const E29 x = load_scalar(in + idx);
    const int32_t c32[NL] = {32, 0, 0, 0, 0, 0, 0, 0, 0};
    A29 t;
    acc_zero(t);
    acc_mul(t, x, c32);
    const E29 canon = redc(t);  // in (V / 2^256 - p, V / 2^256] with V < 2^256: between -p and p
    uint32_t w[8];
    to_mont4<2>(canon, w);
    if (MASK250) w[7] &= 0x03ffffffu;  // TRUNCATION_MASK: keep the low 250 bits
    *reinterpret_cast<uint4*>(out + idx) = make_uint4(w[0], w[1], w[2], w[3]);
    *(reinterpret_cast<uint4*>(out + idx) + 1) = make_uint4(w[4], w[5], w[6], w[7]);
}

// ---- BlsScalar::from_bytes on n device-resident 32-byte records: little-endian canonical value v -> the Montgomery
// limbs of v * 2^256 mod p.  ok[i] = 1 iff v < p (from_bytes fails otherwise; the limbs written are those of v mod p).
// One generic product by r2 = 2^517 mod p (balanced digits, kernel argument): v * 2^517 / 2^261 = v * 2^256. ----
__global__ void __launch_bounds__(P252_BLOCK) k_from_canonical(const Scalar32* in, Scalar32* out, uint8_t* __restrict__ ok,
                                                               size_t n, Digits9 r2) {  // (in place allowed)
    const size_t idx = (size_t)blockIdx.x * P252_BLOCK + threadIdx.x;
    if (idx >= n) return;
    const uint4 lo = *reinterpret_cast<const uint4*>(in + idx);
    const uint4 hi = *(reinterpret_cast<const uint4*>(in + idx) + 1);
    const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    if (ok) {
        const uint32_t PW[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
        uint32_t borrow = 0;  // v - p borrows  <=>  v < p
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint64_t d = (uint64_t)w[k] - PW[k] - borrow;
            borrow = (uint32_t)(d >> 63);
        }
        ok[idx] = (uint8_t)borrow;
    }
    uint32_t m[8];
    to_mont4<2>(mul_c(from_mont4(w), r2.d), m);  // |v r2| / 2^261 < 2^250: the tight reduction leaves a value in (-1.1 p, 0.1 p)
    *reinterpret_cast<uint4*>(out + idx) = make_uint4(m[0], m[1], m[2], m[3]);
    *(reinterpret_cast<uint4*>(out + idx) + 1) = make_uint4(m[4], m[5], m[6], m[7]);
}

// ---- batched Merkle opening: recompute the root from a leaf and its sibling path (arity 4).
// Per level l the node hashed is Hash::digest(Merkle4, children) with children[pos[l]] = current value
// and the 3 siblings in the remaining slots in order; depth sequential permutations per lane.
// Layout: leaves[n], siblings[n][depth][3], positions[n][depth] (u8, 0..3), roots[n].
// HBM reads: a level's sibling triple is 96 B of the lane's depth x 96 contiguous bytes, so three of every four triples
// straddle a cache line, and every line is fetched twice, a permutation apart; the position bytes (one per level, 12 lanes
// to a line) once per level: 1.31 x the algorithmic traffic (profiles/r02_pmc_k_merkle4_path.txt).  LINES = true (depth a
// multiple of 4 and both arrays line-aligned, kernel-uniform — every lane's path then starts on a line boundary and the
// pattern is the same for all lanes): whole lines, one per level for three levels out of four (level 4 j + 3 needs none);
// the scalars a later level needs wait in LDS (<= 3 raw records = 96 B per lane, private to the lane: no barrier), and the
// position bytes of 16 levels are fetched at once (4 VGPRs). ----
template <bool LINES>
__device__ __forceinline__ void merkle4_path_body(const int32_t* __restrict__ tab, const TagArg& tag,
                                                  const Scalar32* __restrict__ leaves, const Scalar32* __restrict__ siblings,
                                                  const uint8_t* __restrict__ positions, unsigned depth,
                                                  Scalar32* __restrict__ roots, size_t n) {
    __shared__ uint4 carry[LINES ? 6 : 1][LINES ? P252_BLOCK : 1];  // [half record][lane]: conflict-free 16-byte accesses
    const size_t idx = (size_t)blockIdx.x * P252_BLOCK + threadIdx.x;
    if (idx >= n) return;
    E29 cur = load_scalar(leaves + idx);
    const Scalar32* sib = siblings + idx * depth * 3;
    const uint8_t* pos = positions + idx * depth;
    const unsigned t = threadIdx.x;
    auto put = [&](int slot, const Raw32& r) {
        carry[LINES ? 2 * slot : 0][LINES ? t : 0] = r.lo;
        carry[LINES ? 2 * slot + 1 : 0][LINES ? t : 0] = r.hi;
    };
    auto get = [&](int slot) {
        Raw32 r;
        r.lo = carry[LINES ? 2 * slot : 0][LINES ? t : 0];
        r.hi = carry[LINES ? 2 * slot + 1 : 0][LINES ? t : 0];
        return r;
    };
    uint32_t pw[4] = {0u, 0u, 0u, 0u};  // LINES: the position bytes of levels 16 j .. 16 j + 15
#pragma unroll 1
    for (unsigned l = 0; l < depth; ++l) {
        unsigned p;
        Raw32 ra, rb, rc;
        if (LINES) {
            if ((l & 15u) == 0) {  // (depth % 4 == 0 and the array is 4-byte aligned: whole words)
                const uint32_t* pq = reinterpret_cast<const uint32_t*>(pos + l);
#pragma unroll
                for (int k = 0; k < 4; ++k) pw[k] = l + 4 * k < depth ? pq[k] : 0u;
            }
            const unsigned word = (l >> 2) & 3u, byte = l & 3u;  // wave-uniform
            const uint32_t w = word == 0 ? pw[0] : word == 1 ? pw[1] : word == 2 ? pw[2] : pw[3];
            p = (w >> (8 * byte)) & 3u;
            const Scalar32* g = sib + (l & ~3u) * 3;  // the 3 lines (12 scalars) of this group of four levels
            switch (l & 3u) {                         // wave-uniform
                case 0: {  // line 0: the triple and the first scalar of the next one
                    ra = load_raw(g + 0), rb = load_raw(g + 1), rc = load_raw(g + 2);
                    put(0, load_raw(g + 3));
                    break;
                }
                case 1: {  // line 1
                    ra = get(0);
                    rb = load_raw(g + 4), rc = load_raw(g + 5);
                    const Raw32 x = load_raw(g + 6), y = load_raw(g + 7);
                    put(0, x);
                    put(1, y);
                    break;
                }
                case 2: {  // line 2
                    ra = get(0), rb = get(1);
                    rc = load_raw(g + 8);
                    const Raw32 x = load_raw(g + 9), y = load_raw(g + 10), z = load_raw(g + 11);
                    put(0, x);
                    put(1, y);
                    put(2, z);
                    break;
                }
                default: {  // nothing to fetch
                    ra = get(0), rb = get(1), rc = get(2);
                    break;
                }
            }
        } else {
            p = pos[l] & 3u;
            ra = load_raw(sib + l * 3 + 0), rb = load_raw(sib + l * 3 + 1), rc = load_raw(sib + l * 3 + 2);
        }
        E29 s[WIDTH];
#pragma unroll
        for (int k = 0; k < NL; ++k) s[0].d[k] = tag.x0[k];
        const E29 a = raw_to_e29(ra), b = raw_to_e29(rb), c = raw_to_e29(rc);
        // children = siblings with `cur` inserted at slot p (per-lane select, no divergence)
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            s[1].d[k] = p == 0 ? cur.d[k] : a.d[k];
            s[2].d[k] = p == 1 ? cur.d[k] : (p < 1 ? a.d[k] : b.d[k]);
            s[3].d[k] = p == 2 ? cur.d[k] : (p < 2 ? b.d[k] : c.d[k]);
            s[4].d[k] = p == 3 ? cur.d[k] : c.d[k];
        }
        hades_permute<0x02u, true>(s, tab);
        cur = s[1];
    }
    store_scalar(roots + idx, cur);
}
__global__ void __launch_bounds__(P252_BLOCK) P252_WAVES_ATTR k_merkle4_path(const int32_t* __restrict__ tab, TagArg tag,
                                                             const Scalar32* __restrict__ leaves,
                                                             const Scalar32* __restrict__ siblings,
                                                             const uint8_t* __restrict__ positions,
                                                             unsigned depth, Scalar32* __restrict__ roots,
                                                             size_t n) {
    merkle4_path_body<false>(tab, tag, leaves, siblings, positions, depth, roots, n);
}
__global__ void __launch_bounds__(P252_BLOCK) P252_WAVES_ATTR k_merkle4_path_lines(const int32_t* __restrict__ tab, TagArg tag,
                                                                   const Scalar32* __restrict__ leaves,
                                                                   const Scalar32* __restrict__ siblings,
                                                                   const uint8_t* __restrict__ positions,
                                                                   unsigned depth, Scalar32* __restrict__ roots,
                                                                   size_t n) {
    merkle4_path_body<true>(tab, tag, leaves, siblings, positions, depth, roots, n);
}

// ---- shader-clock probe: a measurement aid for bench.py, not part of the hashing path.  ONE wave reads the shader-clock
// counter (s_memtime) and the constant-rate real-time counter (s_memrealtime, 100 MHz on MI300-class parts) before and
// after sleeping for `spin_ticks` real-time ticks: (memtime delta) / (realtime delta) x 100 MHz = the shader clock over
// that interval.  Launched on a second stream while the hashing kernels run it shares their CUs (32 VGPRs: it fits beside
// three resident digest waves), so the figure is the clock UNDER LOAD — what GRBM_GUI_ACTIVE / duration gives in a
// rocprofv3 --pmc pass (profiles/r02_pmc_k_merkle4.txt), but inside an ordinary run.  A chain of 1,024 dependent v_add_u32 is
// timed with the same counter as a plausibility check (a fixed number of cycles per add whatever the clock).
// out[6] = {memtime0, realtime0, memtime after the chain, memtime1, realtime1, chain result}. ----
__global__ void __launch_bounds__(64) k_clock_probe(uint64_t* __restrict__ out, unsigned spin_ticks) {
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    const uint64_t m0 = __builtin_amdgcn_s_memtime();
    uint32_t x = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 1024; ++i) asm volatile("v_add_u32 %0, %0, %0" : "+v"(x));
    const uint64_t mc = __builtin_amdgcn_s_memtime();
    uint64_t r1 = __builtin_amdgcn_s_memrealtime();
    while (r1 - r0 < (uint64_t)spin_ticks) {
        __builtin_amdgcn_s_sleep(32);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const uint64_t m1 = __builtin_amdgcn_s_memtime();
    r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[0] = m0;
        out[1] = r0;
        out[2] = mc;
        out[3] = m1;
        out[4] = r1;
        out[5] = x;
    }
}

}  // namespace p252

// ---------------------------------------------------------------------------------------------
// launchers (C++ linkage, called from api.cpp)
// ---------------------------------------------------------------------------------------------
namespace p252 {

static inline unsigned grid_for(size_t n) { return (unsigned)((n + P252_BLOCK - 1) / P252_BLOCK); }

// largest batch that runs on the cooperative (several lanes per state) kernels: P252_COOP_MAX_NODES, default 16384
// (8 lanes per state up to 8,192 — every entry point; 4 lanes up to 16,384 — Merkle digests only); 0 = never
static size_t coop_max_nodes() {
    static const size_t v = [] {
        const char* e = std::getenv("P252_COOP_MAX_NODES");
        return e ? (size_t)std::strtoull(e, nullptr, 10) : (size_t)16384;
    }();
    return v;
}
// whole-cache-line fetches in the sponge and opening kernels (k_sponge_lines, k_merkle4_path_lines); P252_LINE_FETCH=0: off
static bool line_fetch() {
    static const bool on = [] {
        const char* e = std::getenv("P252_LINE_FETCH");
        return !(e && e[0] == '0');
    }();
    return on;
}
static inline bool coop8(size_t n) { return n <= coop_max_nodes() && n * 8 <= (size_t)65536; }

hipError_t launch_permute(const int32_t* tab, const void* in, void* out, size_t n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    if (coop8(n)) {
        hipLaunchKernelGGL(k_permute_coop, dim3(grid_for(n * 8)), dim3(P252_BLOCK), 0, st, tab,
                           static_cast<const Scalar32*>(in), static_cast<Scalar32*>(out), n);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_permute, dim3(grid_for(n)), dim3(P252_BLOCK), 0, st, tab,
                       static_cast<const Scalar32*>(in), static_cast<Scalar32*>(out), n);
    return hipGetLastError();
}

hipError_t launch_merkle4(const int32_t* tab, const TagArg& tag, const void* children, size_t n_children,
                          void* out, size_t n, hipStream_t st, unsigned arity, size_t pad_lanes, bool trunc250) {
    if (n == 0) return hipSuccess;
    if (trunc250) {  // digest_truncated: lane groups for what cannot fill the chip, the 3-waves-per-SIMD build otherwise
        if (coop8(n)) {
            const size_t lanes = n * 8;
            hipLaunchKernelGGL(k_merkle4_coop8_trunc, dim3(grid_for(lanes)), dim3(P252_BLOCK), 0, st, tab, tag,
                               static_cast<const Scalar32*>(children), n_children, static_cast<Scalar32*>(out), n, arity, lanes);
        } else {
            hipLaunchKernelGGL(k_merkle4_trunc, dim3(grid_for(n)), dim3(P252_BLOCK), 0, st, tab, tag,
                               static_cast<const Scalar32*>(children), n_children, static_cast<Scalar32*>(out), n, arity);
        }
        return hipGetLastError();
    }
    if (pad_lanes >= ((size_t)1 << 31)) pad_lanes = 0;  // (the padded kernels index with 32 bits)
    // launches of at most one wave per SIMD with 8 (4) lanes per node: the cooperative low-latency builds
    if (n <= coop_max_nodes() && n * 4 <= (size_t)65536) {
        const bool eight = n * 8 <= (size_t)65536;
        const size_t group = eight ? 8 : 4;
        const size_t want = n * group;
        // whole groups only: a partial last group would exchange with lanes that have already left the kernel
        const size_t lanes = ((want < pad_lanes ? pad_lanes : want) + group - 1) & ~(group - 1);
        if (eight)
            hipLaunchKernelGGL(k_merkle4_coop<8>, dim3(grid_for(lanes)), dim3(P252_BLOCK), 0, st, tab, tag,
                               static_cast<const Scalar32*>(children), n_children, static_cast<Scalar32*>(out), n, arity, lanes);
        else
            hipLaunchKernelGGL(k_merkle4_coop<4>, dim3(grid_for(lanes)), dim3(P252_BLOCK), 0, st, tab, tag,
                               static_cast<const Scalar32*>(children), n_children, static_cast<Scalar32*>(out), n, arity, lanes);
        return hipGetLastError();
    }
    if (n < pad_lanes) {
        hipLaunchKernelGGL(k_merkle4_pad, dim3(grid_for(pad_lanes)), dim3(P252_BLOCK), 0, st, tab, tag,
                           static_cast<const Scalar32*>(children), n_children, static_cast<Scalar32*>(out), n, arity, pad_lanes);
        return hipGetLastError();
    }
    // <= one wave per SIMD on the whole chip (256 CUs x 4 SIMDs x 64 lanes): the latency build (169 VGPRs: two waves per
    // SIMD).  It also takes launches of two and of FOUR waves per SIMD: at three resident waves a fourth runs alone
    // afterwards at a lone wave's latency (3 + 1), two and two do not (2^18 digests: see profiles/r02_ab_coop.txt).
    // (P252_LAT_WAVES: bit w set = launches of w waves per SIMD use it; default 0x16 = {1, 2, 4})
    static const unsigned lat_mask = [] {
        const char* e = std::getenv("P252_LAT_WAVES");
        return e ? (unsigned)std::strtoul(e, nullptr, 0) : 0x16u;
    }();
    const size_t waves_per_simd = (n + 65535) / 65536;
    if (waves_per_simd < 32 && ((lat_mask >> waves_per_simd) & 1u))
        hipLaunchKernelGGL(k_merkle4_lat, dim3(grid_for(n)), dim3(P252_BLOCK), 0, st, tab, tag,
                           static_cast<const Scalar32*>(children), n_children, static_cast<Scalar32*>(out), n, arity);
    else
        hipLaunchKernelGGL(k_merkle4, dim3(grid_for(n)), dim3(P252_BLOCK), 0, st, tab, tag,
                           static_cast<const Scalar32*>(children), n_children, static_cast<Scalar32*>(out), n, arity);
    return hipGetLastError();
}

hipError_t launch_scatter_scalars(const void* index, const void* values, void* dst, size_t k, size_t n_dst, void* n_bad, hipStream_t st) {
    if (k == 0) return hipSuccess;
    hipLaunchKernelGGL(k_scatter_scalars, dim3(grid_for(k)), dim3(P252_BLOCK), 0, st, static_cast<const uint32_t*>(index),
                       static_cast<const Scalar32*>(values), static_cast<Scalar32*>(dst), k, n_dst, static_cast<unsigned*>(n_bad));
    return hipGetLastError();
}

hipError_t launch_merkle4_update(const int32_t* tab, const TagArg& tag, const void* index, unsigned shift, const void* children,
                                 size_t n_children, void* out, size_t k, hipStream_t st) {
    if (k == 0) return hipSuccess;
    if (coop8(k))
        hipLaunchKernelGGL(k_merkle4_update_coop, dim3(grid_for(k * 8)), dim3(P252_BLOCK), 0, st, tab, tag,
                           static_cast<const uint32_t*>(index), shift, static_cast<const Scalar32*>(children), n_children,
                           static_cast<Scalar32*>(out), k);
    else
        hipLaunchKernelGGL(k_merkle4_update, dim3(grid_for(k)), dim3(P252_BLOCK), 0, st, tab, tag,
                           static_cast<const uint32_t*>(index), shift, static_cast<const Scalar32*>(children), n_children,
                           static_cast<Scalar32*>(out), k);
    return hipGetLastError();
}

hipError_t launch_sponge(const int32_t* tab, const TagArg& tag, const void* in, unsigned in_len,
                         unsigned out_len, void* out, size_t n, hipStream_t st, bool trunc250) {
    if (n == 0) return hipSuccess;
    if (trunc250) {
        if (coop8(n))
            hipLaunchKernelGGL(k_sponge_coop_trunc, dim3(grid_for(n * 8)), dim3(P252_BLOCK), 0, st, tab, tag,
                               static_cast<const Scalar32*>(in), in_len, out_len, static_cast<Scalar32*>(out), n);
        else if (line_fetch() && (in_len & 1u) == 0 && (reinterpret_cast<uintptr_t>(in) & 63u) == 0)
            hipLaunchKernelGGL(k_sponge_lines_trunc, dim3(grid_for(n)), dim3(P252_BLOCK), 0, st, tab, tag,
                               static_cast<const Scalar32*>(in), in_len, out_len, static_cast<Scalar32*>(out), n);
        else
            hipLaunchKernelGGL(k_sponge_trunc, dim3(grid_for(n)), dim3(P252_BLOCK), 0, st, tab, tag,
                               static_cast<const Scalar32*>(in), in_len, out_len, static_cast<Scalar32*>(out), n);
        return hipGetLastError();
    }
    if (coop8(n)) {
        hipLaunchKernelGGL(k_sponge_coop, dim3(grid_for(n * 8)), dim3(P252_BLOCK), 0, st, tab, tag,
                           static_cast<const Scalar32*>(in), in_len, out_len, static_cast<Scalar32*>(out), n);
        return hipGetLastError();
    }
    // whole-line fetches when every message starts on a 64-byte boundary (P252_LINE_FETCH=0: never — A/B and tests)
    if (line_fetch() && (in_len & 1u) == 0 && (reinterpret_cast<uintptr_t>(in) & 63u) == 0)
        hipLaunchKernelGGL(k_sponge_lines, dim3(grid_for(n)), dim3(P252_BLOCK), 0, st, tab, tag,
                           static_cast<const Scalar32*>(in), in_len, out_len, static_cast<Scalar32*>(out), n);
    else
        hipLaunchKernelGGL(k_sponge, dim3(grid_for(n)), dim3(P252_BLOCK), 0, st, tab, tag,
                           static_cast<const Scalar32*>(in), in_len, out_len, static_cast<Scalar32*>(out), n);
    return hipGetLastError();
}

hipError_t launch_crypt(bool decrypt, const int32_t* tab, const TagArg& tag, const void* in, const void* secrets,
                        const void* nonces, unsigned len, void* out, void* flags, size_t n, const uint32_t* prog,
                        unsigned n_calls, hipStream_t st) {
    if (n == 0) return hipSuccess;
    if (coop8(n)) {
        if (decrypt)
            hipLaunchKernelGGL(k_crypt_coop<true>, dim3(grid_for(n * 8)), dim3(P252_BLOCK), 0, st, tab, tag,
                               static_cast<const Scalar32*>(in), static_cast<const Scalar32*>(secrets),
                               static_cast<const Scalar32*>(nonces), len, static_cast<Scalar32*>(out),
                               static_cast<uint8_t*>(flags), n, prog, n_calls);
        else
            hipLaunchKernelGGL(k_crypt_coop<false>, dim3(grid_for(n * 8)), dim3(P252_BLOCK), 0, st, tab, tag,
                               static_cast<const Scalar32*>(in), static_cast<const Scalar32*>(secrets),
                               static_cast<const Scalar32*>(nonces), len, static_cast<Scalar32*>(out),
                               static_cast<uint8_t*>(flags), n, prog, n_calls);
        return hipGetLastError();
    }
    if (decrypt)
        hipLaunchKernelGGL(k_crypt<true>, dim3(grid_for(n)), dim3(P252_BLOCK), 0, st, tab, tag,
                           static_cast<const Scalar32*>(in), static_cast<const Scalar32*>(secrets),
                           static_cast<const Scalar32*>(nonces), len, static_cast<Scalar32*>(out),
                           static_cast<uint8_t*>(flags), n, prog, n_calls);
    else
        hipLaunchKernelGGL(k_crypt<false>, dim3(grid_for(n)), dim3(P252_BLOCK), 0, st, tab, tag,
                           static_cast<const Scalar32*>(in), static_cast<const Scalar32*>(secrets),
                           static_cast<const Scalar32*>(nonces), len, static_cast<Scalar32*>(out),
                           static_cast<uint8_t*>(flags), n, prog, n_calls);
    return hipGetLastError();
}

hipError_t launch_truncate250(const void* in, void* out, size_t n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_to_canonical<true>, dim3(grid_for(n)), dim3(P252_BLOCK), 0, st, static_cast<const Scalar32*>(in),
                       static_cast<Scalar32*>(out), n);
    return hipGetLastError();
}

hipError_t launch_to_canonical(const void* in, void* out, size_t n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_to_canonical<false>, dim3(grid_for(n)), dim3(P252_BLOCK), 0, st, static_cast<const Scalar32*>(in),
                       static_cast<Scalar32*>(out), n);
    return hipGetLastError();
}

hipError_t launch_from_canonical(const void* in, void* out, void* ok, size_t n, const Digits9& r2, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_from_canonical, dim3(grid_for(n)), dim3(P252_BLOCK), 0, st, static_cast<const Scalar32*>(in),
                       static_cast<Scalar32*>(out), static_cast<uint8_t*>(ok), n, r2);
    return hipGetLastError();
}

hipError_t launch_clock_probe(void* out6, unsigned spin_ticks, hipStream_t st) {
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, st, static_cast<uint64_t*>(out6), spin_ticks);
    return hipGetLastError();
}

hipError_t launch_merkle4_path(const int32_t* tab, const TagArg& tag, const void* leaves, const void* siblings,
                               const void* positions, unsigned depth, void* roots, size_t n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    if (coop8(n)) {
        hipLaunchKernelGGL(k_merkle4_path_coop, dim3(grid_for(n * 8)), dim3(P252_BLOCK), 0, st, tab, tag,
                           static_cast<const Scalar32*>(leaves), static_cast<const Scalar32*>(siblings),
                           static_cast<const uint8_t*>(positions), depth, static_cast<Scalar32*>(roots), n);
        return hipGetLastError();
    }
    // whole-line fetches when every lane's sibling path starts on a line boundary (depth x 96 B a multiple of 128 B)
    if (line_fetch() && depth != 0 && (depth & 3u) == 0 && (reinterpret_cast<uintptr_t>(siblings) & 127u) == 0 &&
        (reinterpret_cast<uintptr_t>(positions) & 3u) == 0)
        hipLaunchKernelGGL(k_merkle4_path_lines, dim3(grid_for(n)), dim3(P252_BLOCK), 0, st, tab, tag,
                           static_cast<const Scalar32*>(leaves), static_cast<const Scalar32*>(siblings),
                           static_cast<const uint8_t*>(positions), depth, static_cast<Scalar32*>(roots), n);
    else
        hipLaunchKernelGGL(k_merkle4_path, dim3(grid_for(n)), dim3(P252_BLOCK), 0, st, tab, tag,
                           static_cast<const Scalar32*>(leaves), static_cast<const Scalar32*>(siblings),
                           static_cast<const uint8_t*>(positions), depth, static_cast<Scalar32*>(roots), n);
    return hipGetLastError();
}

}  // namespace p252
