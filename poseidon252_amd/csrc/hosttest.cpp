// hosttest.cpp — CPU build of the DEVICE arithmetic headers, for unit tests only (tests/test_host_arith.py).
// Compiles fr29.hpp / hades29.hpp / tables.hpp with g++ so the exact code the kernels run can be
// checked against the oracle in a container without a GPU.  Not part of the product library and not
// a CPU fallback: libposeidon252_hip.so contains none of these entry points.
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>

#include <pthread.h>

#include <mutex>
#include <thread>

#include "coop29.hpp"
#include "fastdiv.hpp"
#include "hades29.hpp"
#include "_gen/assets.inc"

using namespace p252;

static const std::vector<int32_t>& tab29() {
    static std::vector<int32_t> t = [] {
        HadesTables T;
        derive_tables(ARC_BIN, MDS_BIN, T);
        return encode_tables29(T);
    }();
    return t;
}

extern "C" {
// openings.hip's FAST extraction kernel: record index / depth through fastdiv.hpp.  Returns the number of mismatches with the
// true quotient over recs[0..n) (and the first one in *first_bad); round 5's 40-bit multiply-shift is mode 1, kept to show the test
// sees what ADVICE r5 found.
size_t ht_fastdiv_check(const uint32_t* recs, size_t n, unsigned depth, int mode, uint32_t* first_bad) {
    const unsigned long long m = fast_div_reciprocal(depth), m40 = ((1ull << 40) / depth) + 1;
    size_t bad = 0;
    for (size_t i = 0; i < n; ++i) {
        const unsigned q = mode == 1 ? (unsigned)(((unsigned long long)recs[i] * m40) >> 40) : fast_div(recs[i], m);
        if (q != recs[i] / depth && !bad++ && first_bad) *first_bad = recs[i];
    }
    return bad;
}
// every rec in [lo, hi) — a dense sweep without marshalling 2^32 indices
size_t ht_fastdiv_sweep(uint64_t lo, uint64_t hi, unsigned depth) {
    const unsigned long long m = fast_div_reciprocal(depth);
    size_t bad = 0;
    for (uint64_t r = lo; r < hi; ++r) bad += fast_div((unsigned)r, m) != (unsigned)r / depth;
    return bad;
}
int ht_tables29_total() { return Tab29Layout::TOTAL; }
void ht_tables29(int32_t* out) { std::memcpy(out, tab29().data(), sizeof(int32_t) * Tab29Layout::TOTAL); }

// raw optimised-schedule constants as Montgomery limbs (for comparison with tests/pymodel.py)
//  order: c_first[5], full_add[8][5], mds_pre[25], then per sparse round w[4], d, b[4], add4; last_add[4]
size_t ht_tables_raw(uint64_t* out) {
    HadesTables T;
    derive_tables(ARC_BIN, MDS_BIN, T);
    size_t k = 0;
    auto put = [&](const FrHost& v) { std::memcpy(out + 4 * k++, v.l, 32); };
    for (int i = 0; i < 5; ++i) put(T.c_first[i]);
    for (int f = 0; f < 8; ++f) for (int i = 0; i < 5; ++i) put(T.full_add[f][i]);
    for (int i = 0; i < 5; ++i) for (int j = 0; j < 5; ++j) put(T.mds_pre[i][j]);
    for (int q = 0; q < 60; ++q) {
        for (int j = 0; j < 4; ++j) put(T.sparse[q].w[j]);
        put(T.sparse[q].d);
        for (int j = 0; j < 4; ++j) put(T.sparse[q].b[j]);
        put(T.sparse[q].add4);
    }
    for (int i = 0; i < 4; ++i) put(T.last_add[i]);
    return k;
}

// integer-ARMA constants (residues, canonical words): ai_kappa[8][5], ent_fix[3], ent_add[4], then per round
// q = 1..60: K_{q+1}, G_q; ex_fix[4], ex_add[4], F.  Returns the count, or 0 when mds.bin lacks the structure.
size_t ht_tables_armaint_raw(uint64_t* out) {
    HadesTables T;
    derive_tables(ARC_BIN, MDS_BIN, T);
    if (!T.int_ok) return 0;
    size_t k = 0;
    auto put = [&](const FrHost& v) { v.to_canonical(out + 4 * k++); };
    for (int f = 0; f < 8; ++f) for (int i = 0; i < 5; ++i) put(T.ai_kappa[f][i]);
    for (int i = 0; i < 3; ++i) put(T.ai_ent_fix[i]);
    for (int i = 0; i < 4; ++i) put(T.ai_ent_add[i]);
    for (int q = 0; q < 60; ++q) { put(T.ai_k[q]); put(T.ai_g[q]); }
    for (int i = 0; i < 4; ++i) put(T.ai_ex_fix[i]);
    for (int i = 0; i < 4; ++i) put(T.ai_ex_add[i]);
    put(T.ai_f);
    return k;
}

void ht_roundtrip29(const uint64_t* in, uint64_t* out, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        E29 e = from_mont4(reinterpret_cast<const uint32_t*>(in + 4 * i));
        to_mont4(e, reinterpret_cast<uint32_t*>(out + 4 * i));
    }
}

// out = a * b  (both Montgomery limbs): b is encoded like a table multiplier ("MP" form)
void ht_mul29(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
    const FrHost fMP = FrHost::pow2(261);
    for (size_t i = 0; i < n; ++i) {
        E29 ea = from_mont4(reinterpret_cast<const uint32_t*>(a + 4 * i));
        int32_t cb[NL];
        encode_balanced29(FrHost::from_limbs(b + 4 * i) * fMP, cb);
        A29 t;
        acc_zero(t);
        acc_mul(t, ea, cb);
        E29 r = redc(t);
        to_mont4(r, reinterpret_cast<uint32_t*>(out + 4 * i));
    }
}

// out = a^5 (Montgomery limbs): S-box output carries 2^-20, undone by an "MS"-encoded one
void ht_sbox29(const uint64_t* a, uint64_t* out, size_t n) {
    int32_t one_ms[NL];
    encode_balanced29(FrHost::pow2(261 + 20), one_ms);
    for (size_t i = 0; i < n; ++i) {
        E29 ea = from_mont4(reinterpret_cast<const uint32_t*>(a + 4 * i));
        E29 v = sbox(ea);
        A29 t;
        acc_zero(t);
        acc_mul(t, v, one_ms);
        E29 r = redc(t);
        to_mont4(r, reinterpret_cast<uint32_t*>(out + 4 * i));
    }
}

// schedule: 0 = integer ARMA (what the kernels run), 1 = all-sparse, 2 = integer MDS in all rounds
void ht_permute29_sched(const uint64_t* states, uint64_t* out, size_t n, int schedule) {
    const int32_t* tab = tab29().data();
    for (size_t i = 0; i < n; ++i) {
        E29 s[WIDTH];
        for (int k = 0; k < WIDTH; ++k) s[k] = from_mont4(reinterpret_cast<const uint32_t*>(states + (i * 5 + k) * 4));
        if (schedule == 0)
            hades_permute(s, tab);
        else if (schedule == 2)
            hades_permute_int(s, tab);
        else
            hades_permute_sparse(s, tab);
        for (int k = 0; k < WIDTH; ++k) to_mont4(s[k], reinterpret_cast<uint32_t*>(out + (i * 5 + k) * 4));
    }
}
void ht_permute29(const uint64_t* states, uint64_t* out, size_t n) { ht_permute29_sched(states, out, n, 0); }

// the Merkle4-digest path exactly as k_merkle4 runs it: lane 0 enters after its first S-box (hades_pre0, computed once
// per tag), only row 1 of the last linear layer is formed (OUT_ROWS = 0x02)
void ht_merkle4_digest29(const uint64_t* tag, const uint64_t* children, uint64_t* out, size_t n) {
    const int32_t* tab = tab29().data();
    const E29 x0 = hades_pre0(from_mont4(reinterpret_cast<const uint32_t*>(tag)), tab);
    for (size_t i = 0; i < n; ++i) {
        E29 s[WIDTH];
        s[0] = x0;
        for (int k = 0; k < 4; ++k) s[1 + k] = from_mont4(reinterpret_cast<const uint32_t*>(children + (i * 4 + k) * 4));
        hades_permute<0x02u, true>(s, tab);
        to_mont4(s[1], reinterpret_cast<uint32_t*>(out + i * 4));
    }
}

// the cooperative digest (coop29.hpp) with its eight lanes played by eight threads: Comm = shared slots + a barrier.
// Exactly the code k_merkle4_coop runs; the device replaces the exchange by ds_bpermute / DPP.
extern "C++" {
namespace {
struct HostGroup {
    pthread_barrier_t bar;
    E29 slot[8];
    std::mutex mu;
#if defined(P252_TRACK_BOUNDS)
    BoundTrack merged;
#endif
};
struct HostComm {
    HostGroup* g;
    int j;
    int lane() const { return j; }
    E29 exchange(const E29& v, int src) {
        g->slot[j] = v;
        pthread_barrier_wait(&g->bar);
        const E29 r = g->slot[src];
        pthread_barrier_wait(&g->bar);
        return r;
    }
    template <int M>
    E29 get(const E29& v) {
        return exchange(v, M);
    }
    E29 swap1(const E29& v) { return exchange(v, j ^ 1); }
};

// states[n][5] -> out[n][5] (DIGEST: only element 1 of each output is written, the rest left untouched)
template <int LANES, bool DIGEST>
void permute_coop_threads(const uint64_t* states, uint64_t* out, size_t n) {
    const int32_t* tab = tab29().data();
    HostGroup g;
    pthread_barrier_init(&g.bar, nullptr, LANES);
    std::vector<std::thread> th;
    constexpr int OWN = CoopLane<LANES>::OWN;
    for (int j = 0; j < LANES; ++j)
        th.emplace_back([&, j] {
            HostComm cm{&g, j};
            CoopLane<LANES> L = coop_lane<LANES>(tab, cm);
                for (size_t i = 0; i < n; ++i) {
                E29 s = from_mont4(reinterpret_cast<const uint32_t*>(states + (i * 5 + L.row) * 4));
                E29 s4 = from_mont4(reinterpret_cast<const uint32_t*>(states + (i * 5 + 4) * 4));
                hades_permute_coop<LANES, !DIGEST>(s, s4, tab, cm, L);
                if (j < OWN && (!DIGEST || j == 1)) to_mont4(s, reinterpret_cast<uint32_t*>(out + (i * 5 + j) * 4));
                if (LANES == 4 && j == 0 && !DIGEST) to_mont4(s4, reinterpret_cast<uint32_t*>(out + (i * 5 + 4) * 4));
            }
#if defined(P252_TRACK_BOUNDS)
            std::lock_guard<std::mutex> lk(g.mu);
            const BoundTrack& b = bound_track();
            if (b.max_col > g.merged.max_col) g.merged.max_col = b.max_col;
            if (b.max_top > g.merged.max_top) g.merged.max_top = b.max_top;
            if (b.max_top1 > g.merged.max_top1) g.merged.max_top1 = b.max_top1;
#endif
        });
    for (auto& t : th) t.join();
    pthread_barrier_destroy(&g.bar);
#if defined(P252_TRACK_BOUNDS)
    BoundTrack& b = bound_track();
    if (g.merged.max_col > b.max_col) b.max_col = g.merged.max_col;
    if (g.merged.max_top > b.max_top) b.max_top = g.merged.max_top;
    if (g.merged.max_top1 > b.max_top1) b.max_top1 = g.merged.max_top1;
#endif
}
}  // namespace
}  // extern "C++"

// lanes = 8 or 4 (the two group sizes of coop29.hpp); returns 0, or -1 for any other value
int ht_permute_coop(const uint64_t* states, uint64_t* out, size_t n, int lanes) {
    if (lanes == 8)
        permute_coop_threads<8, false>(states, out, n);
    else if (lanes == 4)
        permute_coop_threads<4, false>(states, out, n);
    else
        return -1;
    return 0;
}
// the Merkle4 digest as the cooperative kernels form it: state [tag, c0..c3], element 1 of the result
int ht_merkle4_digest_coop(const uint64_t* tag, const uint64_t* children, uint64_t* out, size_t n, int lanes) {
    std::vector<uint64_t> st(n * 20), res(n * 20, ~0ull);
    for (size_t i = 0; i < n; ++i) {
        std::memcpy(&st[i * 20], tag, 32);
        std::memcpy(&st[i * 20 + 4], children + i * 16, 128);
    }
    if (lanes == 8)
        permute_coop_threads<8, true>(st.data(), res.data(), n);
    else if (lanes == 4)
        permute_coop_threads<4, true>(st.data(), res.data(), n);
    else
        return -1;
    for (size_t i = 0; i < n; ++i) std::memcpy(out + i * 4, &res[i * 20 + 4], 32);
    return 0;
}

// Static worst-case |column| of every lazy accumulation in the schedules, assuming state digits
// < 2^29 (top digit < 2^24) and using the ACTUAL table constants; includes the < 2^61 the
// reduction itself adds.  Must stay below 2^63 (tests/test_host_arith.py).
// integer-MDS constants as Montgomery-limb words of the raw residues: kappa[68][5], G[60], F; returns count, or 0
// when mds.bin does not have the Cauchy structure
size_t ht_tables_int_raw(uint64_t* out) {
    HadesTables T;
    derive_tables(ARC_BIN, MDS_BIN, T);
    if (!T.int_ok) return 0;
    size_t k = 0;
    auto put = [&](const FrHost& v) { v.to_canonical(out + 4 * k++); };
    for (int r = 0; r < ROUNDS; ++r) for (int i = 0; i < 5; ++i) put(T.int_kappa[r][i]);
    for (int q = 0; q < PARTIAL_ROUNDS; ++q) put(T.int_g[q]);
    put(T.int_f);
    return k;
}

// 1 when derive_tables accepts the built-in assets; with tweak != 0 one byte of a COPY of mds.bin is changed first
// (entry tweak-1 of the matrix): the integer schedules must then be refused (int_ok = false -> p252_create fails).
int ht_int_ok(int tweak) {
    std::vector<unsigned char> mds(MDS_BIN, MDS_BIN + 25 * 32);
    if (tweak > 0 && tweak <= 25) mds[(size_t)(tweak - 1) * 32] ^= 1;
    HadesTables T;
    derive_tables(ARC_BIN, mds.data(), T);
    return T.int_ok ? 1 : 0;
}

double ht_max_column_bound29() { return max_column_bound29(tab29().data()); }

// instrumented build (-DP252_TRACK_BOUNDS): largest |column| seen by / |top digit| produced by any reduction since the
// last reset; -1 when the library was built without the instrumentation
void ht_bounds_reset() {
#if defined(P252_TRACK_BOUNDS)
    bound_track() = BoundTrack();
#endif
}
double ht_bounds_max_col() {
#if defined(P252_TRACK_BOUNDS)
    return (double)bound_track().max_col;
#else
    return -1.0;
#endif
}
double ht_bounds_max_top1() {
#if defined(P252_TRACK_BOUNDS)
    return (double)bound_track().max_top1;
#else
    return -1.0;
#endif
}
double ht_bounds_max_top() {
#if defined(P252_TRACK_BOUNDS)
    return (double)bound_track().max_top;
#else
    return -1.0;
#endif
}
}
