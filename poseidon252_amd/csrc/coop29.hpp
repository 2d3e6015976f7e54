// coop29.hpp — ONE permutation computed by a group of 8 (or 4) lanes: the low-latency build (kernels *_coop of kernels.hip).
//
// Why: a permutation is ~80 k dependent instructions for the lane that runs it, so a launch of at most one wave per SIMD
// (a tree's levels of <= 65,536 nodes, small batches) takes 0.165 ms whatever its size.  Such launches
// leave most lanes of the chip idle; this schedule spends them on the permutation's own coarse parallelism — whole field
// products, never parts of one (splitting a product over lanes was priced in §3.6: 1.2x) :
//   full rounds   the five S-boxes of a round run on five lanes (lane i holds state element i); the integer MDS layer
//                 gathers the five outputs and every lane forms its own row      3 sequential products instead of 15
//   partial phase W_q = x^5 G_q: the even lane of a pair squares twice (x^2, x^4) while the odd lane computes x G_q,
//                 one exchange, then both form x^4 * (x G_q) — the same bits on both, so every lane carries the
//                 complete history and the recurrence needs no exchange          3 sequential products instead of 4
//   exit          the four exit rows run on four lanes (and land where the next full round wants them)   1 instead of 4
// Sequential generic products per permutation: 8 x 3 + 3 + 60 x 3 + 1 + 1 = 209 (one lane: 365; the last 1 is the output
// scale F, one product per lane = per output element).  Everything else — the entry rows, the recurrence — is computed
// redundantly by all lanes, uniformly, with the constants in SGPRs as before.  Lane i ends up with element i of the
// permuted state, so a sponge keeps its state spread over the group from permutation to permutation.
//
// All arithmetic is the single-lane code of fr29.hpp / hades29.hpp (same tables, same reductions); the values are the
// same residues, in places in a different lazy representative, and the result is canonicalised by to_mont4 as always.
//
// Two group sizes.  LANES = 8 as described (launches of <= 8,192 nodes at one wave per SIMD).  LANES = 4 (<= 16,384
// nodes): lanes 0..3 hold state elements 0..3 and EVERY lane carries element 4 as well, so a full round is two S-boxes
// deep (mine, then element 4 redundantly) and two rows (mine, row 4): 8 x 6 + 3 + 60 x 3 + 1 + 1 = 233 sequential products.
//
// Comm: lane() = index within the group; get<M>(e) = the element held by lane M of my group; swap1(e) = the element held
// by lane ^ 1.  Device (kernels.hip): 8 lanes — ds_bpermute_b32; 4 lanes — DPP quad_perm broadcast; swap1 — DPP
// quad_perm [1,0,3,2].  Host (unit tests): one thread per lane and a barrier (hosttest.cpp).
#pragma once
#include "hades29.hpp"

namespace p252 {

// x^5 G / R'^5 on a pair of lanes (odd = true for the odd lane).  g = the nine digits of G_q (wave-uniform).
template <class Comm, class TP>
P252_HD E29 coop_sbox_g(const E29& u, TP g, bool odd, const RK& K, Comm& cm) {
    // (G is read unconditionally — nine scalar loads for the wave — and then selected per lane: a load inside the
    // conditional expression cannot be speculated and becomes nine exec-masked branches)
    int32_t gd[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) gd[k] = g[k];
    E29 op;
#pragma unroll
    for (int k = 0; k < NL; ++k) op.d[k] = odd ? gd[k] : u.d[k];
    A29 t;
    acc_zero_w(t, K);
    acc_mul(t, u, op.d);  // even: u^2 / R'; odd: u G / R'
    const E29 t1 = redc_w(t, K);
    acc_zero_w(t, K);
    acc_sqr(t, t1);  // even: u^4 / R'^3 (odd: unused)
    const E29 t2 = redc_w(t, K);
    E29 mine;
#pragma unroll
    for (int k = 0; k < NL; ++k) mine.d[k] = odd ? t1.d[k] : t2.d[k];
    const E29 other = cm.swap1(mine);
    // even: u^4 * (u G); odd: (u G) * u^4 — the same 81 digit products summed (exactly) into the same columns: identical bits
    // (W_q leaves as wide digits: it only meets the recurrence and the exit rows)
    acc_zero_w<true>(t, K);
    acc_mul(t, mine, other.d);
    return redc_w<true>(t, K);
}

template <int QM, class Comm, class TP>
P252_HD void coop_ai_round(E29 Us[HIST], E29 Ws[HIST], TP ab, TP kg, bool odd, const RK& K, Comm& cm) {
    Ws[QM] = coop_sbox_g(Us[QM], kg + NL, odd, K, cm);
    ai_recur<QM>(Us, Ws, ab, kg, K);
}

// a table entry at a per-lane index, loaded where the source says (see merkle4_digest_coop)
template <class TP>
P252_HD int32_t lane_const(TP tab, int index) {
    return tab[index];
}
// "this value is needed here": the load that produces it cannot be sunk below this point (emits no instruction)
P252_HD void pin_loaded(int32_t& x) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(x));
#endif
}

// my row of the integer MDS layer: sum_m hn[m] xs[m] + kappa, one digit step (hn = my five Hankel entries, in VGPRs)
P252_HD E29 coop_int_row(const E29 xs[WIDTH], const int32_t hn[WIDTH], const int32_t kap[NL]) {
    R29 t;
    row_set_c(t, kap);
#pragma unroll
    for (int m = 0; m < WIDTH; ++m) row_mac(t, xs[m], hn[m]);
    return row_redc1(t);
}

// What a lane keeps for the whole kernel: its place in the group and its rows of the small constant tables.  Per-lane
// table rows are vector loads (~1 us of latency) with nothing to overlap them at their point of use, which is where the
// optimiser sinks them; so they are fetched ahead: hn / c0 once per kernel, the kappa of round f + 1 while round f runs
// (round 7 fetches round 0's for the next permutation of the same kernel; round 4's is fetched ahead of the exit row),
// the exit row ahead of the entry rows — each pinned (pin_loaded) where it must have been issued.
template <int LANES>
struct CoopLane {
    static constexpr int OWN = LANES == 8 ? WIDTH : 4;  // state elements that live on a lane of their own
    int lane, row;        // row = my state element = my row of the linear layers (lanes >= OWN shadow the last one)
    bool odd;
    int32_t hn[WIDTH];    // my row of the integer MDS matrix (Hankel: h[row + m])
    int32_t c0[NL];       // first round constant of my element
    int32_t kap[NL];      // kappa of the NEXT full round to run, my row (round 0's between permutations)
};
template <class TP>
P252_HD void coop_load_kappa(int32_t kap[NL], TP tab, int f, int row) {
#pragma unroll
    for (int k = 0; k < NL; ++k) kap[k] = lane_const(tab, Tab29Layout::AI_KAPPA + (f * WIDTH + row) * NL + k);
}
template <int LANES, class Comm, class TP>
P252_HD CoopLane<LANES> coop_lane(TP tab, Comm& cm) {
    typedef Tab29Layout Lay;
    CoopLane<LANES> L;
    L.lane = cm.lane();
    L.row = L.lane < CoopLane<LANES>::OWN ? L.lane : CoopLane<LANES>::OWN - 1;
    L.odd = (L.lane & 1) != 0;
#pragma unroll
    for (int m = 0; m < WIDTH; ++m) L.hn[m] = lane_const(tab, Lay::INT_N + L.row + m);
#pragma unroll
    for (int k = 0; k < NL; ++k) L.c0[k] = lane_const(tab, Lay::C_FIRST + L.row * NL + k);
    coop_load_kappa(L.kap, tab, 0, L.row);
    return L;
}

// One permutation on a group.  In: s = my state element (lane i < OWN: element i; LANES = 8: lanes 5..7 anything),
// s4 = element 4 on every lane (LANES = 4 only).  Out: s = my element of the permuted state (lane i < OWN), s4 = element 4
// (LANES = 4), at the reference's Montgomery scale (lazy; to_mont4 canonicalises).  ROW4 = false (a digest: only element
// 1 is squeezed) skips the last layer's row 4 and its scaling on the 4-lane groups.
template <int LANES, bool ROW4 = true, class Comm, class TP>
P252_HD void hades_permute_coop(E29& s, E29& s4, TP tab, Comm& cm, CoopLane<LANES>& L) {
    static_assert(LANES == 8 || LANES == 4, "group sizes: 8 (five S-boxes side by side) or 4 (element 4 on every lane)");
    typedef Tab29Layout Lay;
    constexpr int RF = FULL_ROUNDS / 2;
    const RK K = make_rk();
    add_c(s, L.c0);
    if (LANES == 4) add_c(s4, tab + Lay::C_FIRST + 4 * NL);
    E29 xs[WIDTH];
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
#pragma unroll 1
        for (int f = half * RF; f < (half + 1) * RF; ++f) {
            int32_t kap_next[NL];
            if (f != RF - 1) coop_load_kappa(kap_next, tab, (f + 1) % (2 * RF), L.row);  // (round 3 has no row: the entry)
            const E29 x = sbox_w<true>(s, K);  // (wide digits: the outputs only meet the integer rows)
            xs[0] = cm.template get<0>(x);
            xs[1] = cm.template get<1>(x);
            xs[2] = cm.template get<2>(x);
            xs[3] = cm.template get<3>(x);
            if (LANES == 8)
                xs[4] = cm.template get<4>(x);
            else
                xs[4] = sbox_w<true>(s4, K);
            if (f != RF - 1) {
                s = coop_int_row(xs, L.hn, L.kap);
                if (LANES == 4 && (ROW4 || f != 2 * RF - 1))
                    s4 = int_row(xs, tab + Lay::INT_N + 4, tab + Lay::AI_KAPPA + (f * WIDTH + 4) * NL);
#pragma unroll
                for (int k = 0; k < NL; ++k) L.kap[k] = kap_next[k];
            }
        }
        if (half == 0) {
            // my exit row's constants (hidden by the entry and the 60 rounds below)
            const int er = L.lane < 4 ? L.lane : 3;
            int32_t exn[2 * NL], exfix[NL], exadd[NL];
#pragma unroll
            for (int k = 0; k < 2 * NL; ++k) exn[k] = lane_const(tab, Lay::AI_EX_N + er * 2 * NL + k);
#pragma unroll
            for (int k = 0; k < NL; ++k) {
                exfix[k] = lane_const(tab, Lay::AI_EX_FIX + er * NL + k);
                exadd[k] = lane_const(tab, Lay::AI_EX_ADD + er * NL + k);
            }
            E29 Us[HIST], Ws[HIST];
            arma_entry(xs, Us, Ws, tab, K);  // redundantly on every lane: 3 products, and no exchange afterwards
#pragma unroll
            for (int k = 0; k < 2 * NL; ++k) pin_loaded(exn[k]);
#pragma unroll
            for (int k = 0; k < NL; ++k) {
                pin_loaded(exfix[k]);
                pin_loaded(exadd[k]);
            }
#pragma unroll 1
            for (int it = 0; it < PARTIAL_ROUNDS / HIST; ++it) {
                const TP kg = tab + Lay::AI_KG + it * HIST * 2 * NL;
                coop_ai_round<1>(Us, Ws, tab + Lay::AI_AB, kg, L.odd, K, cm);
                coop_ai_round<2>(Us, Ws, tab + Lay::AI_AB, kg + 2 * NL, L.odd, K, cm);
                coop_ai_round<3>(Us, Ws, tab + Lay::AI_AB, kg + 4 * NL, L.odd, K, cm);
                coop_ai_round<4>(Us, Ws, tab + Lay::AI_AB, kg + 6 * NL, L.odd, K, cm);
                coop_ai_round<0>(Us, Ws, tab + Lay::AI_AB, kg + 8 * NL, L.odd, K, cm);
            }
            coop_load_kappa(L.kap, tab, RF, L.row);  // round 4's, hidden by the exit row and the S-box that follow
            const E29* const us[4] = {&Us[3], &Us[4], &Us[0], &Us[1]};
            const E29* const ws[4] = {&Ws[2], &Ws[3], &Ws[4], &Ws[0]};
            const E29 r = exit_row(us, ws, exn, exfix, exadd, K);
#pragma unroll
            for (int k = 0; k < NL; ++k) s.d[k] = L.lane < 4 ? r.d[k] : Us[1].d[k];  // lanes 0..3: my row; lane 4: U_61
            s4 = Us[1];
        }
    }
    // the scale F of the output (the last round's row was formed above; tight reduction: what to_mont4 expects)
    s = mul_c(s, tab + Lay::AI_F);
    if (LANES == 4 && ROW4) s4 = mul_c(s4, tab + Lay::AI_F);
}

}  // namespace p252
