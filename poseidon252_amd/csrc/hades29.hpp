// hades29.hpp — the Hades permutation on E29 lazy residues (one state per lane).
//
// Replaces Hades::perm (src/hades/permutation.rs:105-123) with its ARC / S-box / MDS steps
// (src/hades/permutation/scalar.rs:39-64).  Two algebraically equivalent schedules, both producing
// the reference's field elements (tables.hpp derives their constants):
//   hades_permute_sparse  4 full | 60 sparse partial rounds (12 mults, 8 reductions each) | 4 full
//   hades_permute         4 full (the 4th with the entry matrix) | 4 entry rounds (3 + <=4 mults, 4 redc)
//                         | 56 "ARMA" partial rounds (3 + 8 mults, 4 reductions each) | exit (28 mults,
//                         4 reductions) | 4 full; state re-scaled after every linear layer so that one
//                         coefficient per row is the free constant tau (tables.hpp step 4)   <- kernels
//
// TP is any pointer-like giving int32 digits: tab[i].  In kernels it is a wave-uniform pointer so
// the compiler keeps constants in SGPRs (s_load) and feeds them to v_mad_i64_i32 as scalar operands.
#pragma once
#include "fr29.hpp"
#include "tables.hpp"

#ifndef P252_ARMA_UNROLL
#define P252_ARMA_UNROLL 4
#endif

namespace p252 {

// One full round: state <- Mat * sbox(state) + add     (ARC of this round was folded into the
// previous layer's `add`).  5 S-boxes (15 mults, 15 redc) + 25 products + 5 redc.
// ROWS < 5 computes only the first ROWS output lanes (the last round of a digest needs lane 1 only).
// unit0: column 0 of `mat` is the constant tau (scaled schedule) — its 5 products are plain additions.
template <class TP>
P252_HD void full_round(E29 s[WIDTH], TP mat, TP add, unsigned rows = 0x1fu, bool unit0 = false) {
    E29 v[WIDTH];
#pragma unroll
    for (int i = 0; i < WIDTH; ++i) v[i] = sbox(s[i]);
#pragma unroll
    for (int k = 0; k < WIDTH; ++k) {
        if (!((rows >> k) & 1u)) continue;  // wave-uniform
        A29 t;
        acc_set_hi_c(t, add + k * NL);
        if (unit0)  // wave-uniform
            acc_add_hi(t, v[0]);
        else
            acc_mul(t, v[0], mat + (k * WIDTH) * NL);
#pragma unroll
        for (int j = 1; j < WIDTH; ++j) acc_mul(t, v[j], mat + (k * WIDTH + j) * NL);  // MDS[k][j] * state[j]
        s[k] = redc(t);
    }
}

// One partial round in sparse form: v = sbox(s4);  s4' = <w, s[0..3]> + d*v + add4;  s_i' = s_i + b_i*v.
// 3 + 9 products-of-elements, 3 + 5 redc.  Returns v.  AXPY=false skips the lane 0..3 update.
template <class TP>
P252_HD E29 partial_round(E29 s[WIDTH], TP sp, bool axpy = true) {
    typedef Tab29Layout Lay;
    const E29 v = sbox(s[4]);
    A29 t;
    acc_set_hi_c(t, sp + Lay::SP_ADD4);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc_mul(t, s[j], sp + Lay::SP_W + j * NL);
    acc_mul(t, v, sp + Lay::SP_D);
    const E29 y4 = redc(t);
    if (axpy) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            A29 u;
            acc_set_hi(u, s[i]);
            acc_mul(u, v, sp + Lay::SP_B + i * NL);
            s[i] = redc(u);
        }
    }
    s[4] = y4;
    return v;
}

// ---- schedule 1: all 60 partial rounds in sparse form (kept as an independent cross-check) ----
template <class TP>
P252_HD void hades_permute_sparse(E29 s[WIDTH], TP tab) {
    typedef Tab29Layout Lay;
    constexpr int RF = FULL_ROUNDS / 2;
#pragma unroll
    for (int i = 0; i < WIDTH; ++i) add_c(s[i], tab + Lay::C_FIRST + i * NL);
#pragma unroll 1
    for (int r = 0; r < ROUNDS; ++r) {
        if (r < RF || r >= RF + PARTIAL_ROUNDS) {
            const int f = r < RF ? r : r - PARTIAL_ROUNDS;
            full_round(s, tab + (r == RF - 1 ? Lay::MDS_PRE : Lay::MDS), tab + Lay::FULL_ADD + f * WIDTH * NL);
        } else {
            partial_round(s, tab + Lay::SPARSE + (r - RF) * Lay::SPARSE_STRIDE);
            if (r == RF + PARTIAL_ROUNDS - 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) add_c(s[i], tab + Lay::LAST_ADD + i * NL);
            }
        }
    }
}

// ---- schedule 2: ARMA partial rounds ----
// History layout while in the ARMA phase (q = index of the partial round about to run, 5..60):
//   h[0..3] = u_q, u_{q-1}, u_{q-2}, u_{q-3}      (S-box inputs, constants included)
//   h[4..8] = v_{q-1}, v_{q-2}, v_{q-3}, v_{q-4}, (free)   -> after the S-box: v_q .. v_{q-4}
// One step:  v_q = sbox(u_q);  u_{q+1} = sum a_m u_{q+1-m} + sum beta_n v_{q-n} + kappa_{q+1}.
template <class TP>
P252_HD void arma_round(E29 h[9], TP tab, int q) {
    typedef Tab29Layout Lay;
    // shift v history, newest first
    h[8] = h[7];
    h[7] = h[6];
    h[6] = h[5];
    h[5] = h[4];
    h[4] = sbox(h[0]);
    A29 t;
    acc_set_hi_c(t, tab + Lay::ARMA_KAPPA + (q + 1 - 6) * NL);
#pragma unroll
    for (int m = 0; m < 4; ++m) acc_mul(t, h[m], tab + Lay::ARMA_A + m * NL);
#pragma unroll
    for (int n = 0; n < 5; ++n) {
        if (n == 3)
            acc_add_hi(t, h[4 + n]);  // scaled schedule: beta_3 * lam^4 == tau, the free constant
        else
            acc_mul(t, h[4 + n], tab + Lay::ARMA_BETA + n * NL);
    }
    const E29 unew = redc(t);
    h[3] = h[2];
    h[2] = h[1];
    h[1] = h[0];
    h[0] = unew;
}

// After round 60: h[0..3] = u_61..u_58, h[4..7] = v_60..v_57.  Recover lanes 0..3 of the state
// (closing constants included); lane 4 = u_61.
template <class TP>
P252_HD void arma_exit(const E29 h[9], E29 s[WIDTH], TP tab) {
    typedef Tab29Layout Lay;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        A29 t;
        acc_set_hi_c(t, tab + Lay::EXIT_ADD + i * NL);
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // Gy[i][r] multiplies u_{58+r} = h[3-r]; Gv[i][r] multiplies v_{57+r} = h[7-r]
            acc_mul(t, h[3 - r], tab + Lay::EXIT_GY + (i * 4 + r) * NL);
            if (r == 3)
                acc_add_hi(t, h[7 - r]);  // scaled schedule: the coefficient of v_60 is tau in every row
            else
                acc_mul(t, h[7 - r], tab + Lay::EXIT_GV + (i * 4 + r) * NL);
        }
        s[i] = redc(t);
    }
    s[4] = h[0];
}

// Entry rounds q = 1..4 of the partial phase.  Full round 3 (matrix MDS_ENTRY) has left the projections
// p_q = c^T A^(q-1) L_1 + k_{q+1} in s[0..3] and u_1 in h[0]; then
//   v_q = sbox(u_q);   u_{q+1} = p_q + sum_{n=0..3} g_n v_{q-n}      (v_j = 0 for j < 1: zero history)
// s[0..3] rotate so that s[0] is always the current projection.
template <class TP>
P252_HD void entry_round(E29 s[WIDTH], E29 h[9], TP tab, int q) {
    typedef Tab29Layout Lay;
    h[7] = h[6];
    h[6] = h[5];
    h[5] = h[4];
    h[4] = sbox(h[0]);
    A29 t;
    acc_set_hi(t, s[0]);
#pragma unroll
    for (int n = 0; n < 4; ++n)
        if (n < q) acc_mul(t, h[4 + n], tab + Lay::ENTRY_G + n * NL);  // v_{q-n} exists only for n < q (wave-uniform)
    const E29 unew = redc(t);
    h[3] = h[2];
    h[2] = h[1];
    h[1] = h[0];
    h[0] = unew;
    s[0] = s[1];
    s[1] = s[2];
    s[2] = s[3];
}

// One loop with a wave-uniform phase switch, so each large unrolled body exists once in the
// instruction stream.  Phases (step): 0-3 full (3 = entry matrix) | 4-7 entry q=1..4 | 8-63 ARMA
// q=5..60 | 64 exit | 65-68 full.  OUT_ROWS: bit k set = lane k of the result is needed (a Merkle4
// digest needs lane 1 only, so the last round computes 1 of its 5 rows).
template <unsigned OUT_ROWS = 0x1fu, int ARMA_UNROLL_T = P252_ARMA_UNROLL, class TP>
P252_HD void hades_permute(E29 s[WIDTH], TP tab) {
    typedef Tab29Layout Lay;
    constexpr int RF = FULL_ROUNDS / 2;
    constexpr int N_ENTRY = 4;
    constexpr int STEP_ARMA0 = RF + N_ENTRY;                           // 8
    constexpr int ARMA_UNROLL = ARMA_UNROLL_T;  // ARMA rounds per loop step (history shifts become renames)
    static_assert((PARTIAL_ROUNDS - N_ENTRY) % ARMA_UNROLL == 0, "56 ARMA rounds must split evenly");
    constexpr int STEP_EXIT = STEP_ARMA0 + (PARTIAL_ROUNDS - N_ENTRY) / ARMA_UNROLL;
    constexpr int STEP_END = STEP_EXIT + 1 + RF;                       // 69
#pragma unroll
    for (int i = 0; i < WIDTH; ++i) add_c(s[i], tab + Lay::C_FIRST + i * NL);
    E29 h[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) h[i] = e29_zero();
    // the row mask of the last round must look like a run-time value: with a visible constant the
    // compiler peels the last iteration and emits a SECOND copy of the 50 KB full-round body
    // (measured: 136 KB code, I-cache thrash, 2.7e8 -> 1.7e8 perm/s)
    unsigned last_rows = OUT_ROWS;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(last_rows));
#endif
#pragma unroll 1
    for (int step = 0; step < STEP_END; ++step) {
        if (step < RF || step > STEP_EXIT) {
            const bool entry = step == RF - 1;
            const int f = step < RF ? step : step - STEP_EXIT - 1 + RF;
            const unsigned rows = step == STEP_END - 1 ? last_rows : 0x1fu;
            // scaled schedule: one matrix per round; column 0 is the free constant except in the entry
            // round (f = 3) and in the last round (f = 7, which must output the true state)
            full_round(s, tab + Lay::SC_MATS + f * WIDTH * WIDTH * NL, tab + Lay::SC_ADDS + f * WIDTH * NL, rows,
                       f != RF - 1 && f != FULL_ROUNDS - 1);
            if (entry) h[0] = s[4];  // u_1
        } else if (step < STEP_ARMA0) {
            entry_round(s, h, tab, step - RF + 1);
        } else if (step < STEP_EXIT) {
#pragma unroll
            for (int r = 0; r < ARMA_UNROLL; ++r) arma_round(h, tab, (step - STEP_ARMA0) * ARMA_UNROLL + r + 5);
        } else {
            arma_exit(h, s, tab);
        }
    }
}

// ---- schedule 3: integer MDS (tables.hpp step 5) ----
// The MDS matrix is (R/L) * N with N[i][j] = L/(i+j+5) a one-digit integer, and x -> x^5 is homogeneous,
// so the field factor travels in the scale of the stored state and every round is
//     X_j = sbox(Z_j)  (all lanes / lane 4 only; in partial rounds lane 4 is then multiplied by G_k — the
//                       one generic product of the round — so that all five X_j carry the same scale)
//     Z'_i = (sum_j N_ij X_j + kappa_i) / 2^29          45 one-digit MACs + 1 Montgomery digit step per lane
// A full round is 15 + 0 generic products (reference: 15 + 25), a partial round 3 + 1 (reference: 3 + 25).
// One multiplication by F per output lane restores the reference's Montgomery scale at the end.
// n = &h[i] for output lane i (N is a Hankel matrix: N[i][j] = h[i+j], nine distinct values).
template <class TP>
P252_HD E29 int_row(const E29 x[WIDTH], TP n, TP kappa) {
    R29 t;
    row_set_c(t, kappa);
#pragma unroll
    for (int j = 0; j < WIDTH; ++j) row_mac(t, x[j], n[j]);
    return row_redc1(t);
}

template <unsigned OUT_ROWS = 0x1fu, class TP>
P252_HD void hades_permute_int(E29 s[WIDTH], TP tab) {
    typedef Tab29Layout Lay;
    constexpr int RF = FULL_ROUNDS / 2;
#pragma unroll
    for (int i = 0; i < WIDTH; ++i) add_c(s[i], tab + Lay::C_FIRST + i * NL);
#pragma unroll 1
    for (int k = 0; k < ROUNDS; ++k) {
        const bool full = k < RF || k >= RF + PARTIAL_ROUNDS;  // wave-uniform
        E29 x[WIDTH];
        if (full) {
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = sbox(s[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = s[j];
        }
        x[4] = sbox(s[4]);
        if (!full) x[4] = mul_c(x[4], tab + Lay::INT_G + (k - RF) * NL);
        // all five lanes in every round, the last one included: skipping unused lanes there would save 0.3 % and
        // cost a branch per lane (the products must stay in one basic block with their operands' sign extensions,
        // or instruction selection falls back to 64 x 64-bit multiplies)
#pragma unroll
        for (int i = 0; i < WIDTH; ++i) s[i] = int_row(x, tab + Lay::INT_N + i, tab + Lay::INT_KAPPA + (k * WIDTH + i) * NL);
    }
#pragma unroll
    for (int i = 0; i < WIDTH; ++i)
        if ((OUT_ROWS >> i) & 1u) s[i] = mul_c(s[i], tab + Lay::INT_F);
}

}  // namespace p252
