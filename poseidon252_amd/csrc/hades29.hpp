// hades29.hpp — the Hades permutation on E29 lazy residues (one state per lane).
//
// Replaces Hades::perm (src/hades/permutation.rs:105-123) with its ARC / S-box / MDS steps
// (src/hades/permutation/scalar.rs:39-64).  Three algebraically equivalent schedules, all producing
// the reference's field elements (tables.hpp derives their constants):
//   hades_permute_sparse  4 full | 60 sparse partial rounds (12 generic products, 8 reductions each) | 4 full
//   hades_permute_int     integer MDS in all 68 rounds: 25 one-digit products per linear layer, one generic
//                         product per partial round
//   hades_permute         integer MDS in the full rounds, integer ARMA recurrence in the partial rounds
//                         (nine one-digit products + one generic product per partial round)      <- kernels
// The first two are kept as independent cross-checks (tests/test_host_arith.py runs all three on the host).
//
// TP is any pointer-like giving int32 digits: tab[i].  In kernels it is a wave-uniform pointer so
// the compiler keeps constants in SGPRs (s_load) and feeds them to v_mad_i64_i32 as scalar operands.
#pragma once
#include "fr29.hpp"
#include "tables.hpp"

namespace p252 {

// One full round of the sparse schedule: state <- Mat * sbox(state) + add   (ARC of this round was folded
// into the previous layer's `add`).  5 S-boxes (15 mults, 15 redc) + 25 generic products + 5 redc.
template <class TP>
P252_HD void full_round(E29 s[WIDTH], TP mat, TP add) {
    E29 v[WIDTH];
#pragma unroll
    for (int i = 0; i < WIDTH; ++i) v[i] = sbox(s[i]);
#pragma unroll
    for (int k = 0; k < WIDTH; ++k) {
        A29 t;
        acc_set_hi_c(t, add + k * NL);
#pragma unroll
        for (int j = 0; j < WIDTH; ++j) acc_mul(t, v[j], mat + (k * WIDTH + j) * NL);  // MDS[k][j] * state[j]
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

// ---- schedule 2: integer MDS in all rounds (tables.hpp (B)) ----
// The MDS matrix is (R/L) * N with N[i][j] = L/(i+j+5) a one-digit integer, and x -> x^5 is homogeneous,
// so the field factor travels in the scale of the stored state and every round is
//     X_j = sbox(Z_j)  (all lanes / lane 4 only; in partial rounds lane 4 is then multiplied by G_k — the
//                       one generic product of the round — so that all five X_j carry the same scale)
//     Z'_i = (sum_j N_ij X_j + kappa_i) / 2^29          45 one-digit MACs + 1 Montgomery digit step per lane
// A full round is 15 + 0 generic products (reference: 15 + 25), a partial round 3 + 1 (reference: 3 + 25).
// One multiplication by F per output lane restores the reference's Montgomery scale at the end.
// n = &h[i] for output lane i (N is a Hankel matrix: N[i][j] = h[i+j], nine distinct values).
// LAZY: leave the digits un-carried (row_redc1_lazy) — for lanes that feed only the next integer layer.
template <bool LAZY = false, class TP>
P252_HD E29 int_row(const E29 x[WIDTH], TP n, TP kappa) {
    R29 t;
    row_set_c(t, kappa);
#pragma unroll
    for (int j = 0; j < WIDTH; ++j) row_mac(t, x[j], n[j]);
    return LAZY ? row_redc1_lazy(t) : row_redc1(t);
}

// All five lanes are computed in every round, the last one included: skipping the unused lanes of a digest
// there would save 0.3 % and cost a branch per lane, and the products must stay in ONE basic block with their
// operands' sign extensions (instruction selection works per block: a sign extension hoisted out of it turns
// v_mad_i64_i32 into a 64 x 64-bit multiply emulation).
template <unsigned OUT_ROWS = 0x1fu, class TP>
P252_HD void hades_permute_int(E29 s[WIDTH], TP tab) {
    typedef Tab29Layout Lay;
    constexpr int RF = FULL_ROUNDS / 2;
#pragma unroll
    for (int i = 0; i < WIDTH; ++i) add_c(s[i], tab + Lay::C_FIRST + i * NL);
#pragma unroll 1
    for (int k = 0; k < ROUNDS; ++k) {
        const TP kap = tab + Lay::INT_KAPPA + k * WIDTH * NL;
        E29 x[WIDTH];
        if (k < RF || k >= RF + PARTIAL_ROUNDS) {  // wave-uniform
            if (k == RF + PARTIAL_ROUNDS) {
#pragma unroll
                for (int j = 0; j < 4; ++j) normalize(s[j]);  // lanes 0..3 leave the partial rounds un-carried
            }
#pragma unroll
            for (int j = 0; j < WIDTH; ++j) x[j] = sbox(s[j]);
#pragma unroll
            for (int i = 0; i < WIDTH; ++i) s[i] = int_row(x, tab + Lay::INT_N + i, kap + i * NL);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = s[j];
            x[4] = mul_c(sbox(s[4]), tab + Lay::INT_G + (k - RF) * NL);
#pragma unroll
            for (int i = 0; i < 4; ++i) s[i] = int_row<true>(x, tab + Lay::INT_N + i, kap + i * NL);
            s[4] = int_row(x, tab + Lay::INT_N + 4, kap + 4 * NL);
        }
    }
#pragma unroll
    for (int i = 0; i < WIDTH; ++i)
        if ((OUT_ROWS >> i) & 1u) s[i] = mul_c(s[i], tab + Lay::INT_F);
}

// ---- schedule 3: integer MDS in the full rounds + integer ARMA recurrence in the partial rounds (tables.hpp (C)) ----
// History of the partial phase: rings of HIST entries, U_q at Us[q mod HIST] and W_q at Ws[q mod HIST]; with
// HIST rounds per loop iteration every ring position is a compile-time constant and nothing is ever moved.
// One step (q = index of the partial round, 1..60):  W_q = sbox(U_q) * G_q / R';
//   U_{q+1} = sum_m A_m U_{q+1-m} + sum_n B_n W_{q-n} + K_{q+1}   (mod p: fold_top)
// The stored values carry the geometric scale sigma_{q+1} = sigma_q D / R (tables.hpp), for which ALL nine coefficients
// are one-digit integers at the same weight: nine one-digit products into nine columns that start as K's digits, then
// the reduction from the top (fr29.hpp fold_top: 13 instructions + the carry chain).  Round 2's recurrence divided by 2^29
// per round of age instead (terms of age j one digit lower, five Montgomery digit steps = 50 instructions).
// ab = A_1..A_4, B_0..B_4 (ints); kg = K_{q+1}[9], G_q[9].  U_{q+1} overwrites U_{q-4}, W_q overwrites W_{q-5}.
constexpr int HIST = 5;
template <int QM /* q mod HIST */, class TP>
P252_HD void ai_recur(E29 Us[HIST], const E29 Ws[HIST], TP ab, TP kg, const RK& K) {  // U_{q+1} from the rings (W_q included)
    constexpr int Q = QM + HIST;  // keeps (Q - j) % HIST non-negative
    int64_t c[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) c[k] = kg[k];
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // A_{j+1} U_{q-j} and B_j W_{q-j}
        const int64_t aj = ab[j], bj = ab[4 + j];
#pragma unroll
        for (int k = 0; k < NL; ++k)
            c[k] += (int64_t)Us[(Q - j) % HIST].d[k] * aj + (int64_t)Ws[(Q - j) % HIST].d[k] * bj;
    }
    {
        const int64_t b4 = ab[8];
#pragma unroll
        for (int k = 0; k < NL; ++k) c[k] += (int64_t)Ws[(Q - 4) % HIST].d[k] * b4;
    }
    Us[(Q + 1) % HIST] = fold_top(c, K);
}
template <int QM /* q mod HIST */, class TP>
P252_HD void ai_round(E29 Us[HIST], E29 Ws[HIST], TP ab, TP kg, const RK& K) {
    Ws[QM] = mul_c_w<true>(sbox_w<true>(Us[QM], K), kg + NL, K);  // (wide digits: x^5 meets G, W_q the recurrence / exit rows)
    ai_recur<QM>(Us, Ws, ab, kg, K);
}

// Exit row i: lanes 0..3 of the state after round 60 from (U_58..U_61, W_57..W_60).  The coefficients are rationals
// with a common denominator per row: eight two-digit integer products, all at the same weight (the history's scale is
// geometric with the very ratio the coefficients' denominators have), two Montgomery digit steps, then ONE generic product
// by fix_i = 2^58 R' / den_i; add_i rides in the high columns.
// n = 8 x (lo, hi) digits: U_58..U_61 then W_57..W_60.
template <class TP>
P252_HD E29 exit_row(const E29* const u[4], const E29* const w[4], TP n, TP fix, TP add, const RK& K) {
    int64_t c[NL + 2];
#pragma unroll
    for (int k = 0; k < NL + 2; ++k) c[k] = k < 2 ? K.bias : 0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t ylo = n[2 * r], yhi = n[2 * r + 1], vlo = n[8 + 2 * r], vhi = n[8 + 2 * r + 1];
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            c[k] += (int64_t)u[r]->d[k] * ylo + (int64_t)w[r]->d[k] * vlo;
            c[k + 1] += (int64_t)u[r]->d[k] * yhi + (int64_t)w[r]->d[k] * vhi;
        }
    }
    P252_WSTEP(c, 0, NL + 2, K)
    P252_WSTEP(c, 1, NL + 2, K)
    E29 r;
    int64_t carry = 0;
#pragma unroll
    for (int k = 0; k < NL - 1; ++k) {
        const int64_t v = c[2 + k] + carry;
        r.d[k] = opaque_digit((int32_t)((uint32_t)v & DMASK));
        carry = v >> WB;
    }
    P252_TRK_COL(c[NL + 1] + carry);
    r.d[NL - 1] = opaque_digit((int32_t)(c[NL + 1] + carry));
    P252_TRK_TOP(r.d[NL - 1]);
    A29 t;
    acc_set_hi_c_w(t, add, K);
    acc_mul(t, r, fix);
    return redc_w(t, K);
}

// Entry row i (virtual history U_0, U_-1, U_-2): an integer combination of the S-box outputs 0..3 of full round 3
// (NDIG-digit coefficients, NDIG Montgomery digit steps), then ONE generic product by fix_i; add_i rides in the
// high columns.  n = 4 x (lo, hi) digits.
template <int NDIG, class TP>
P252_HD E29 entry_row(const E29 x[WIDTH], TP n, TP fix, TP add, const RK& K) {
    int64_t c[NL + NDIG];
#pragma unroll
    for (int k = 0; k < NL + NDIG; ++k) c[k] = k < NDIG ? K.bias : 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t lo = n[2 * j], hi = n[2 * j + 1];
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            c[k] += (int64_t)x[j].d[k] * lo;
            if (NDIG == 2) c[k + 1] += (int64_t)x[j].d[k] * hi;
        }
    }
    P252_WSTEP(c, 0, NL + NDIG, K)
    if (NDIG == 2) P252_WSTEP(c, 1, NL + NDIG, K)
    E29 r;
    int64_t carry = 0;
#pragma unroll
    for (int k = 0; k < NL - 1; ++k) {
        const int64_t v = c[NDIG + k] + carry;
        r.d[k] = opaque_digit((int32_t)((uint32_t)v & DMASK));
        carry = v >> WB;
    }
    P252_TRK_COL(c[NL + NDIG - 1] + carry);
    r.d[NL - 1] = opaque_digit((int32_t)(c[NL + NDIG - 1] + carry));
    P252_TRK_TOP(r.d[NL - 1]);
    A29 t;
    acc_set_hi_c_w(t, add, K);
    acc_mul(t, r, fix);
    return redc_w(t, K);
}

// x * m + add for a small integer m (W_0 = 28 X_4 + const): nine products, then the reduction from the top — the result is
// an ordinary lazy residue (below 1.2 p), which the aligned recurrence's quotient estimate assumes of every term.
template <class TP>
P252_HD E29 small_mul_add(const E29& x, int32_t m, TP add, const RK& K) {
    int64_t c[NL];
    const int64_t mm = m;
#pragma unroll
    for (int k = 0; k < NL; ++k) c[k] = (int64_t)x.d[k] * mm + (int64_t)add[k];
    return fold_top(c, K);
}

// The entry of the partial phase from the five S-box outputs x of full round 3: U_1 (the integer row of lane 4) and the
// virtual history U_0, U_-1, U_-2, W_0 (older W = 0), placed in the rings where round q = 1 expects them.
template <class TP>
P252_HD void arma_entry(const E29 x[WIDTH], E29 Us[HIST], E29 Ws[HIST], TP tab, const RK& K) {
    typedef Tab29Layout Lay;
    constexpr int RF = FULL_ROUNDS / 2;
    {  // U_1: lane 4's integer row of full round 3, reduced from the top instead of by a digit step (scale e L / R)
        int64_t c[NL];
#pragma unroll
        for (int k = 0; k < NL; ++k) c[k] = (tab + Lay::AI_KAPPA + ((RF - 1) * WIDTH + 4) * NL)[k];
#pragma unroll
        for (int j = 0; j < WIDTH; ++j) {
            const int64_t nj = (tab + Lay::INT_N + 4)[j];
#pragma unroll
            for (int k = 0; k < NL; ++k) c[k] += (int64_t)x[j].d[k] * nj;
        }
        Us[1] = fold_top(c, K);
    }
    sched_fence();
    // U_0, U_-1, U_-2: virtual
    Us[0] = entry_row<1>(x, tab + Lay::AI_ENT_N, tab + Lay::AI_ENT_FIX, tab + Lay::AI_ENT_ADD, K);
    sched_fence();
    Us[HIST - 1] = entry_row<1>(x, tab + Lay::AI_ENT_N + NL, tab + Lay::AI_ENT_FIX + NL, tab + Lay::AI_ENT_ADD + NL, K);
    sched_fence();
    Us[HIST - 2] = entry_row<2>(x, tab + Lay::AI_ENT_N + 2 * NL, tab + Lay::AI_ENT_FIX + 2 * NL, tab + Lay::AI_ENT_ADD + 2 * NL, K);
    sched_fence();
    Us[2] = e29_zero();  // (free slot)
    // W_0: virtual, = the lane-4 S-box output at the W scale (28 = 13 D / K); W_-1, W_-2, W_-3 = 0
    Ws[0] = small_mul_add(x[4], ENTRY_W0_INT, tab + Lay::AI_ENT_ADD + 3 * NL, K);
#pragma unroll
    for (int i = 1; i < HIST; ++i) Ws[i] = e29_zero();
}

// Loop nest: two halves, each = four full rounds (one copy of that body in the instruction stream), the first half
// followed by the partial phase: 12 iterations of HIST = 5 ARMA rounds in a loop of their own (its loop-carried
// values are exactly the two history rings), then the exit rows (exit_row).  Full round 3 is the entry: its linear
// layer produces U_1 and the virtual history (entry_row, small_mul_add).  OUT_ROWS: bit k set = lane k of the result is needed (a
// Merkle4 digest needs lane 1 only: the multiplication by F is done for that lane alone).
// PRE0: s[0] already holds lane 0's S-box OUTPUT of the first round (hades_pre0 below): the capacity lane of a digest is
// the tag, the same for every item of a batch, so that S-box is computed once on the host and travels as a kernel argument.
template <unsigned OUT_ROWS = 0x1fu, bool PRE0 = false, class TP>
P252_HD void hades_permute(E29 s[WIDTH], TP tab) {
    typedef Tab29Layout Lay;
    constexpr int RF = FULL_ROUNDS / 2;
    static_assert(PARTIAL_ROUNDS % HIST == 0, "60 ARMA rounds must split evenly");
    const RK K = make_rk();  // bias 2^31 / constant 8 of the wide Montgomery step (fr29.hpp), in registers throughout
#pragma unroll
    for (int i = PRE0 ? 1 : 0; i < WIDTH; ++i) add_c(s[i], tab + Lay::C_FIRST + i * NL);
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
#pragma unroll 1
        for (int f = half * RF; f < (half + 1) * RF; ++f) {
            const TP kap = tab + Lay::AI_KAPPA + f * WIDTH * NL;
            E29 x[WIDTH];
#pragma unroll
            for (int j = 1; j < WIDTH; ++j) x[j] = sbox_w<true>(s[j], K);  // (wide digits: they only meet the integer rows)
            if (PRE0 && f == 0)  // (wave-uniform) first round of a digest: lane 0's S-box output came with the launch
                x[0] = s[0];
            else
                x[0] = sbox_w<true>(s[0], K);
            if (f == RF - 1) {  // the linear layer of round 3 is the entry below: hand over the S-box outputs
#pragma unroll
                for (int i = 0; i < WIDTH; ++i) s[i] = x[i];
            } else if (OUT_ROWS != 0x1fu && f == 2 * RF - 1) {
                // last round of a digest: only the rows that are squeezed (their own block: one row of products each)
#pragma unroll
                for (int i = 0; i < WIDTH; ++i)
                    if ((OUT_ROWS >> i) & 1u) s[i] = int_row(x, tab + Lay::INT_N + i, kap + i * NL);
            } else {
#pragma unroll
                for (int i = 0; i < WIDTH; ++i) s[i] = int_row(x, tab + Lay::INT_N + i, kap + i * NL);
            }
        }
        if (half == 0) {
            // (the history rings are defined here, after the loop above, so that they are not carried through it)
            E29 Us[HIST], Ws[HIST];
            arma_entry(s, Us, Ws, tab, K);
#pragma unroll 1
            for (int it = 0; it < PARTIAL_ROUNDS / HIST; ++it) {  // rounds q = 5 it + 1 .. 5 it + 5
                const TP kg = tab + Lay::AI_KG + it * HIST * 2 * NL;
                ai_round<1>(Us, Ws, tab + Lay::AI_AB, kg, K);
                ai_round<2>(Us, Ws, tab + Lay::AI_AB, kg + 2 * NL, K);
                ai_round<3>(Us, Ws, tab + Lay::AI_AB, kg + 4 * NL, K);
                ai_round<4>(Us, Ws, tab + Lay::AI_AB, kg + 6 * NL, K);
                ai_round<0>(Us, Ws, tab + Lay::AI_AB, kg + 8 * NL, K);
            }
            // after round 60 (60 mod 5 = 0): U_58..U_61 at Us[3], Us[4], Us[0], Us[1]; W_57..W_60 at Ws[2], Ws[3], Ws[4], Ws[0]
            const E29* const us[4] = {&Us[3], &Us[4], &Us[0], &Us[1]};
            const E29* const ws[4] = {&Ws[2], &Ws[3], &Ws[4], &Ws[0]};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s[i] = exit_row(us, ws, tab + Lay::AI_EX_N + i * 2 * NL, tab + Lay::AI_EX_FIX + i * NL, tab + Lay::AI_EX_ADD + i * NL, K);
                sched_fence();
            }
            s[4] = Us[1];
        }
    }
#pragma unroll
    for (int i = 0; i < WIDTH; ++i)
        if ((OUT_ROWS >> i) & 1u) {
            s[i] = mul_c(s[i], tab + Lay::AI_F);
            sched_fence();
        }
}

// Lane 0's S-box output of the first round for the capacity element `tag` (host side, once per launch): exactly what
// hades_permute computes for lane 0 before its first linear layer.  Any representative of the residue class would do
// (every later step is exact modular arithmetic on lazy residues); this is the one the device itself would produce.
template <class TP>
P252_HD E29 hades_pre0(const E29& tag, TP tab) {
    typedef Tab29Layout Lay;
    const RK K = make_rk();
    E29 t = tag;
    add_c(t, tab + Lay::C_FIRST);
    return sbox_w(t, K);
}

}  // namespace p252
