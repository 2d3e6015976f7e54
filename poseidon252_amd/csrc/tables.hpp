// tables.hpp — derives the constant tables the kernels consume from the reference's two asset files.
//
// Input: the bytes of assets/arc.bin (340 x 32 B) and assets/mds.bin (25 x 32 B), read exactly as
// src/hades/round_constants.rs:26-54 and src/hades/mds_matrix.rs:17-39 read them
// (u64_from_buffer little-endian, BlsScalar::from_raw => field value = that integer mod p).
//
// Output: the constants of an algebraically equivalent schedule of Hades::perm
// (src/hades/permutation.rs:105-123) — same field elements out, hence bit-identical limbs:
//   * round constants of the 60 partial rounds pushed forward through the linear layer so that
//     each partial round adds ONE constant (to lane 4, before its S-box);
//   * the MDS matrix of the partial rounds factored M = M'' * M' (M' commutes with the lane-4
//     S-box, M'' is identity except row 4 / column 4): 9 multiplications per partial round
//     instead of 25, plus one dense pre-matrix merged into full round 3.
// Field-multiplication count per permutation: 8*40 + 60*12 = 1040 (reference schedule: 2000).
// tests/pymodel.py holds an independent big-int derivation; tests/test_tables.py compares.
#pragma once
#include <cstdint>
#include <vector>

#include "fr29.hpp"
#include "fr_host.hpp"

namespace p252 {

constexpr int WIDTH = 5;            // src/hades.rs:34
constexpr int FULL_ROUNDS = 8;      // src/hades.rs:29
constexpr int PARTIAL_ROUNDS = 60;  // src/hades.rs:31
constexpr int ROUNDS = FULL_ROUNDS + PARTIAL_ROUNDS;

struct SparseRound {
    FrHost w[4];    // row 4 of M''_q, columns 0..3
    FrHost d;       // M''_q[4][4]
    FrHost b[4];    // column 4 of M''_q, rows 0..3
    FrHost add4;    // constant folded into lane 4 of this layer's output (next S-box input constant)
};

struct HadesTables {
    FrHost c_first[WIDTH];            // ARC of round 0
    FrHost full_add[FULL_ROUNDS][WIDTH];  // vector added after the matrix of full round f (f = 0..7)
    FrHost mds[WIDTH][WIDTH];         // M
    FrHost mds_pre[WIDTH][WIDTH];     // M'_1 * M, used by full round index 3
    SparseRound sparse[PARTIAL_ROUNDS];
    FrHost last_add[4];               // lanes 0..3 after the last sparse layer
    // ---- "ARMA" form of partial rounds 5..60 (see derive_tables) ----
    FrHost arma_a[4];                 // a_1..a_4: A^4 = a1 A^3 + a2 A^2 + a3 A + a4 I
    FrHost arma_beta[5];              // beta_0..beta_4
    FrHost arma_kappa[PARTIAL_ROUNDS - 4];  // kappa_6 .. kappa_61  (index q-6)
    FrHost exit_gy[4][4];             // L_61 = Gy (u_58..u_61) + Gv (v_57..v_60) + exit_add
    FrHost exit_gv[4][4];
    FrHost exit_add[4];
    // ---- direct entry into the ARMA phase: full round 3 outputs the projections p_q = c^T A^(q-1) L_1
    //      (rows 0..3) and u_1 (row 4); then u_{q+1} = p_q + sum_{n<q} g_n v_{q-n}  for q = 1..4 ----
    FrHost mds_entry[WIDTH][WIDTH];   // rows 0..3: (c^T A^(q-1)) * M[0..3][:], row 4: M[4][:]   (unscaled)
    FrHost entry_add[WIDTH];          // k_2, k_3, k_4, k_5, k_1                                  (unscaled)
    FrHost entry_g[4];                // Markov parameters g_0..g_3 (scaled by lam^4, see below)
    // ---- state re-scaling (x^5 is homogeneous: a diagonal scaling commutes with an S-box layer up to 5th
    //      powers).  After every linear layer the state is re-scaled so that ONE coefficient per output row
    //      equals tau = 2^-20, the value whose device encoding is exactly 2^261: that product becomes a
    //      plain addition.  arma_beta / arma_kappa / exit_* / entry_g above hold the SCALED values;
    //      sc_mats / sc_adds are the per-round matrices and additive constants of the 8 full rounds
    //      (index 3 = the entry matrix).  Column 0 of rounds 0,1,2,4,5,6 is tau. ----
    FrHost sc_mats[FULL_ROUNDS][WIDTH][WIDTH];
    FrHost sc_adds[FULL_ROUNDS][WIDTH];
    FrHost lam;                       // time-invariant scale of the partial phase: beta_3 * lam^4 == tau
    // ---- integer-MDS schedule (derive_tables step 5): residues exactly as the device holds them ----
    FrHost int_kappa[ROUNDS][WIDTH];  // added (at the low end of the row accumulator) by the linear layer of round k
    FrHost int_g[PARTIAL_ROUNDS];     // G_k: equalises the scale of the lane-4 S-box output in partial round k
    FrHost int_f;                     // F: restores the Montgomery scale R after round 67
    bool int_ok;                      // mds.bin really is R/(i+j+5) (the structure this schedule rests on)
};

// The MDS matrix is R/(i+j+5) (mds_matrix.rs:21-36 reads Montgomery words of 1/(i+j+5) with from_raw):
// INT_L/(i+j+5) is an integer below 2^17 for every entry.
constexpr int32_t INT_L = 360360;  // lcm(5..13)

inline uint64_t u64_from_buffer(const unsigned char* buf, size_t i) {  // src/hades.rs:40-51
    uint64_t v = 0;
    for (int k = 7; k >= 0; --k) v = (v << 8) | buf[i + k];
    return v;
}

inline void mat4_inverse(const FrHost A[4][4], FrHost out[4][4]) {
    FrHost a[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            a[i][j] = A[i][j];
            a[i][4 + j] = (i == j) ? FrHost::one() : FrHost::zero();
        }
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        while (piv < 4 && a[piv][c].is_zero()) ++piv;  // MDS sub-blocks are invertible
        if (piv != c)
            for (int j = 0; j < 8; ++j) {
                FrHost tmp = a[c][j];
                a[c][j] = a[piv][j];
                a[piv][j] = tmp;
            }
        FrHost inv = a[c][c].inv();
        for (int j = 0; j < 8; ++j) a[c][j] = a[c][j] * inv;
        for (int r = 0; r < 4; ++r) {
            if (r == c || a[r][c].is_zero()) continue;
            FrHost f = a[r][c];
            for (int j = 0; j < 8; ++j) a[r][j] = a[r][j] - f * a[c][j];
        }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) out[i][j] = a[i][4 + j];
}

inline void derive_tables(const unsigned char* arc_bin, const unsigned char* mds_bin, HadesTables& T) {
    FrHost C[ROUNDS][WIDTH];
    for (int j = 0; j < ROUNDS * WIDTH; ++j) {  // round_constants.rs:40-51
        uint64_t raw[4];
        for (int k = 0; k < 4; ++k) raw[k] = u64_from_buffer(arc_bin, (size_t)j * 32 + 8 * k);
        C[j / WIDTH][j % WIDTH] = FrHost::from_raw(raw);
    }
    FrHost (&M)[WIDTH][WIDTH] = T.mds;
    for (int i = 0, k = 0; i < WIDTH; ++i)  // mds_matrix.rs:24-36, row-major
        for (int j = 0; j < WIDTH; ++j, k += 32) {
            uint64_t raw[4];
            for (int q = 0; q < 4; ++q) raw[q] = u64_from_buffer(mds_bin, k + 8 * q);
            M[i][j] = FrHost::from_raw(raw);
        }
    const int RF = FULL_ROUNDS / 2;
    // ---- (1) forward-push the partial-round constants ----
    FrHost k_const[PARTIAL_ROUNDS];
    FrHost delta[WIDTH];
    for (int i = 0; i < WIDTH; ++i) delta[i] = FrHost::zero();
    for (int q = 0; q < PARTIAL_ROUNDS; ++q) {
        FrHost a[WIDTH];
        for (int i = 0; i < WIDTH; ++i) a[i] = delta[i] + C[RF + q][i];
        k_const[q] = a[4];
        for (int r = 0; r < WIDTH; ++r) {  // delta = M * (a with lane 4 zeroed)
            FrHost s = FrHost::zero();
            for (int j = 0; j < 4; ++j) s = s + M[r][j] * a[j];
            delta[r] = s;
        }
    }
    FrHost closing_first[WIDTH];
    for (int i = 0; i < WIDTH; ++i) closing_first[i] = C[RF + PARTIAL_ROUNDS][i] + delta[i];
    // ---- (2) sparse factorisation, last partial round first ----
    FrHost Mk[WIDTH][WIDTH];
    for (int i = 0; i < WIDTH; ++i)
        for (int j = 0; j < WIDTH; ++j) Mk[i][j] = M[i][j];
    for (int q = PARTIAL_ROUNDS - 1; q >= 0; --q) {
        FrHost A[4][4], Ainv[4][4];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) A[i][j] = Mk[i][j];
        mat4_inverse(A, Ainv);
        SparseRound& S = T.sparse[q];
        for (int j = 0; j < 4; ++j) {  // w = c^T A^{-1}
            FrHost s = FrHost::zero();
            for (int i = 0; i < 4; ++i) s = s + Mk[4][i] * Ainv[i][j];
            S.w[j] = s;
        }
        S.d = Mk[4][4];
        for (int i = 0; i < 4; ++i) S.b[i] = Mk[i][4];
        S.add4 = FrHost::zero();
        // Mk <- M' * M with M' = blockdiag(A, 1)
        FrHost next[WIDTH][WIDTH];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < WIDTH; ++j) {
                FrHost s = FrHost::zero();
                for (int k = 0; k < 4; ++k) s = s + A[i][k] * M[k][j];
                next[i][j] = s;
            }
        for (int j = 0; j < WIDTH; ++j) next[4][j] = M[4][j];
        for (int i = 0; i < WIDTH; ++i)
            for (int j = 0; j < WIDTH; ++j) Mk[i][j] = next[i][j];
    }
    for (int i = 0; i < WIDTH; ++i)
        for (int j = 0; j < WIDTH; ++j) T.mds_pre[i][j] = Mk[i][j];
    for (int q = 0; q + 1 < PARTIAL_ROUNDS; ++q) T.sparse[q].add4 = k_const[q + 1];
    T.sparse[PARTIAL_ROUNDS - 1].add4 = closing_first[4];
    for (int i = 0; i < 4; ++i) T.last_add[i] = closing_first[i];
    // ---- additive constants of the full rounds ----
    for (int i = 0; i < WIDTH; ++i) T.c_first[i] = C[0][i];
    for (int f = 0; f < FULL_ROUNDS; ++f)
        for (int i = 0; i < WIDTH; ++i) T.full_add[f][i] = FrHost::zero();
    for (int f = 0; f < RF - 1; ++f)  // after full round f comes ARC of round f+1
        for (int i = 0; i < WIDTH; ++i) T.full_add[f][i] = C[f + 1][i];
    T.full_add[RF - 1][4] = k_const[0];  // after the pre-matrix: only the first partial S-box constant
    for (int f = RF; f < FULL_ROUNDS - 1; ++f)  // closing rounds: round index f + PARTIAL
        for (int i = 0; i < WIDTH; ++i) T.full_add[f][i] = C[f + PARTIAL_ROUNDS + 1][i];

    // ---- (3) ARMA form.  With M = [[A, b], [c^T, d]] the partial rounds are the 4th-order LTI system
    //   L_{q+1} = A L_q + b v_q,   u_{q+1} = c^T L_q + d v_q + k_{q+1}   (u = S-box input, v = u^5).
    // Cayley-Hamilton on A eliminates L:  u_{q+1} = sum a_m u_{q+1-m} + sum beta_n v_{q-n} + kappa_{q+1}
    // for q >= 5: 9 multiplications and ONE reduction per partial round.  Rounds 1..4 run in the sparse
    // form (they create the history); the state lanes 0..3 are recovered after round 60 through the
    // observability matrix.  tests/pymodel.py::derive_arma is the independent big-int derivation.
    FrHost A[4][4], bvec[4], cvec[4];
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 4; ++j) A[i][j] = M[i][j];
        bvec[i] = M[i][4];
        cvec[i] = M[4][i];
    }
    const FrHost dval = M[4][4];
    auto mat4mul = [](const FrHost X[4][4], const FrHost Y[4][4], FrHost Z[4][4]) {
        FrHost tmp[4][4];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                FrHost acc = FrHost::zero();
                for (int k = 0; k < 4; ++k) acc = acc + X[i][k] * Y[k][j];
                tmp[i][j] = acc;
            }
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) Z[i][j] = tmp[i][j];
    };
    {  // characteristic polynomial by Faddeev-LeVerrier
        FrHost Mk4[4][4];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) Mk4[i][j] = (i == j) ? FrHost::one() : FrHost::zero();
        for (int k = 1; k <= 4; ++k) {
            FrHost AM[4][4];
            mat4mul(A, Mk4, AM);
            FrHost tr = AM[0][0] + AM[1][1] + AM[2][2] + AM[3][3];
            FrHost ck = (tr * FrHost::from_u64((uint64_t)k).inv()).neg();  // coefficient of x^(4-k)
            T.arma_a[k - 1] = ck.neg();
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) Mk4[i][j] = (i == j) ? AM[i][j] + ck : AM[i][j];
        }
    }
    FrHost powA[5][4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) powA[0][i][j] = (i == j) ? FrHost::one() : FrHost::zero();
    for (int e = 1; e <= 4; ++e) mat4mul(powA[e - 1], A, powA[e]);
    FrHost g[5];  // Markov parameters g_0 = d, g_i = c^T A^(i-1) b
    g[0] = dval;
    for (int e = 1; e <= 4; ++e) {
        FrHost acc = FrHost::zero();
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) acc = acc + cvec[i] * powA[e - 1][i][j] * bvec[j];
        g[e] = acc;
    }
    for (int n = 0; n < 5; ++n) {
        FrHost acc = g[n];
        for (int m = 1; m <= 4 && m <= n; ++m) acc = acc - T.arma_a[m - 1] * g[n - m];
        T.arma_beta[n] = acc;
    }
    FrHost kq[PARTIAL_ROUNDS + 2];  // k_1..k_60, k_61 := closing constant of lane 4
    for (int q = 1; q <= PARTIAL_ROUNDS; ++q) kq[q] = k_const[q - 1];
    kq[PARTIAL_ROUNDS + 1] = closing_first[4];
    for (int q = 6; q <= PARTIAL_ROUNDS + 1; ++q) {
        FrHost acc = kq[q];
        for (int m = 1; m <= 4; ++m) acc = acc - T.arma_a[m - 1] * kq[q - m];
        T.arma_kappa[q - 6] = acc;
    }
    {  // exit matrices
        FrHost O[4][4], Oinv[4][4], A4Oinv[4][4], Toep[4][4], A4OinvT[4][4];
        for (int r = 0; r < 4; ++r)
            for (int j = 0; j < 4; ++j) {
                FrHost acc = FrHost::zero();
                for (int i = 0; i < 4; ++i) acc = acc + cvec[i] * powA[r][i][j];
                O[r][j] = acc;  // row r = c^T A^r
            }
        mat4_inverse(O, Oinv);
        mat4mul(powA[4], Oinv, A4Oinv);
        for (int r = 0; r < 4; ++r)
            for (int c2 = 0; c2 < 4; ++c2) Toep[r][c2] = (c2 <= r) ? g[r - c2] : FrHost::zero();
        mat4mul(A4Oinv, Toep, A4OinvT);
        for (int i = 0; i < 4; ++i)
            for (int s2 = 0; s2 < 4; ++s2) {
                FrHost kb = FrHost::zero();  // (A^(3-s) b)[i]
                for (int j = 0; j < 4; ++j) kb = kb + powA[3 - s2][i][j] * bvec[j];
                T.exit_gv[i][s2] = kb - A4OinvT[i][s2];
                T.exit_gy[i][s2] = A4Oinv[i][s2];
            }
        for (int i = 0; i < 4; ++i) {
            FrHost acc = closing_first[i];
            for (int r = 0; r < 4; ++r) acc = acc - T.exit_gy[i][r] * kq[58 + r];
            T.exit_add[i] = acc;
        }
    }
    // direct entry: row q-1 of mds_entry = (c^T A^(q-1)) * M[0..3][:]  (q = 1..4), row 4 = M[4][:]
    for (int q = 1; q <= 4; ++q)
        for (int j = 0; j < WIDTH; ++j) {
            FrHost acc = FrHost::zero();
            for (int i = 0; i < 4; ++i) {
                FrHost pi = FrHost::zero();  // (c^T A^(q-1))[i]
                for (int r = 0; r < 4; ++r) pi = pi + cvec[r] * powA[q - 1][r][i];
                acc = acc + pi * M[i][j];
            }
            T.mds_entry[q - 1][j] = acc;
        }
    for (int j = 0; j < WIDTH; ++j) T.mds_entry[4][j] = M[4][j];
    for (int q = 1; q <= 4; ++q) T.entry_add[q - 1] = kq[q + 1];
    T.entry_add[4] = kq[1];
    for (int n = 0; n < 4; ++n) T.entry_g[n] = g[n];

    // ---- (4) state re-scaling (see HadesTables).  tests/pymodel.py::derive_scaled is the big-int twin. ----
    const FrHost tau = FrHost::pow2(20).inv();
    const FrHost tau_inv = FrHost::pow2(20);
    T.lam = (tau * T.arma_beta[3].inv()).fourth_root();  // exists: checked in tests (and asserted below by use)
    const FrHost lam = T.lam, lam_inv = lam.inv();
    const FrHost lam4 = (lam * lam) * (lam * lam), lam5 = lam4 * lam;
    FrHost L[WIDTH];
    for (int i = 0; i < WIDTH; ++i) L[i] = FrHost::one();
    for (int r = 0; r < RF - 1; ++r) {  // opening rounds 0..2: column 0 normalised
        FrHost L5[WIDTH], Ln[WIDTH];
        for (int j = 0; j < WIDTH; ++j) L5[j] = L[j].pow5();
        for (int i = 0; i < WIDTH; ++i) Ln[i] = M[i][0] * L5[0] * tau_inv;
        for (int i = 0; i < WIDTH; ++i) {
            const FrHost li = Ln[i].inv();
            for (int j = 0; j < WIDTH; ++j) T.sc_mats[r][i][j] = li * M[i][j] * L5[j];
            T.sc_adds[r][i] = li * C[r + 1][i];
        }
        for (int i = 0; i < WIDTH; ++i) L[i] = Ln[i];
    }
    {  // round 3 = entry matrix, scaled by 1/lam (its outputs are the partial phase's scaled projections and u_1)
        FrHost L5[WIDTH];
        for (int j = 0; j < WIDTH; ++j) L5[j] = L[j].pow5();
        for (int i = 0; i < WIDTH; ++i) {
            for (int j = 0; j < WIDTH; ++j) T.sc_mats[RF - 1][i][j] = lam_inv * T.mds_entry[i][j] * L5[j];
            T.sc_adds[RF - 1][i] = lam_inv * T.entry_add[i];
        }
    }
    for (int n = 0; n < 4; ++n) T.entry_g[n] = g[n] * lam4;
    for (int n = 0; n < 5; ++n) T.arma_beta[n] = T.arma_beta[n] * lam4;  // arma_beta[3] == tau now
    for (int q = 0; q < PARTIAL_ROUNDS - 4; ++q) T.arma_kappa[q] = T.arma_kappa[q] * lam_inv;
    FrHost Lc[WIDTH];  // scale of the state leaving the exit: rows 0..3 normalise the coefficient of v_60
    for (int i = 0; i < 4; ++i) Lc[i] = T.exit_gv[i][3] * lam5 * tau_inv;
    Lc[4] = lam;
    for (int i = 0; i < 4; ++i) {
        const FrHost li = Lc[i].inv();
        for (int r = 0; r < 4; ++r) {
            T.exit_gy[i][r] = li * T.exit_gy[i][r] * lam;
            T.exit_gv[i][r] = li * T.exit_gv[i][r] * lam5;
        }
        T.exit_add[i] = li * T.exit_add[i];
    }
    for (int i = 0; i < WIDTH; ++i) L[i] = Lc[i];
    for (int f = RF; f < FULL_ROUNDS; ++f) {  // closing rounds: 4,5,6 normalised, 7 outputs the true state
        const bool last = f == FULL_ROUNDS - 1;
        FrHost L5[WIDTH], Ln[WIDTH];
        for (int j = 0; j < WIDTH; ++j) L5[j] = L[j].pow5();
        for (int i = 0; i < WIDTH; ++i) Ln[i] = last ? FrHost::one() : M[i][0] * L5[0] * tau_inv;
        for (int i = 0; i < WIDTH; ++i) {
            const FrHost li = Ln[i].inv();
            for (int j = 0; j < WIDTH; ++j) T.sc_mats[f][i][j] = li * M[i][j] * L5[j];
            T.sc_adds[f][i] = last ? FrHost::zero() : li * C[f + PARTIAL_ROUNDS + 1][i];
        }
        for (int i = 0; i < WIDTH; ++i) L[i] = Ln[i];
    }

    // ---- (5) integer-MDS schedule.  M = (R/L) N with N[i][j] = L/(i+j+5) a one-digit integer, so the linear
    // layer costs 9 MACs per term; the field factor R/L lives in the scale s_k of the stored state
    // (stored = s_k * true S-box input, the same s_k on all five lanes), which x -> x^5 turns into s_k^5:
    //   full round     X_j = Z_j^5 / R'^4                                   scale e = s^5 / R'^4
    //   partial round  X_j = Z_j (j<4), X_4 = Z_4^5/R'^4 * G_k/R'           G_k = R'^5 / s^4  =>  e = s
    //   linear layer   Z'_i = (sum_j N_ij X_j + kappa_i) / 2^29             s' = e L / (R 2^29),
    //                                                                       kappa_i = 2^29 s' C_{k+1}[i]
    //   after round 67 out_i R = Z'_i F / R',  F = R R' / s_68.     tests/pymodel.py::derive_int is the twin.
    {
        T.int_ok = true;
        for (int i = 0; i < WIDTH; ++i)
            for (int j = 0; j < WIDTH; ++j)  // a different mds.bin must not pass silently
                if (!(M[i][j] * FrHost::from_u64((uint64_t)(i + j + 5)) == FrHost::pow2(256))) T.int_ok = false;
        const FrHost RP = FrHost::pow2(261), RM = FrHost::pow2(256), T29 = FrHost::pow2(29);
        const FrHost RP4inv = ((RP * RP) * (RP * RP)).inv(), RP5 = (RP * RP) * (RP * RP) * RP;
        const FrHost step = FrHost::from_u64((uint64_t)INT_L) * RM.inv() * T29.inv();
        FrHost s = RM;
        for (int k = 0; k < ROUNDS; ++k) {
            const bool full = k < RF || k >= RF + PARTIAL_ROUNDS;
            FrHost e;
            if (full) {
                e = s.pow5() * RP4inv;
            } else {
                T.int_g[k - RF] = ((s * s) * (s * s)).inv() * RP5;
                e = s;
            }
            s = e * step;
            for (int i = 0; i < WIDTH; ++i) T.int_kappa[k][i] = k + 1 < ROUNDS ? T29 * s * C[k + 1][i] : FrHost::zero();
        }
        T.int_f = RM * RP * s.inv();
    }
}

// =============================================================================================
// Device encoding for the 29-bit-limb kernels (fr29.hpp)
// =============================================================================================
// Flat int32 table, 9 digits per constant, balanced digits in [-2^28, 2^28] of the CENTRED integer
// representative (|n| <= p/2):
//   additive constants ("A"):  n = a * 2^256            (same form as the state)
//   multipliers of plain state lanes ("MP"): n = c * 2^261          (redc divides by 2^261)
//   multipliers of S-box outputs ("MS"):     n = c * 2^261 * 2^20   (S-box output carries 2^-20)
struct Tab29Layout {
    static constexpr int C_FIRST = 0;                                   // [5][9]   A
    static constexpr int FULL_ADD = C_FIRST + WIDTH * NL;               // [8][5][9] A
    static constexpr int MDS = FULL_ADD + FULL_ROUNDS * WIDTH * NL;     // [5][5][9] MS
    static constexpr int MDS_PRE = MDS + WIDTH * WIDTH * NL;            // [5][5][9] MS
    static constexpr int SPARSE = MDS_PRE + WIDTH * WIDTH * NL;         // [60][SPARSE_STRIDE]
    //   per sparse round: w[4][9] MP | d[9] MS | b[4][9] MS | add4[9] A
    static constexpr int SP_W = 0, SP_D = 4 * NL, SP_B = 5 * NL, SP_ADD4 = 9 * NL;
    static constexpr int SPARSE_STRIDE = 10 * NL;
    static constexpr int LAST_ADD = SPARSE + PARTIAL_ROUNDS * SPARSE_STRIDE;  // [4][9] A
    static constexpr int ARMA_A = LAST_ADD + 4 * NL;                          // [4][9] MP  (a_1..a_4)
    static constexpr int ARMA_BETA = ARMA_A + 4 * NL;                         // [5][9] MS  (beta_0..beta_4)
    static constexpr int ARMA_KAPPA = ARMA_BETA + 5 * NL;                     // [56][9] A  (kappa_6..kappa_61)
    static constexpr int EXIT_GY = ARMA_KAPPA + (PARTIAL_ROUNDS - 4) * NL;    // [4][4][9] MP
    static constexpr int EXIT_GV = EXIT_GY + 16 * NL;                         // [4][4][9] MS
    static constexpr int EXIT_ADD = EXIT_GV + 16 * NL;                        // [4][9] A
    static constexpr int SC_MATS = EXIT_ADD + 4 * NL;                         // [8][5][5][9] MS  per-round matrices
    static constexpr int SC_ADDS = SC_MATS + FULL_ROUNDS * WIDTH * WIDTH * NL;  // [8][5][9] A
    static constexpr int ENTRY_G = SC_ADDS + FULL_ROUNDS * WIDTH * NL;        // [4][9] MS  (g_0..g_3, scaled)
    // integer-MDS schedule: raw residues (digits encode the value itself)
    static constexpr int INT_N = ENTRY_G + 4 * NL;                            // [9] one int each: N[i][j] = h[i+j], h[d] = L/(d+5)
    static constexpr int INT_KAPPA = INT_N + NL;                              // [68][5][9]
    static constexpr int INT_G = INT_KAPPA + ROUNDS * WIDTH * NL;             // [60][9]
    static constexpr int INT_F = INT_G + PARTIAL_ROUNDS * NL;                 // [9]
    static constexpr int TOTAL = INT_F + NL;
};

inline void encode_balanced29(const FrHost& field_value, int32_t out[NL]) {
    uint64_t n[4];
    field_value.to_canonical(n);
    // centre: if n > (p-1)/2 use n - p (negative)
    static const uint64_t HALF[4] = {0x7fffffff80000000ULL, 0xa9ded2017fff2dffULL, 0x199cec0404d0ec02ULL,
                                     0x39f6d3a994cebea4ULL};  // (p-1)/2
    bool neg = false;
    for (int i = 3; i >= 0; --i) {
        if (n[i] > HALF[i]) { neg = true; break; }
        if (n[i] < HALF[i]) break;
    }
    if (neg) {  // n <- p - n
        u128_t borrow = 0;
        for (int i = 0; i < 4; ++i) {
            u128_t d = (u128_t)FrHost::P[i] - n[i] - borrow;
            n[i] = (uint64_t)d;
            borrow = (d >> 64) & 1;
        }
    }
    int64_t dig[NL];
    for (int i = 0; i < NL; ++i) {
        int bit = WB * i, q = bit >> 6, s = bit & 63;
        u128_t window = n[q];
        if (q + 1 < 4) window |= (u128_t)n[q + 1] << 64;
        dig[i] = (int64_t)((uint64_t)(window >> s) & DMASK);
    }
    for (int i = 0; i < NL - 1; ++i)
        if (dig[i] > (1 << (WB - 1))) {
            dig[i] -= (int64_t)1 << WB;
            dig[i + 1] += 1;
        }
    for (int i = 0; i < NL; ++i) out[i] = (int32_t)(neg ? -dig[i] : dig[i]);
}

inline std::vector<int32_t> encode_tables29(const HadesTables& T) {
    typedef Tab29Layout Lay;
    std::vector<int32_t> tab(Lay::TOTAL, 0);
    const FrHost fA = FrHost::pow2(256);
    const FrHost fMP = FrHost::pow2(261);
    const FrHost fMS = FrHost::pow2(261 + 20);
    auto put = [&](int off, const FrHost& v, const FrHost& scale) { encode_balanced29(v * scale, &tab[off]); };
    for (int i = 0; i < WIDTH; ++i) put(Lay::C_FIRST + i * NL, T.c_first[i], fA);
    for (int f = 0; f < FULL_ROUNDS; ++f)
        for (int i = 0; i < WIDTH; ++i) put(Lay::FULL_ADD + (f * WIDTH + i) * NL, T.full_add[f][i], fA);
    for (int i = 0; i < WIDTH; ++i)
        for (int j = 0; j < WIDTH; ++j) {
            put(Lay::MDS + (i * WIDTH + j) * NL, T.mds[i][j], fMS);
            put(Lay::MDS_PRE + (i * WIDTH + j) * NL, T.mds_pre[i][j], fMS);
        }
    for (int q = 0; q < PARTIAL_ROUNDS; ++q) {
        const int base = Lay::SPARSE + q * Lay::SPARSE_STRIDE;
        for (int j = 0; j < 4; ++j) put(base + Lay::SP_W + j * NL, T.sparse[q].w[j], fMP);
        put(base + Lay::SP_D, T.sparse[q].d, fMS);
        for (int i = 0; i < 4; ++i) put(base + Lay::SP_B + i * NL, T.sparse[q].b[i], fMS);
        put(base + Lay::SP_ADD4, T.sparse[q].add4, fA);
    }
    for (int i = 0; i < 4; ++i) put(Lay::LAST_ADD + i * NL, T.last_add[i], fA);
    for (int m = 0; m < 4; ++m) put(Lay::ARMA_A + m * NL, T.arma_a[m], fMP);
    for (int n = 0; n < 5; ++n) put(Lay::ARMA_BETA + n * NL, T.arma_beta[n], fMS);
    for (int q = 0; q < PARTIAL_ROUNDS - 4; ++q) put(Lay::ARMA_KAPPA + q * NL, T.arma_kappa[q], fA);
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 4; ++r) {
            put(Lay::EXIT_GY + (i * 4 + r) * NL, T.exit_gy[i][r], fMP);
            put(Lay::EXIT_GV + (i * 4 + r) * NL, T.exit_gv[i][r], fMS);
        }
    for (int i = 0; i < 4; ++i) put(Lay::EXIT_ADD + i * NL, T.exit_add[i], fA);
    for (int f = 0; f < FULL_ROUNDS; ++f)
        for (int i = 0; i < WIDTH; ++i) {
            for (int j = 0; j < WIDTH; ++j) put(Lay::SC_MATS + ((f * WIDTH + i) * WIDTH + j) * NL, T.sc_mats[f][i][j], fMS);
            put(Lay::SC_ADDS + (f * WIDTH + i) * NL, T.sc_adds[f][i], fA);
        }
    for (int n = 0; n < 4; ++n) put(Lay::ENTRY_G + n * NL, T.entry_g[n], fMS);
    const FrHost one = FrHost::one();
    for (int d = 0; d < 2 * WIDTH - 1; ++d) tab[Lay::INT_N + d] = INT_L / (d + 5);
    for (int k = 0; k < ROUNDS; ++k)
        for (int i = 0; i < WIDTH; ++i) put(Lay::INT_KAPPA + (k * WIDTH + i) * NL, T.int_kappa[k][i], one);
    for (int q = 0; q < PARTIAL_ROUNDS; ++q) put(Lay::INT_G + q * NL, T.int_g[q], one);
    put(Lay::INT_F, T.int_f, one);
    return tab;
}

// Worst-case |column| (as a double) over every lazy accumulation the schedules perform, for the
// ACTUAL constants in `tab`: state digits are bounded by 2^29 (top digit by 2^25: |V| < 4p), the
// high-column initialisation by 2^30, and the reduction adds at most 2^29 * sum(p digits) + carries.
// The kernels are correct iff this stays below 2^63.
inline double max_column_bound29(const int32_t* tab) {
    typedef Tab29Layout Lay;
    const double DIG = 536870912.0 /* 2^29 */, TOP = 33554432.0 /* 2^25 */;
    const double P_SUM = (double)P252_P29_1 + P252_P29_2 + P252_P29_3 + P252_P29_4 + P252_P29_5 + P252_P29_6 +
                         P252_P29_7 + P252_P29_8;
    const double REDC = DIG * P_SUM + 68719476736.0 /* carries < 2^36 */ + 1073741824.0 /* hi init 2^30 */;
    double worst = 0;
    auto group = [&](std::initializer_list<int> offsets) {
        double col[2 * NL] = {0};
        for (int off : offsets)
            for (int j = 0; j < NL; ++j) {
                const double cj = tab[off + j] < 0 ? -(double)tab[off + j] : (double)tab[off + j];
                for (int i = 0; i < NL; ++i) col[i + j] += (i == NL - 1 ? TOP : DIG) * cj;
            }
        for (int k = 0; k < 2 * NL; ++k)
            if (col[k] + REDC > worst) worst = col[k] + REDC;
    };
    group({Lay::ENTRY_G, Lay::ENTRY_G + NL, Lay::ENTRY_G + 2 * NL, Lay::ENTRY_G + 3 * NL});
    for (int base : {Lay::MDS, Lay::MDS_PRE, Lay::SC_MATS, Lay::SC_MATS + 225 * 1, Lay::SC_MATS + 225 * 2, Lay::SC_MATS + 225 * 3,
                     Lay::SC_MATS + 225 * 4, Lay::SC_MATS + 225 * 5, Lay::SC_MATS + 225 * 6, Lay::SC_MATS + 225 * 7})
        for (int k = 0; k < WIDTH; ++k)
            group({base + (k * 5 + 0) * NL, base + (k * 5 + 1) * NL, base + (k * 5 + 2) * NL, base + (k * 5 + 3) * NL,
                   base + (k * 5 + 4) * NL});
    for (int q = 0; q < PARTIAL_ROUNDS; ++q) {
        const int b = Lay::SPARSE + q * Lay::SPARSE_STRIDE;
        group({b + Lay::SP_W, b + Lay::SP_W + NL, b + Lay::SP_W + 2 * NL, b + Lay::SP_W + 3 * NL, b + Lay::SP_D});
        for (int i = 0; i < 4; ++i) group({b + Lay::SP_B + i * NL});
    }
    group({Lay::ARMA_A, Lay::ARMA_A + NL, Lay::ARMA_A + 2 * NL, Lay::ARMA_A + 3 * NL, Lay::ARMA_BETA,
           Lay::ARMA_BETA + NL, Lay::ARMA_BETA + 2 * NL, Lay::ARMA_BETA + 3 * NL, Lay::ARMA_BETA + 4 * NL});
    for (int i = 0; i < 4; ++i)
        group({Lay::EXIT_GY + (i * 4 + 0) * NL, Lay::EXIT_GY + (i * 4 + 1) * NL, Lay::EXIT_GY + (i * 4 + 2) * NL,
               Lay::EXIT_GY + (i * 4 + 3) * NL, Lay::EXIT_GV + (i * 4 + 0) * NL, Lay::EXIT_GV + (i * 4 + 1) * NL,
               Lay::EXIT_GV + (i * 4 + 2) * NL, Lay::EXIT_GV + (i * 4 + 3) * NL});
    for (int q = 0; q < PARTIAL_ROUNDS; ++q) group({Lay::INT_G + q * NL});
    group({Lay::INT_F});
    {  // integer rows: nine columns, five one-digit terms + kappa, then one digit step
        double nsum = 0;
        for (int j = 0; j < WIDTH; ++j) nsum += (double)tab[Lay::INT_N + j];  // row 0 is the largest
        const double row = DIG * nsum + 268435456.0 /* kappa digit */ + DIG * (double)P252_P29_1 + 68719476736.0;
        if (row > worst) worst = row;
    }
    // S-box: element x element (9 products of 2^29 x 2^29) and squarings (<= 4.5 * 2^59)
    const double sbox_col = 9.0 * DIG * DIG + REDC;
    if (sbox_col > worst) worst = sbox_col;
    return worst;
}

}  // namespace p252
