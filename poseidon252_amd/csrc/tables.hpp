// tables.hpp — derives the constant tables the kernels consume from the reference's two asset files.
//
// Input: the bytes of assets/arc.bin (340 x 32 B) and assets/mds.bin (25 x 32 B), read exactly as
// src/hades/round_constants.rs:26-54 and src/hades/mds_matrix.rs:17-39 read them
// (u64_from_buffer little-endian, BlsScalar::from_raw => field value = that integer mod p).
//
// Output: the constants of three algebraically equivalent schedules of Hades::perm
// (src/hades/permutation.rs:105-123) — same field elements out, hence bit-identical limbs:
//   (A) "sparse": partial-round constants pushed forward, MDS of the partial rounds factored into sparse
//       matrices (9 generic products per partial round instead of 25).  Host cross-check only.
//   (B) "integer MDS": mds.bin is R/(i+j+5), i.e. (R/L) times the INTEGER matrix N[i][j] = L/(i+j+5) (L = 360360,
//       entries < 2^17); x -> x^5 is homogeneous, so the field factor travels in the scale of the stored
//       state and every linear layer is 25 one-digit products.  Host cross-check only.
//   (C) "integer ARMA": (B) in the full rounds; the 60 partial rounds as the 4th-order linear recurrence
//       u_{q+1} = sum a_m u_{q+1-m} + sum beta_n v_{q-n} + kappa_{q+1}, whose coefficients are one-digit
//       integers after geometric re-scaling.  This is what the kernels run.
// tests/pymodel.py holds independent big-int derivations of all three; tests/test_host_arith.py compares.
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

// mds.bin[i][j] = R/(i+j+5) (mds_matrix.rs:21-36 reads Montgomery words of 1/(i+j+5) with from_raw):
constexpr int32_t INT_L = 360360;  // lcm(5..13): INT_L/(i+j+5) is an integer below 2^17 for every entry
// The rational recurrence coefficients a~_m, b~_n of the Cauchy system become integers below 2^25 as
// A_m = a~_m D^m and B_n = b~_n D^n K  (tests/pymodel.py::_rational_arma derives them with exact fractions;
// derive_tables checks them against the field values computed from mds.bin).
constexpr int32_t INT_D = 27720, INT_K = 12870;
constexpr int32_t ARMA_A_INT[4] = {15104, -4729406, 18244864, -419265};
constexpr int32_t ARMA_B_INT[5] = {990, -1555121, 23296324, -2924911, 1694};
// Exit rows (lanes 0..3 after round 60 from U_58..U_61, W_57..W_60): D^(3-r) Gy~[i][r] and D^(3-t) K Gv~[i][t] are
// rationals; with the row-wise common denominator den_i they are integers below 2^47 (two digits).
constexpr int64_t EXIT_DEN_INT[4] = {114095520LL, 3422865600LL, 51342984000LL, 13446972000LL};
constexpr int64_t EXIT_NY_INT[4][4] = {{-8614218690LL, 374845978529LL, -96605623422LL, 196355311LL},
                                       {-141506549415LL, 6157981489484LL, -1602233112249LL, 4987354612LL},
                                       {-1079458116660LL, 46977158603111LL, -12312746836236LL, 64898311033LL},
                                       {-111443571855LL, 4850104782358LL, -1278622187073LL, 15013381034LL}};
constexpr int64_t EXIT_NV_INT[4][4] = {{34804924LL, -60095168892LL, 478555604051LL, -31235164290LL},
                                       {571743634LL, -987190198929LL, 7863736475042LL, -532253038680LL},
                                       {4361446936LL, -7530617199006LL, 60001569615953LL, -4178036642670LL},
                                       {450277058LL, -777463806933LL, 6195760216744LL, -441369753660LL}};
// Entry rows: the virtual history (u_0, u_-1, u_-2) as integer combinations of the S-box outputs 0..3 of full round 3
// (times a generic factor per row; lane 4 does not enter), and v_0 = the lane-4 S-box output itself (W_0 = 28 X_4).
constexpr int64_t ENTRY_N_INT[3][4] = {{-35LL, 252LL, -630LL, 660LL},
                                       {-14040740LL, 66398598LL, -98452620LL, 46450965LL},
                                       {-2121912819635LL, 9983493173052LL, -14744906064630LL, 6935043456660LL}};
constexpr int ENTRY_DIGITS[3] = {1, 1, 2};
constexpr int32_t ENTRY_W0_INT = 28;  // = 13 D / K

struct SparseRound {
    FrHost w[4];    // row 4 of M''_q, columns 0..3
    FrHost d;       // M''_q[4][4]
    FrHost b[4];    // column 4 of M''_q, rows 0..3
    FrHost add4;    // constant folded into lane 4 of this layer's output (next S-box input constant)
};

struct HadesTables {
    // ---- (A) sparse schedule: field VALUES ----
    FrHost c_first[WIDTH];                // ARC of round 0
    FrHost full_add[FULL_ROUNDS][WIDTH];  // vector added after the matrix of full round f (f = 0..7)
    FrHost mds[WIDTH][WIDTH];             // M
    FrHost mds_pre[WIDTH][WIDTH];         // M'_1 * M, used by full round index 3
    SparseRound sparse[PARTIAL_ROUNDS];
    FrHost last_add[4];                   // lanes 0..3 after the last sparse layer
    // ---- the partial rounds as a linear system (true field values; shared by the tests) ----
    FrHost arma_a[4];                     // a_1..a_4: A^4 = a1 A^3 + a2 A^2 + a3 A + a4 I
    FrHost arma_beta[5];                  // beta_0..beta_4
    FrHost exit_gy[4][4];                 // lanes 0..3 after round 60 = Gy (u_58..u_61) + Gv (v_57..v_60) + const
    FrHost exit_gv[4][4];
    // ---- (B) integer MDS, all 68 rounds: RESIDUES exactly as the device holds them ----
    FrHost int_kappa[ROUNDS][WIDTH];      // added (at the low end of the row accumulator) by the linear layer of round k
    FrHost int_g[PARTIAL_ROUNDS];         // G_k: equalises the scale of the lane-4 S-box output in partial round k
    FrHost int_f;                         // F: restores the Montgomery scale R after round 67
    bool int_ok;                          // mds.bin really is R/(i+j+5), and ARMA_*_INT match it
    // ---- (C) integer ARMA: residues as the device holds them ----
    FrHost ai_kappa[FULL_ROUNDS][WIDTH];  // integer rows of the full rounds f = 0..7 (f = 3: lane 4 only)
    FrHost ai_ent_fix[3];                 // entry: the one generic product of each virtual-history row (U_0, U_-1, U_-2)
    FrHost ai_ent_add[4];                 //        additive constants of U_0, U_-1, U_-2, W_0
    FrHost ai_k[PARTIAL_ROUNDS];          // K_{q+1} at [q-1], q = 1..60
    FrHost ai_g[PARTIAL_ROUNDS];          // G_q at [q-1]
    FrHost ai_ex_fix[4];                  // exit rows: 2^58 R' / den_i (the one generic product of an exit row)
    FrHost ai_ex_add[4];
    FrHost ai_f;
};

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

inline FrHost fr_from_i64(int64_t v) { return v < 0 ? FrHost::from_u64((uint64_t)(-v)).neg() : FrHost::from_u64((uint64_t)v); }
inline FrHost fr_pow_u(const FrHost& b, unsigned e) {
    FrHost r = FrHost::one();
    for (unsigned i = 0; i < e; ++i) r = r * b;
    return r;
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
    // ================= (A) sparse schedule =================
    // ---- forward-push the partial-round constants ----
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
    // ---- sparse factorisation, last partial round first ----
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

    // ================= the partial rounds as a linear system =================
    // With M = [[A, b], [c^T, d]]:  x_{q+1} = A x_q + b v_q + const,  u_{q+1} = c^T x_q + d v_q + const
    // (x = lanes 0..3, u = S-box input of lane 4, v = u^5).  Cayley-Hamilton on A eliminates x:
    //   u_{q+1} = sum a_m u_{q+1-m} + sum beta_n v_{q-n} + kappa_{q+1};
    // the lanes 0..3 are recovered from the history through the observability matrix (exit_gy / exit_gv).
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
    }

    // ================= (B) integer MDS, all rounds =================
    //   stored Z = s_k * (true S-box input of round k), one scale for all five lanes
    //   full round     X_j = Z_j^5 / R'^4                                   scale e = s^5 / R'^4
    //   partial round  X_j = Z_j (j<4), X_4 = Z_4^5/R'^4 * G_k/R'           G_k = R'^5 / s^4  =>  e = s
    //   linear layer   Z'_i = (sum_j N_ij X_j + kappa_i) / 2^29             s' = e L / (R 2^29),
    //                                                                       kappa_i = 2^29 s' C_{k+1}[i]
    //   after round 67 out_i R = Z'_i F / R',  F = R R' / s_68.     tests/pymodel.py::derive_int is the twin.
    const FrHost RP = FrHost::pow2(261), RM = FrHost::pow2(256), T29 = FrHost::pow2(29);
    const FrHost RP4inv = ((RP * RP) * (RP * RP)).inv(), RP5 = (RP * RP) * (RP * RP) * RP;
    const FrHost step = FrHost::from_u64((uint64_t)INT_L) * RM.inv() * T29.inv();
    T.int_ok = true;
    for (int i = 0; i < WIDTH; ++i)
        for (int j = 0; j < WIDTH; ++j)  // a different mds.bin must not pass silently
            if (!(M[i][j] * FrHost::from_u64((uint64_t)(i + j + 5)) == RM)) T.int_ok = false;
    {
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

    // ================= (C) integer ARMA =================
    // a_m = R^m a~_m and beta_n = R^(n+1) b~_n with A_m = a~_m D^m, B_n = b~_n D^n K one-digit integers.
    // Stored U_q = sigma_q u_q with sigma_{q+1} = sigma_q mu, mu = D / R; W_q = (sigma_q D / K) v_q =
    // sbox(U_q) * G_q / R' with G_q = R'^5 D / (K sigma_q^4).  Then (hades29.hpp::ai_round)
    //   U_{q+1} = sum_m A_m U_{q+1-m} + sum_n B_n W_{q-n} + K_{q+1}  (mod p, reduced from the top: fr29.hpp fold_top),
    //   K_{q+1} = sigma_{q+1} kappa_{q+1}
    // — every term at the same weight.  (Round 2 used mu = D / (R 2^29): a Montgomery digit step per round of age.)
    // All additive constants come from the zero-input trajectory of the affine system (they are
    // trajectory-independent).  Entry: (u_0, u_-1, u_-2, v_0) = H y_1 + h_0 (y_1 = state entering the first
    // partial round) is the virtual history for which the recurrence already holds at q = 1..4 with
    // kappa_2..5 = 0 and v_-1.. = 0.  tests/pymodel.py::derive_armaint is the twin.
    {
        const FrHost Df = FrHost::from_u64((uint64_t)INT_D), Kf = FrHost::from_u64((uint64_t)INT_K);
        for (int m = 1; m <= 4; ++m)
            if (!(T.arma_a[m - 1] * fr_pow_u(Df, m) == fr_pow_u(RM, m) * fr_from_i64(ARMA_A_INT[m - 1]))) T.int_ok = false;
        for (int n = 0; n < 5; ++n)
            if (!(T.arma_beta[n] * fr_pow_u(Df, n) * Kf == fr_pow_u(RM, n + 1) * fr_from_i64(ARMA_B_INT[n]))) T.int_ok = false;
        const FrHost* a = T.arma_a;
        const FrHost* beta = T.arma_beta;
        // zero-input trajectory from the first partial-round state y1 (5 lanes): u[q], x[q] for q = 1..61
        struct Traj {
            FrHost x[PARTIAL_ROUNDS + 2][4], u[PARTIAL_ROUNDS + 2];
        };
        auto traj = [&](const FrHost y1[WIDTH], Traj& t) {
            for (int i = 0; i < 4; ++i) t.x[1][i] = y1[i];
            t.u[1] = y1[4];
            for (int q = 1; q <= PARTIAL_ROUNDS; ++q) {
                const int k = RF + q;  // constants added after partial round q
                for (int i = 0; i < 4; ++i) {
                    FrHost acc = C[k][i];
                    for (int j = 0; j < 4; ++j) acc = acc + A[i][j] * t.x[q][j];
                    t.x[q + 1][i] = acc;
                }
                FrHost acc = C[k][4];
                for (int j = 0; j < 4; ++j) acc = acc + cvec[j] * t.x[q][j];
                t.u[q + 1] = acc;
            }
        };
        auto residuals = [&](const Traj& t, FrHost r[4]) {
            for (int q = 1; q <= 4; ++q) {
                FrHost acc = t.u[q + 1];
                for (int m = 1; m <= 4; ++m)
                    if (q + 1 - m >= 1) acc = acc - a[m - 1] * t.u[q + 1 - m];
                r[q - 1] = acc;
            }
        };
        const FrHost a4inv = a[3].inv(), b4inv = beta[4].inv();
        auto solve = [&](const FrHost r[4], FrHost th[4]) {  // th = (u_0, u_-1, u_-2, v_0)
            const FrHost v0 = r[3] * b4inv;
            const FrHost um0 = (r[2] - beta[3] * v0) * a4inv;
            const FrHost um1 = (r[1] - beta[2] * v0 - a[2] * um0) * a4inv;
            const FrHost um2 = (r[0] - beta[1] * v0 - a[1] * um0 - a[2] * um1) * a4inv;
            th[0] = um0, th[1] = um1, th[2] = um2, th[3] = v0;
        };
        std::vector<Traj> tr(2);
        Traj &t0 = tr[0], &tj = tr[1];
        FrHost zero5[WIDTH];
        for (int i = 0; i < WIDTH; ++i) zero5[i] = FrHost::zero();
        traj(zero5, t0);
        FrHost kappa[PARTIAL_ROUNDS + 2];  // kappa[q+1], q = 1..60
        for (int q = 1; q <= PARTIAL_ROUNDS; ++q) {
            if (q <= 4) {
                kappa[q + 1] = FrHost::zero();
                continue;
            }
            FrHost acc = t0.u[q + 1];
            for (int m = 1; m <= 4; ++m) acc = acc - a[m - 1] * t0.u[q + 1 - m];
            kappa[q + 1] = acc;
        }
        FrHost exit_add[4];
        for (int i = 0; i < 4; ++i) {
            FrHost acc = t0.x[PARTIAL_ROUNDS + 1][i];
            for (int r = 0; r < 4; ++r) acc = acc - T.exit_gy[i][r] * t0.u[58 + r];
            exit_add[i] = acc;
        }
        FrHost r0[4], th0[4], H[4][WIDTH];
        residuals(t0, r0);
        solve(r0, th0);
        for (int j = 0; j < WIDTH; ++j) {
            FrHost e[WIDTH];
            for (int i = 0; i < WIDTH; ++i) e[i] = (i == j) ? FrHost::one() : FrHost::zero();
            traj(e, tj);
            FrHost rj[4], col[4];
            residuals(tj, rj);
            for (int i = 0; i < 4; ++i) rj[i] = rj[i] - r0[i];
            solve(rj, col);
            for (int i = 0; i < 4; ++i) H[i][j] = col[i];
        }
        // ---- scales ----
        const FrHost mu = Df * RM.inv();
        const FrHost mu_inv = mu.inv();
        FrHost s = RM, sigma1 = FrHost::one(), e3 = FrHost::one();
        for (int k = 0; k < RF; ++k) {
            const FrHost e = s.pow5() * RP4inv;
            const FrHost sn = e * step;
            for (int i = 0; i < WIDTH; ++i) T.ai_kappa[k][i] = (k < RF - 1 || i == 4) ? T29 * sn * C[k + 1][i] : FrHost::zero();
            if (k == RF - 1) sigma1 = sn * T29, e3 = e;  // U_1 = the integer row of lane 4 WITHOUT its digit step (arma_entry): e L / R
            s = sn;
        }
        FrHost sig[PARTIAL_ROUNDS + 5];  // sig[q + 2] = sigma_q for q = -2..61
        sig[3] = sigma1;
        for (int q = 2; q <= PARTIAL_ROUNDS + 1; ++q) sig[q + 2] = sig[q + 1] * mu;
        for (int q = 0; q >= -2; --q) sig[q + 2] = sig[q + 3] * mu_inv;
        const FrHost DK = Df * Kf.inv();
        {  // entry rows: theta_i = sum_j (H M)_ij v_j + (H C_4 + h0)_i,  v_j = X_j / e3
            const FrHost tscale[4] = {sig[0 + 2], sig[-1 + 2], sig[-2 + 2], sig[0 + 2] * DK};
            const FrHost e3inv = e3.inv();
            for (int i = 0; i < 4; ++i) {
                FrHost hc = th0[i];
                for (int t = 0; t < WIDTH; ++t) hc = hc + H[i][t] * C[RF][t];
                T.ai_ent_add[i] = tscale[i] * hc;
                FrHost gen[WIDTH];  // generic form of the row: coefficient * R'
                for (int j = 0; j < WIDTH; ++j) {
                    FrHost hm = FrHost::zero();
                    for (int t = 0; t < WIDTH; ++t) hm = hm + H[i][t] * M[t][j];
                    gen[j] = tscale[i] * hm * e3inv * RP;
                }
                // integer form (hades29.hpp::entry_row): theta_i = (sum_j n_ij X_j) / 2^(29 digits_i) * fix_i / R' + add_i
                if (i < 3) {
                    const FrHost shift = FrHost::pow2(29u * (unsigned)ENTRY_DIGITS[i]);
                    T.ai_ent_fix[i] = gen[0] * shift * fr_from_i64(ENTRY_N_INT[i][0]).inv();
                    for (int j = 0; j < 4; ++j)
                        if (!(gen[j] * shift == T.ai_ent_fix[i] * fr_from_i64(ENTRY_N_INT[i][j]))) T.int_ok = false;
                    if (!gen[4].is_zero()) T.int_ok = false;
                } else {
                    for (int j = 0; j < 4; ++j)
                        if (!gen[j].is_zero()) T.int_ok = false;
                    if (!(gen[4] == RP * FrHost::from_u64((uint64_t)ENTRY_W0_INT))) T.int_ok = false;
                }
            }
        }
        for (int q = 1; q <= PARTIAL_ROUNDS; ++q) {
            const FrHost s2 = sig[q + 2] * sig[q + 2];
            T.ai_g[q - 1] = RP5 * DK * (s2 * s2).inv();
            T.ai_k[q - 1] = sig[q + 1 + 2] * kappa[q + 1];
        }
        // exit rows in integer form (hades29.hpp::exit_row):
        //   Z_i = ( sum_r ny_ir U_{58+r} + sum_t nv_it W_{57+t} ) / 2^58 * fix_i / R' + add_i
        s = sig[PARTIAL_ROUNDS + 1 + 2];
        for (int i = 0; i < 4; ++i) {
            const FrHost den = FrHost::from_u64((uint64_t)EXIT_DEN_INT[i]);
            for (int r = 0; r < 4; ++r) {
                if (!(T.exit_gy[i][r] * fr_pow_u(Df, 3 - r) * den == fr_pow_u(RM, 3 - r) * fr_from_i64(EXIT_NY_INT[i][r]))) T.int_ok = false;
                if (!(T.exit_gv[i][r] * fr_pow_u(Df, 3 - r) * Kf * den == fr_pow_u(RM, 4 - r) * fr_from_i64(EXIT_NV_INT[i][r]))) T.int_ok = false;
            }
            T.ai_ex_fix[i] = FrHost::pow2(58) * den.inv() * RP;
            T.ai_ex_add[i] = s * exit_add[i];
        }
        for (int k = RF + PARTIAL_ROUNDS; k < ROUNDS; ++k) {
            s = s.pow5() * RP4inv * step;
            for (int i = 0; i < WIDTH; ++i)
                T.ai_kappa[k - PARTIAL_ROUNDS][i] = k + 1 < ROUNDS ? T29 * s * C[k + 1][i] : FrHost::zero();
        }
        T.ai_f = RM * RP * s.inv();
    }
}

// =============================================================================================
// Device encoding for the 29-bit-limb kernels (fr29.hpp)
// =============================================================================================
// Flat int32 table, 9 digits per constant, balanced digits in [-2^28, 2^28] of the CENTRED integer
// representative (|n| <= p/2).  Schedule (A): field values in one of three forms
//   additive constants ("A"):  n = a * 2^256            (same form as the state)
//   multipliers of plain state lanes ("MP"): n = c * 2^261          (redc divides by 2^261)
//   multipliers of S-box outputs ("MS"):     n = c * 2^261 * 2^20   (S-box output carries 2^-20)
// Schedules (B), (C): the residues of HadesTables as they are ("raw"), plus plain one-digit integers.
struct Tab29Layout {
    static constexpr int C_FIRST = 0;                                   // [5][9]   A     (all schedules)
    // ---- (C) integer ARMA: what the kernels read ----
    static constexpr int INT_N = C_FIRST + WIDTH * NL;                  // [9] ints: N[i][j] = h[i+j], h[d] = L/(d+5)
    static constexpr int AI_AB = INT_N + NL;                            // [9] ints: A_1..A_4, B_0..B_4
    static constexpr int AI_KAPPA = AI_AB + NL;                         // [8][5][9] raw
    static constexpr int AI_ENT_N = AI_KAPPA + FULL_ROUNDS * WIDTH * NL;  // [3][9] ints: 4 two-digit coefficients (lo, hi) + 1 pad per row
    static constexpr int AI_ENT_FIX = AI_ENT_N + 3 * NL;                // [3][9] raw
    static constexpr int AI_ENT_ADD = AI_ENT_FIX + 3 * NL;              // [4][9] raw
    static constexpr int AI_KG = AI_ENT_ADD + 4 * NL;                   // [60][2][9] raw: K_{q+1}, G_q per round (contiguous)
    static constexpr int AI_EX_N = AI_KG + PARTIAL_ROUNDS * 2 * NL;     // [4][18] ints: 8 two-digit coefficients (lo, hi) + 2 pad per row
    static constexpr int AI_EX_FIX = AI_EX_N + 4 * 2 * NL;              // [4][9] raw
    static constexpr int AI_EX_ADD = AI_EX_FIX + 4 * NL;                // [4][9] raw
    static constexpr int AI_F = AI_EX_ADD + 4 * NL;                     // [9] raw
    // ---- (B) integer MDS in all rounds (host cross-check) ----
    static constexpr int INT_KAPPA = AI_F + NL;                         // [68][5][9] raw
    static constexpr int INT_G = INT_KAPPA + ROUNDS * WIDTH * NL;       // [60][9] raw
    static constexpr int INT_F = INT_G + PARTIAL_ROUNDS * NL;           // [9] raw
    // ---- (A) sparse (host cross-check) ----
    static constexpr int FULL_ADD = INT_F + NL;                         // [8][5][9] A
    static constexpr int MDS = FULL_ADD + FULL_ROUNDS * WIDTH * NL;     // [5][5][9] MS
    static constexpr int MDS_PRE = MDS + WIDTH * WIDTH * NL;            // [5][5][9] MS
    static constexpr int SPARSE = MDS_PRE + WIDTH * WIDTH * NL;         // [60][SPARSE_STRIDE]
    //   per sparse round: w[4][9] MP | d[9] MS | b[4][9] MS | add4[9] A
    static constexpr int SP_W = 0, SP_D = 4 * NL, SP_B = 5 * NL, SP_ADD4 = 9 * NL;
    static constexpr int SPARSE_STRIDE = 10 * NL;
    static constexpr int LAST_ADD = SPARSE + PARTIAL_ROUNDS * SPARSE_STRIDE;  // [4][9] A
    static constexpr int TOTAL = LAST_ADD + 4 * NL;
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
    const FrHost one = FrHost::one();
    auto put = [&](int off, const FrHost& v, const FrHost& scale) { encode_balanced29(v * scale, &tab[off]); };
    for (int i = 0; i < WIDTH; ++i) put(Lay::C_FIRST + i * NL, T.c_first[i], fA);
    // (C)
    for (int d = 0; d < 2 * WIDTH - 1; ++d) tab[Lay::INT_N + d] = INT_L / (d + 5);
    for (int m = 0; m < 4; ++m) tab[Lay::AI_AB + m] = ARMA_A_INT[m];
    for (int n = 0; n < 5; ++n) tab[Lay::AI_AB + 4 + n] = ARMA_B_INT[n];
    for (int f = 0; f < FULL_ROUNDS; ++f)
        for (int i = 0; i < WIDTH; ++i) put(Lay::AI_KAPPA + (f * WIDTH + i) * NL, T.ai_kappa[f][i], one);
    auto split2 = [&](int off, int64_t n) {  // balanced two-digit split: n = lo + hi 2^29, |lo| <= 2^28
        int64_t lo = n & (int64_t)DMASK;
        if (lo > ((int64_t)1 << (WB - 1))) lo -= (int64_t)1 << WB;
        tab[off] = (int32_t)lo;
        tab[off + 1] = (int32_t)((n - lo) >> WB);
    };
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 4; ++j) split2(Lay::AI_ENT_N + i * NL + 2 * j, ENTRY_N_INT[i][j]);
        put(Lay::AI_ENT_FIX + i * NL, T.ai_ent_fix[i], one);
    }
    for (int i = 0; i < 4; ++i) {
        put(Lay::AI_ENT_ADD + i * NL, T.ai_ent_add[i], one);
        for (int t = 0; t < 8; ++t) split2(Lay::AI_EX_N + i * 2 * NL + 2 * t, t < 4 ? EXIT_NY_INT[i][t] : EXIT_NV_INT[i][t - 4]);
        put(Lay::AI_EX_FIX + i * NL, T.ai_ex_fix[i], one);
        put(Lay::AI_EX_ADD + i * NL, T.ai_ex_add[i], one);
    }
    for (int q = 0; q < PARTIAL_ROUNDS; ++q) {
        put(Lay::AI_KG + (q * 2 + 0) * NL, T.ai_k[q], one);
        put(Lay::AI_KG + (q * 2 + 1) * NL, T.ai_g[q], one);
    }
    put(Lay::AI_F, T.ai_f, one);
    // (B)
    for (int k = 0; k < ROUNDS; ++k)
        for (int i = 0; i < WIDTH; ++i) put(Lay::INT_KAPPA + (k * WIDTH + i) * NL, T.int_kappa[k][i], one);
    for (int q = 0; q < PARTIAL_ROUNDS; ++q) put(Lay::INT_G + q * NL, T.int_g[q], one);
    put(Lay::INT_F, T.int_f, one);
    // (A)
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
    return tab;
}

// Worst-case |column| (as a double) over every lazy accumulation the three schedules perform, for the
// ACTUAL constants in `tab`: state digits are bounded by 2^29 (un-carried lanes of schedule (B) by 2^30.1),
// top digits by 2^27 (|V| < 16 p; the values the kernels' schedule produces stay below 13 p — measured by the
// instrumented host build, tests/test_host_arith.py::test_dynamic_bounds), the high-column initialisation by 2^30.
// A full tight reduction (redc) adds at most 2^29 * sum(p digits) + carries; a wide one (redc_w, the kernels' schedule
// (C) and its S-boxes) at most 2^31 * sum |balanced p digits| + 8 * 2^31 of carry + the 2^31 bias — the larger of the
// two is charged everywhere.  The kernels are correct iff this stays below 2^63.
inline double max_column_bound29(const int32_t* tab) {
    typedef Tab29Layout Lay;
    const double DIG = 536870912.0 /* 2^29 */, TOP = 134217728.0 /* 2^27 */, LAZY = 1151000000.0 /* 2^30.1 */;
    const double P_SUM = (double)P252_P29_1 + P252_P29_2 + P252_P29_3 + P252_P29_4 + P252_P29_5 + P252_P29_6 +
                         P252_P29_7 + P252_P29_8;
    auto ab = [](double v) { return v < 0 ? -v : v; };
    const double PB_SUM = ab(P252_PB_1) + ab(P252_PB_2) + ab(P252_PB_3) + ab(P252_PB_4) + ab(P252_PB_5) + ab(P252_PB_6) +
                          ab(P252_PB_7) + ab(P252_PB_8);
    const double REDC_T = DIG * P_SUM + 68719476736.0 /* carries < 2^36 */ + 1073741824.0 /* hi init 2^30 */;
    const double REDC_W = 2147483648.0 * PB_SUM + 8.0 * 2147483648.0 + 2147483648.0 + 1073741824.0;
    const double REDC = REDC_T > REDC_W ? REDC_T : REDC_W;
    const double WSTEP = 2147483648.0 * ab(P252_PB_5);  // the largest single contribution of one wide step to a column
    double worst = 0;
    auto absd = [](int32_t v) { return v < 0 ? -(double)v : (double)v; };
    const double WIDE = 2147483648.0;  // |wide digit| <= 2^31 (fr29.hpp redc_w<true>): S-box outputs, x^5 of a partial round, W_q
    auto group = [&](std::initializer_list<int> offsets, bool wide = false) {  // sum of 9-digit x 9-digit products into 18 columns
        double col[2 * NL] = {0};
        for (int off : offsets)
            for (int j = 0; j < NL; ++j)
                for (int i = 0; i < NL; ++i) col[i + j] += (i == NL - 1 ? TOP : (wide ? WIDE : DIG)) * absd(tab[off + j]);
        for (int k = 0; k < 2 * NL; ++k)
            if (col[k] + REDC > worst) worst = col[k] + REDC;
    };
    // (A)
    for (int base : {Lay::MDS, Lay::MDS_PRE})
        for (int k = 0; k < WIDTH; ++k)
            group({base + (k * 5 + 0) * NL, base + (k * 5 + 1) * NL, base + (k * 5 + 2) * NL, base + (k * 5 + 3) * NL,
                   base + (k * 5 + 4) * NL});
    for (int q = 0; q < PARTIAL_ROUNDS; ++q) {
        const int b = Lay::SPARSE + q * Lay::SPARSE_STRIDE;
        group({b + Lay::SP_W, b + Lay::SP_W + NL, b + Lay::SP_W + 2 * NL, b + Lay::SP_W + 3 * NL, b + Lay::SP_D});
        for (int i = 0; i < 4; ++i) group({b + Lay::SP_B + i * NL});
    }
    // (B), (C): generic products
    for (int q = 0; q < PARTIAL_ROUNDS; ++q) {
        group({Lay::INT_G + q * NL});
        group({Lay::AI_KG + (q * 2 + 1) * NL}, true);  // x^5 (wide digits) times G_q
    }
    group({Lay::INT_F});
    group({Lay::AI_F});
    for (int i = 0; i < 4; ++i) {
        if (i < 3) {
            group({Lay::AI_ENT_FIX + i * NL});
            double esum = 0;  // integer entry row: at most one digit product per coefficient digit per column, two digit steps
            for (int t = 0; t < 8; ++t) esum += absd(tab[Lay::AI_ENT_N + i * NL + t]);
            const double erow = WIDE * esum + 2.0 * WSTEP + 68719476736.0;  // (the S-box outputs of round 3: wide digits)
            if (erow > worst) worst = erow;
        }
        group({Lay::AI_EX_FIX + i * NL});
        // integer exit row: a column collects at most one digit product per coefficient digit (16 of them), then two digit steps
        double nsum = 0;  // (coefficients 0..7 meet U_58..U_61: carried digits; 8..15 meet W_57..W_60: wide digits)
        for (int t = 0; t < 16; ++t) nsum += (t < 8 ? DIG : WIDE) * absd(tab[Lay::AI_EX_N + i * 2 * NL + t]);
        const double row = nsum + 2.0 * WSTEP + 68719476736.0;
        if (row > worst) worst = row;
    }
    {  // integer rows: five one-digit terms (row 0 has the largest sum) of possibly un-carried lanes + kappa + one digit step
        double nsum = 0;
        for (int j = 0; j < WIDTH; ++j) nsum += absd(tab[Lay::INT_N + j]);
        const double row = (LAZY > WIDE ? LAZY : WIDE) * nsum + 268435456.0 + DIG * (double)P252_P29_1 + 68719476736.0;
        if (row > worst) worst = row;
    }
    {  // integer ARMA row: a column collects one digit of each of the nine terms (all at the same weight) and K's digit, then
        // the quotient of the reduction from the top (|q| < 2^29) times one balanced digit of p
        double asum = 0;  // (A_1..A_4 meet U: carried digits; B_0..B_4 meet W: wide digits)
        for (int t = 0; t < NL; ++t) asum += (t < 4 ? DIG : WIDE) * absd(tab[Lay::AI_AB + t]);
        const double row = asum + 536870912.0 * ab(P252_PB_5) + DIG + 68719476736.0;
        if (row > worst) worst = row;
    }
    // S-box: element x element (9 products of 2^29 x 2^29) and squarings (<= 4.5 * 2^59)
    const double sbox_col = 9.0 * DIG * DIG + REDC;
    if (sbox_col > worst) worst = sbox_col;
    return worst;
}

}  // namespace p252
