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
    return tab;
}

}  // namespace p252
