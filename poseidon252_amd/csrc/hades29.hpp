// hades29.hpp — the Hades permutation and the SAFE sponge on E29 lazy residues (one state per lane).
//
// Replaces Hades::perm (src/hades/permutation.rs:105-123) with its ARC / S-box / MDS steps
// (src/hades/permutation/scalar.rs:39-64) and the sponge loop dusk-safe runs for Hash::finalize
// (src/hash.rs:128-155).  Executes the sparse-partial-round schedule whose constants
// tables.hpp derives; the field elements produced are identical to the reference schedule's.
//
// TP is any pointer-like giving int32 digits: tab[i].  In kernels it is a wave-uniform pointer so
// the compiler keeps constants in SGPRs (s_load) and feeds them to v_mad_i64_i32 as scalar operands.
#pragma once
#include "fr29.hpp"
#include "tables.hpp"

namespace p252 {

// One full round: state <- Mat * sbox(state) + add     (ARC of this round was folded into the
// previous layer's `add`).  5 S-boxes (15 mults, 15 redc) + 25 products + 5 redc.
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
// 3 + 9 products-of-elements, 3 + 5 redc.
template <class TP>
P252_HD void partial_round(E29 s[WIDTH], TP sp) {
    typedef Tab29Layout Lay;
    const E29 v = sbox(s[4]);
    A29 t;
    acc_set_hi_c(t, sp + Lay::SP_ADD4);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc_mul(t, s[j], sp + Lay::SP_W + j * NL);
    acc_mul(t, v, sp + Lay::SP_D);
    const E29 y4 = redc(t);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        A29 u;
        acc_set_hi(u, s[i]);
        acc_mul(u, v, sp + Lay::SP_B + i * NL);
        s[i] = redc(u);
    }
    s[4] = y4;
}

// The whole permutation as ONE loop with a wave-uniform branch, so that the (large, fully unrolled)
// full-round and partial-round bodies each exist once in the instruction stream.
template <class TP>
P252_HD void hades_permute(E29 s[WIDTH], TP tab) {
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

}  // namespace p252
