// fr_host.hpp — host-side BLS12-381 scalar-field arithmetic used by the PRODUCT library to derive
// its device constant tables (sparse partial-round matrices, folded round constants) and for the
// few host-only helpers (tag, truncation).  It is NOT a CPU fallback for hashing: every hashing
// entry point runs on the GPU or fails.  Independent of oracle/ by construction.
//
// Semantics follow dusk-bls12_381 0.14 `BlsScalar` (call sites in the reference:
// src/hades/permutation/scalar.rs:34,47,51,59; src/hades/mds_matrix.rs:32): 4 x u64 little-endian
// limbs holding a*2^256 mod p, always fully reduced.
#pragma once
#include <cstdint>
#include <cstring>

namespace p252 {

typedef unsigned __int128 u128_t;

struct FrHost {
    uint64_t l[4];

    static constexpr uint64_t P[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL,
                                      0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
    static constexpr uint64_t PINV = 0xfffffffeffffffffULL;  // -p^{-1} mod 2^64
    static constexpr uint64_t R2[4] = {0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL,
                                       0x05d314967254398fULL, 0x0748d9d99f59ff11ULL};

    static FrHost zero() { return FrHost{{0, 0, 0, 0}}; }
    static FrHost from_limbs(const uint64_t* m) {
        FrHost r;
        std::memcpy(r.l, m, 32);
        return r;
    }
    // integer (any 256-bit value, LE u64 limbs) -> Montgomery form of (v mod p)   [from_raw]
    static FrHost from_raw(const uint64_t v[4]) {
        FrHost a = from_limbs(v), r2 = from_limbs(R2);
        return mont_mul(a, r2);
    }
    static FrHost from_u64(uint64_t v) {
        uint64_t raw[4] = {v, 0, 0, 0};
        return from_raw(raw);
    }
    static FrHost one() { return from_u64(1); }

    static bool geq_p(const uint64_t a[4]) {
        for (int i = 3; i >= 0; --i) {
            if (a[i] > P[i]) return true;
            if (a[i] < P[i]) return false;
        }
        return true;
    }
    static void sub_p(uint64_t a[4]) {
        u128_t borrow = 0;
        for (int i = 0; i < 4; ++i) {
            u128_t d = (u128_t)a[i] - P[i] - borrow;
            a[i] = (uint64_t)d;
            borrow = (d >> 64) & 1;
        }
    }
    static FrHost mont_mul(const FrHost& a, const FrHost& b) {
        uint64_t t[9] = {0};
        for (int i = 0; i < 4; ++i) {
            u128_t c = 0;
            for (int j = 0; j < 4; ++j) {
                c += (u128_t)a.l[i] * b.l[j] + t[i + j];
                t[i + j] = (uint64_t)c;
                c >>= 64;
            }
            t[i + 4] = (uint64_t)c;
        }
        uint64_t top = 0;
        for (int i = 0; i < 4; ++i) {
            uint64_t m = t[i] * PINV;
            u128_t c = 0;
            for (int j = 0; j < 4; ++j) {
                c += (u128_t)m * P[j] + t[i + j];
                t[i + j] = (uint64_t)c;
                c >>= 64;
            }
            c += (u128_t)t[i + 4] + top;
            t[i + 4] = (uint64_t)c;
            top = (uint64_t)(c >> 64);
        }
        FrHost r = from_limbs(t + 4);
        if (top || geq_p(r.l)) sub_p(r.l);
        return r;
    }
    // canonical integer limbs of the field value
    void to_canonical(uint64_t out[4]) const {
        FrHost onei{{1, 0, 0, 0}};
        FrHost r = mont_mul(*this, onei);
        std::memcpy(out, r.l, 32);
    }
    bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
    bool operator==(const FrHost& o) const { return std::memcmp(l, o.l, 32) == 0; }

    FrHost operator+(const FrHost& b) const {
        FrHost r;
        u128_t c = 0;
        for (int i = 0; i < 4; ++i) {
            c += (u128_t)l[i] + b.l[i];
            r.l[i] = (uint64_t)c;
            c >>= 64;
        }
        if (geq_p(r.l)) sub_p(r.l);
        return r;
    }
    FrHost neg() const {
        if (is_zero()) return *this;
        FrHost r;
        u128_t borrow = 0;
        for (int i = 0; i < 4; ++i) {
            u128_t d = (u128_t)P[i] - l[i] - borrow;
            r.l[i] = (uint64_t)d;
            borrow = (d >> 64) & 1;
        }
        return r;
    }
    FrHost operator-(const FrHost& b) const { return *this + b.neg(); }
    FrHost operator*(const FrHost& b) const { return mont_mul(*this, b); }
    FrHost pow(const uint64_t e[4]) const {
        FrHost acc = one(), base = *this;
        for (int i = 0; i < 256; ++i) {
            if ((e[i / 64] >> (i % 64)) & 1) acc = acc * base;
            base = base * base;
        }
        return acc;
    }
    FrHost inv() const {  // Fermat: a^(p-2)
        uint64_t e[4] = {P[0] - 2, P[1], P[2], P[3]};
        return pow(e);
    }
    FrHost pow5() const {
        FrHost x2 = *this * *this;
        return x2 * x2 * *this;
    }
    bool is_square() const {  // Euler criterion
        uint64_t e[4];        // (p - 1) / 2
        for (int i = 0; i < 4; ++i) e[i] = (P[i] >> 1) | (i < 3 ? P[i + 1] << 63 : 0);
        return is_zero() || pow(e) == one();
    }
    // Tonelli-Shanks; p - 1 = 2^32 * q with q odd.  Precondition: is_square().
    FrHost sqrt() const {
        if (is_zero()) return *this;
        uint64_t q[4], qp1h[4];  // q = (p - 1) >> 32 ; (q + 1) / 2
        const uint64_t pm1[4] = {P[0] - 1, P[1], P[2], P[3]};
        for (int i = 0; i < 4; ++i) q[i] = (pm1[i] >> 32) | (i < 3 ? pm1[i + 1] << 32 : 0);
        uint64_t t4[4] = {q[0] + 1, q[1], q[2], q[3]};  // q odd -> no carry out of limb 0 beyond +1 on an odd value
        for (int i = 0; i < 4; ++i) qp1h[i] = (t4[i] >> 1) | (i < 3 ? t4[i + 1] << 63 : 0);
        FrHost z = from_u64(2);
        while (z.is_square()) z = z + one();
        unsigned m = 32;
        FrHost c = z.pow(q), t = pow(q), r = pow(qp1h);
        const FrHost I = one();
        while (!(t == I)) {
            unsigned i = 0;
            FrHost t2 = t;
            while (!(t2 == I)) {
                t2 = t2 * t2;
                ++i;
            }
            FrHost b = c;
            for (unsigned k = 0; k + i + 1 < m; ++k) b = b * b;
            m = i;
            c = b * b;
            t = t * c;
            r = r * b;
        }
        return r;
    }
    // any x with x^4 == *this.  Precondition: *this is a 4th power.
    FrHost fourth_root() const {
        FrHost r = sqrt();
        if (!r.is_square()) r = r.neg();
        return r.sqrt();
    }
    // 2^k as a field element
    static FrHost pow2(unsigned k) {
        FrHost r = one(), two = from_u64(2);
        for (unsigned i = 0; i < k; ++i) r = r * two;
        return r;
    }
};

}  // namespace p252
