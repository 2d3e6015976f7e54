// blake2b.hpp — unkeyed BLAKE2b-512 (RFC 7693) for the host-side tag helper (p252_tag).
// dusk-bls12_381's BlsScalar::hash_to_scalar (called at src/hades/permutation/scalar.rs:29-31) hashes
// the tag input with BLAKE2b; the crate is not vendored in the reference, so this follows the RFC.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>

namespace p252 {

class Blake2b512 {
  public:
    Blake2b512() {
        for (int i = 0; i < 8; ++i) h_[i] = iv(i);
        h_[0] ^= 0x01010040ULL;  // depth 1, fanout 1, key 0, digest 64
    }
    void update(const uint8_t* p, size_t n) {
        while (n) {
            if (fill_ == 128) {  // a full buffer is only compressed once more input arrives
                total_ += 128;
                compress(false);
                fill_ = 0;
            }
            size_t take = 128 - fill_ < n ? 128 - fill_ : n;
            std::memcpy(buf_ + fill_, p, take);
            fill_ += take;
            p += take;
            n -= take;
        }
    }
    void finish(uint8_t out[64]) {
        total_ += fill_;
        std::memset(buf_ + fill_, 0, 128 - fill_);
        compress(true);
        for (int i = 0; i < 64; ++i) out[i] = (uint8_t)(h_[i >> 3] >> (8 * (i & 7)));
    }

  private:
    static uint64_t iv(int i) {
        static const uint64_t k[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL,
                                      0xa54ff53a5f1d36f1ULL, 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL,
                                      0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
        return k[i];
    }
    static uint64_t ror(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
    void compress(bool last) {
        static const uint8_t sigma[10][16] = {
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
            {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
            {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
            {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
            {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};
        uint64_t m[16], v[16];
        for (int i = 0; i < 16; ++i) {
            uint64_t w = 0;
            for (int b = 7; b >= 0; --b) w = (w << 8) | buf_[8 * i + b];
            m[i] = w;
        }
        for (int i = 0; i < 8; ++i) {
            v[i] = h_[i];
            v[8 + i] = iv(i);
        }
        v[12] ^= total_;
        if (last) v[14] = ~v[14];
        auto mix = [&](int a, int b, int c, int d, uint64_t x, uint64_t y) {
            v[a] += v[b] + x; v[d] = ror(v[d] ^ v[a], 32);
            v[c] += v[d];     v[b] = ror(v[b] ^ v[c], 24);
            v[a] += v[b] + y; v[d] = ror(v[d] ^ v[a], 16);
            v[c] += v[d];     v[b] = ror(v[b] ^ v[c], 63);
        };
        for (int r = 0; r < 12; ++r) {
            const uint8_t* s = sigma[r % 10];
            mix(0, 4, 8, 12, m[s[0]], m[s[1]]);
            mix(1, 5, 9, 13, m[s[2]], m[s[3]]);
            mix(2, 6, 10, 14, m[s[4]], m[s[5]]);
            mix(3, 7, 11, 15, m[s[6]], m[s[7]]);
            mix(0, 5, 10, 15, m[s[8]], m[s[9]]);
            mix(1, 6, 11, 12, m[s[10]], m[s[11]]);
            mix(2, 7, 8, 13, m[s[12]], m[s[13]]);
            mix(3, 4, 9, 14, m[s[14]], m[s[15]]);
        }
        for (int i = 0; i < 8; ++i) h_[i] ^= v[i] ^ v[8 + i];
    }
    uint64_t h_[8];
    uint8_t buf_[128];
    size_t fill_ = 0;
    uint64_t total_ = 0;
};

inline void blake2b_512(const uint8_t* msg, size_t len, uint8_t out[64]) {
    Blake2b512 st;
    st.update(msg, len);
    st.finish(out);
}

}  // namespace p252
