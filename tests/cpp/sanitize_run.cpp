// sanitize_run.cpp — the CPU-side code under AddressSanitizer + UndefinedBehaviorSanitizer (tests/test_sanitizers.py).
// Built from: the oracle (oracle/p252_oracle.c) and the DEVICE arithmetic headers compiled for the host
// (poseidon252_amd/csrc/hosttest.cpp: fr29.hpp, hades29.hpp, coop29.hpp, tables.hpp).  UBSan's signed-overflow check is the
// point for the latter: the kernels' lazy arithmetic accumulates up to 45 products in signed 64-bit columns and never
// reduces in between — an overflow anywhere on these inputs aborts the run.  Every result is also compared with the oracle.
// The reference has no sanitizer or race tooling of its own (single-threaded safe Rust).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../poseidon252_amd/csrc/hosttest.cpp"
extern "C" {
#include "../../oracle/p252_oracle.h"
}

static int fails = 0;
#define EXPECT(cond, what)                                 \
    do {                                                   \
        if (!(cond)) {                                     \
            std::fprintf(stderr, "FAIL: %s\n", what);      \
            ++fails;                                       \
        }                                                  \
    } while (0)

static std::vector<uint64_t> pattern_scalars() {  // saturated / edge 256-bit patterns (not all below p)
    const uint64_t pats[][4] = {
        {~0ull, ~0ull, ~0ull, ~0ull}, {12345, 0, 0, 0x8000000000000000ull}, {0, 0, 0, 0}, {1, 0, 0, 0},
        {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull},  // p
        {0xffffffff00000000ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull},  // p - 1
        {0x5555555555555555ull, 0x5555555555555555ull, 0x5555555555555555ull, 0x5555555555555555ull},
        {0xaaaaaaaaaaaaaaaaull, 0xaaaaaaaaaaaaaaaaull, 0xaaaaaaaaaaaaaaaaull, 0xaaaaaaaaaaaaaaaaull},
        {0, 0, 0, 0xffffff0000000000ull}};
    std::vector<uint64_t> v;
    for (auto& p : pats) v.insert(v.end(), p, p + 4);
    return v;
}

int main(int argc, char** argv) {
    // 1. the reference's known-answer digests through the oracle (argv: n then the expected 64-hex-digit big-endian value, pairs)
    static const char* KAT_IN[10] = {
        "bb67ed265bf1db490ded2e1ede55c0d14c55521509dc73f9c354e98ab76c9625", "7e74220084d75e10c89e9435d47bb5b8075991b2e29be3b84421dac3b1ee6007",
        "5ce5481a4d78cca03498f72761da1b9f1d2aa8fb300be39f0e4fe2534f9d4308", "b1e710e3c4a8c35154b0ce4e4f4af6f498ebd79f8e7cdf3150372c7501be250b",
        "33c9e2025f86b5d82149f1ab8e20a168fc3d99d09b48cbce0286db8752cc3306", "e98206bfdce791e4e5144079b997d4fc25006194b35655f0e48490b26e24ea35",
        "86d2a95cc552de8d5bb20bd4a407fee5ffdc314e93dfe6b2dc792bc71fd8cc2d", "4edd8307ce28a8c70963d20a7bc28df1e1720bbbc93878a18bd07fad7d51fa15",
        "eabc7a296704a68aa01f95adc85f6dd758b175745336d8fc795a17984024b21e", "cfc108673c93df305e31c283b9c767b7097ae4e174a223e0c24b15a67b701a3a"};
    uint8_t in[10][32];
    for (int i = 0; i < 10; ++i)
        for (int b = 0; b < 32; ++b) {
            unsigned x;
            std::sscanf(KAT_IN[i] + 2 * b, "%2x", &x);
            in[i][b] = (uint8_t)x;
        }
    int kats = 0;
    for (int a = 1; a + 1 < argc; a += 2) {
        const int n = std::atoi(argv[a]);
        uint8_t out[32];
        p252o_kat_hash(&in[0][0], (size_t)n, out);
        char hex[65];
        for (int b = 0; b < 32; ++b) std::snprintf(hex + 2 * b, 3, "%02x", out[31 - b]);
        EXPECT(std::strcmp(hex, argv[a + 1]) == 0, "reference KAT digest");
        ++kats;
    }

    // 2. the kernels' three schedules, the digest specialisation and the lane-group forms against the oracle
    const size_t n = 96;
    std::vector<uint64_t> st(n * 20), exp(n * 20), got(n * 20);
    p252o_fill_random(2024, st.data(), n * 5);
    const std::vector<uint64_t> pats = pattern_scalars();
    for (size_t i = 0; i < 40 * 5; ++i) std::memcpy(&st[i * 4], &pats[((i * 7 + i / 5) % (pats.size() / 4)) * 4], 32);
    // (the oracle takes reduced inputs: compare on the patterns' residues — the device code reads 256 bits as an integer)
    std::vector<uint64_t> red(st);
    for (size_t i = 0; i < n * 5; ++i)
        if (!p252o_is_reduced(&red[i * 4])) {  // value mod p by repeated subtraction (patterns are < 2^256 < 3p)
            static const uint64_t P[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
            while (!p252o_is_reduced(&red[i * 4])) {
                unsigned __int128 borrow = 0;
                for (int k = 0; k < 4; ++k) {
                    const unsigned __int128 d = (unsigned __int128)red[i * 4 + k] - P[k] - (uint64_t)borrow;
                    red[i * 4 + k] = (uint64_t)d;
                    borrow = (d >> 64) ? 1 : 0;
                }
            }
        }
    p252o_permute_batch(red.data(), exp.data(), n);
    for (int sched = 0; sched < 3; ++sched) {
        ht_permute29_sched(st.data(), got.data(), n, sched);
        EXPECT(std::memcmp(got.data(), exp.data(), n * 160) == 0, "host build of a kernel schedule vs oracle permutation");
    }
    const size_t nc = 40;  // (the lane groups are threads meeting at a barrier ~600 times per permutation: keep it short)
    for (int lanes : {8, 4}) {
        EXPECT(ht_permute_coop(st.data() + 30 * 20, got.data(), nc, lanes) == 0, "ht_permute_coop");
        EXPECT(std::memcmp(got.data(), exp.data() + 30 * 20, nc * 160) == 0, "lane-group permutation vs oracle");
    }
    {
        std::vector<uint64_t> tag(4), dig(n * 4), dexp(n * 4);
        p252o_fill_random(5, tag.data(), 1);
        p252o_hash_batch(tag.data(), red.data(), 4, 1, dexp.data(), n);  // the first 4 n scalars of the buffer as n nodes
        ht_merkle4_digest29(tag.data(), st.data(), dig.data(), n);
        EXPECT(std::memcmp(dig.data(), dexp.data(), n * 32) == 0, "digest specialisation vs oracle");
        for (int lanes : {8, 4}) {
            ht_merkle4_digest_coop(tag.data(), st.data(), dig.data(), nc, lanes);
            EXPECT(std::memcmp(dig.data(), dexp.data(), nc * 32) == 0, "lane-group digest vs oracle");
        }
    }

    // 3. the oracle's own composite paths on ragged sizes (heap buffers exactly sized: ASan sees any overrun)
    for (size_t leaves : {1u, 2u, 5u, 17u, 64u, 257u}) {
        std::vector<uint64_t> lv(leaves * 4), tag(4), root(4);
        p252o_fill_random(leaves, lv.data(), leaves);
        p252o_fill_random(3, tag.data(), 1);
        size_t total = 0, m = leaves;  // (oracle/__init__.py levels_total)
        do {
            m = (m + 3) / 4;
            total += m;
        } while (m > 1);
        std::vector<uint64_t> levels(total * 4);
        EXPECT(p252o_merkle4_tree(tag.data(), lv.data(), leaves, root.data(), levels.data()) >= 0, "oracle tree");
    }
    for (size_t len : {1u, 2u, 4u, 5u, 21u, 42u})
        for (int variant = 0; variant < 2; ++variant) {
            std::vector<uint64_t> tag(4), msg(len * 4), sec(8), non(4), c((len + 1) * 4), back(len * 4);
            p252o_encryption_tag_v(variant, len, tag.data());
            p252o_fill_random(len, msg.data(), len);
            p252o_fill_random(len + 100, sec.data(), 2);
            p252o_fill_random(len + 200, non.data(), 1);
            EXPECT(p252o_encrypt_v(variant, tag.data(), msg.data(), len, sec.data(), non.data(), c.data()) == 0, "oracle encrypt");
            EXPECT(p252o_decrypt_v(variant, tag.data(), c.data(), len, sec.data(), non.data(), back.data()) == 0, "oracle decrypt");
            EXPECT(std::memcmp(back.data(), msg.data(), len * 32) == 0, "oracle encryption round trip");
        }
    std::printf("sanitize_run: %d KAT digests, %zu states x 3 schedules + lane groups of 8 and 4, trees, encryption: %s\n", kats, n,
                fails ? "FAILED" : "clean");
    return fails ? 1 : 0;
}
