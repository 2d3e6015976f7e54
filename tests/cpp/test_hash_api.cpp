// C++ host-side tests of include/poseidon252.hpp on the GPU, written to read like the reference's own
// tests: README.md:22-51 (doctest), tests/hash.rs:101-116,188-203,277-292 (input shapes 3/5/15,
// truncated, multi-output (3,3) (5,2) (4,7)), src/hades.rs:94-162 (known-answer test, through the HIP
// sponge with tag = 0), src/hash.rs:124-137 (panics -> exceptions).  The oracle (oracle/p252_oracle.h)
// is linked as the checker only.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "poseidon252.hpp"
#include "../../oracle/p252_oracle.h"

using namespace dusk_poseidon_hip;

static int failures = 0;
#define EXPECT(cond)                                                        \
    do {                                                                    \
        if (!(cond)) {                                                      \
            std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond);     \
            ++failures;                                                     \
        }                                                                   \
    } while (0)

static std::vector<BlsScalar> random_scalars(uint64_t seed, size_t n) {
    std::vector<BlsScalar> v(n);
    p252o_fill_random(seed, v[0].data(), n);
    return v;
}

static std::vector<BlsScalar> oracle_hash(const BlsScalar& tag, const std::vector<BlsScalar>& in, size_t in_len, size_t out_len) {
    const size_t n = in.size() / in_len;
    std::vector<BlsScalar> out(n * out_len);
    p252o_hash_batch(tag.data(), in[0].data(), in_len, out_len, out[0].data(), n);
    return out;
}

template <class F>
static bool throws_io(F f, IoPatternError::Kind kind) {
    try {
        f();
    } catch (const IoPatternError& e) {
        return e.kind == kind;
    } catch (...) {
    }
    return false;
}

// the HIP runtime calls a caller with device-resident data needs (libamdhip64), declared here: this test builds with plain g++
extern "C" {
int hipMalloc(void** ptr, std::size_t bytes);
int hipFree(void* ptr);
int hipMemcpy(void* dst, const void* src, std::size_t bytes, int kind);  // 1 = host to device, 2 = device to host
int hipDeviceSynchronize(void);
}

int main() {
    // ---- README.md:31-51 ----
    {
        auto input = random_scalars(0xbeef, 42);
        auto hash = Hash::new_(Domain::Other);
        hash.update(input.data(), 3);
        hash.update(input.data() + 3, 39);
        EXPECT(hash.finalize() == Hash::digest(Domain::Other, input));  // chunked update == one-shot digest
        std::vector<BlsScalar> four(input.begin(), input.begin() + 4);
        EXPECT(Hash::digest(Domain::Merkle4, four) != Hash::digest(Domain::Other, four));
    }
    // ---- round_constants.rs:56-71: to_bytes -> from_bytes is the identity; all-ones is not a valid encoding ----
    {
        auto xs = random_scalars(0xb17e, 100);
        std::vector<std::uint8_t> ok;
        EXPECT(from_bytes(to_bytes(xs), ok) == xs);
        for (auto o : ok) EXPECT(o == 1);
        ScalarBytes ff;
        ff.fill(0xff);
        from_bytes({ff}, ok);
        EXPECT(ok.size() == 1 && ok[0] == 0);
    }
    // ---- tests/hash.rs shapes against the oracle ----
    for (size_t n_in : {3, 5, 15}) {
        auto input = random_scalars(0xbeef + n_in, n_in);
        auto got = Hash::digest(Domain::Other, input);
        EXPECT(got.size() == 1);
        EXPECT(got == oracle_hash(compute_tag(Domain::Other, {n_in}, 1), input, n_in, 1));
    }
    for (auto io : {std::pair<size_t, size_t>{3, 3}, {5, 2}, {4, 7}}) {
        auto input = random_scalars(0xbeef + 16 * io.first + io.second, io.first);
        Hash h(Domain::Other);
        h.output_len(io.second);
        h.update(input);
        auto got = h.finalize();
        EXPECT(got.size() == io.second);
        EXPECT(got == oracle_hash(compute_tag(Domain::Other, {io.first}, io.second), input, io.first, io.second));
    }
    {  // truncated (hash.rs:164-183)
        auto input = random_scalars(0xbeef + 5, 5);
        auto t = Hash::digest_truncated(Domain::Other, input);
        auto full = Hash::digest(Domain::Other, input);
        JubJubRaw exp;
        p252o_truncate250(full[0].data(), exp.data());
        EXPECT(t.size() == 1 && t[0] == exp && (t[0][3] >> 58) == 0);
        // the batched form (HashBatch::digest_truncated: one launch, the digest kernel's truncating output stage), tests/hash.rs:188-203 shape
        HashBatch hb(Domain::Other, 5);
        auto many = random_scalars(0xbeef + 5, 5 * 40);  // item 0 = `input` above
        auto tb = hb.digest_truncated(many);
        auto fb = hb.digest(many);
        EXPECT(tb.size() == 40 && tb[0] == exp);
        for (std::size_t i = 0; i < tb.size(); ++i) {
            JubJubRaw e;
            p252o_truncate250(fb[i].data(), e.data());
            EXPECT(tb[i] == e);
        }
        Context::default_context().wipe();  // (and the context keeps working afterwards)
        EXPECT(hb.digest_truncated(many) == tb);
    }
    // ---- output_len rules (hash.rs:111-115) and panics (hash.rs:124-137) ----
    {
        auto input = random_scalars(1, 8);
        Hash m4(Domain::Merkle4);
        m4.output_len(3);  // ignored for Merkle4
        m4.update(input.data(), 4);
        EXPECT(m4.finalize().size() == 1);
        EXPECT(throws_io([&] { Hash h(Domain::Merkle4); h.update(input.data(), 3); h.finalize(); }, IoPatternError::IOPatternViolation));
        EXPECT(throws_io([&] { Hash h(Domain::Merkle2); h.update(input.data(), 4); h.finalize(); }, IoPatternError::IOPatternViolation));
        EXPECT(throws_io([&] { Hash h(Domain::Other); h.finalize(); }, IoPatternError::InvalidIOPattern));
        EXPECT(throws_io([&] { HashBatch hb(Domain::Merkle4, 5); }, IoPatternError::IOPatternViolation));
        EXPECT(domain_separator(Domain::Merkle4) == 0xf && domain_separator(Domain::Merkle2) == 0x3 &&
               domain_separator(Domain::Encryption) == 0x100000000ULL && domain_separator(Domain::Other) == 0);
    }
    // ---- the reference's known-answer test through the HIP sponge (tag forced to zero) ----
    {
        static const char* INPUTS[10] = {
            "bb67ed265bf1db490ded2e1ede55c0d14c55521509dc73f9c354e98ab76c9625", "7e74220084d75e10c89e9435d47bb5b8075991b2e29be3b84421dac3b1ee6007",
            "5ce5481a4d78cca03498f72761da1b9f1d2aa8fb300be39f0e4fe2534f9d4308", "b1e710e3c4a8c35154b0ce4e4f4af6f498ebd79f8e7cdf3150372c7501be250b",
            "33c9e2025f86b5d82149f1ab8e20a168fc3d99d09b48cbce0286db8752cc3306", "e98206bfdce791e4e5144079b997d4fc25006194b35655f0e48490b26e24ea35",
            "86d2a95cc552de8d5bb20bd4a407fee5ffdc314e93dfe6b2dc792bc71fd8cc2d", "4edd8307ce28a8c70963d20a7bc28df1e1720bbbc93878a18bd07fad7d51fa15",
            "eabc7a296704a68aa01f95adc85f6dd758b175745336d8fc795a17984024b21e", "cfc108673c93df305e31c283b9c767b7097ae4e174a223e0c24b15a67b701a3a"};
        struct { size_t n; const char* be_hex; } KAT[6] = {
            {3, "26abf2d0476f154e69bf19740092fe36265680c294462b8e759ad73a99567dd5"}, {4, "1cc40219c7ec92919d6db7a41cd41953333a2ed544606daca182e4eaa6c7db2d"},
            {5, "707c98a0e9a6e4832ac33ee08811bce122017a58dbbbf66a2f6fcdc69d45462d"}, {6, "26905a794d3d2fb0c3ed2276abc696c27a5bfdea7f106e596cbeedd86891c461"},
            {8, "1b98a2c5f1fe54d21b5ce9bf0dcc99ea8784a64f3c544fa06d3f73569741006e"}, {10, "211b7ea21c9afca93dabdfbda8b2d5275b2dd802fed87bb431e98557c61667d2"}};
        std::vector<BlsScalar> all(11);
        for (int i = 0; i < 10; ++i) {
            uint64_t raw[4] = {0, 0, 0, 0};
            for (int b = 0; b < 32; ++b) {  // from_hex_str: little-endian canonical bytes
                unsigned byte;
                std::sscanf(INPUTS[i] + 2 * b, "%2x", &byte);
                raw[b / 8] |= (uint64_t)byte << (8 * (b % 8));
            }
            p252o_from_raw(raw, all[i].data());
        }
        const uint64_t one_raw[4] = {1, 0, 0, 0};
        BlsScalar one;
        p252o_from_raw(one_raw, one.data());
        for (auto& k : KAT) {
            std::vector<BlsScalar> msg(all.begin(), all.begin() + k.n);
            msg.push_back(one);  // the KAT pads with BlsScalar::one()
            Hash h(Domain::Other);
            h.set_tag(BlsScalar{0, 0, 0, 0});
            h.update(msg);
            auto out = h.finalize();
            uint64_t canon[4];
            p252o_to_canonical(out[0].data(), canon);
            char hex[65];
            for (int b = 0; b < 32; ++b) std::snprintf(hex + 2 * b, 3, "%02x", (unsigned)((canon[3 - b / 8] >> (8 * (7 - b % 8))) & 0xff));
            EXPECT(std::string(hex) == k.be_hex);
        }
    }
    // ---- HashBatch == n x Hash::digest == oracle; Merkle root ----
    {
        HashBatch hb(Domain::Merkle4, 4);
        auto input = random_scalars(0xc10d, 4 * 3000);
        auto got = hb.digest(input);
        EXPECT(got == oracle_hash(hb.tag(), input, 4, 1));
        std::vector<BlsScalar> first(input.begin(), input.begin() + 4);
        EXPECT(Hash::digest(Domain::Merkle4, first)[0] == got[0]);
        HashBatch hb5(Domain::Other, 42, 5);
        auto m = random_scalars(7, 42 * 100);
        EXPECT(hb5.digest(m) == oracle_hash(hb5.tag(), m, 42, 5));
        auto leaves = random_scalars(9, 1000);
        BlsScalar exp_root;
        p252o_merkle4_tree(hb.tag().data(), leaves[0].data(), leaves.size(), exp_root.data(), nullptr);
        EXPECT(merkle4_root(leaves) == exp_root);
        {  // caller-owned buffer page-locked for the scope: same digests
            HostRegistration reg(input.data(), input.size() * sizeof(BlsScalar));
            EXPECT(hb.digest(input) == got);
        }
    }
    // ---- several contexts (one per GPU; here all on device 0): same bytes as one context, sharded tree root ----
    {
        Context c0(0), c1(0), c2(0), c3(0);
        std::vector<Context*> ctxs = {&c0, &c1, &c2, &c3};
        HashBatch hb(Domain::Merkle4, 4);
        auto input = random_scalars(0xabc, 4 * 1001);
        EXPECT(digest_multi(ctxs, hb, 4, input) == hb.digest(input));
        auto leaves = random_scalars(11, 4 * 256);  // 4 complete subtrees of 4^4 leaves
        EXPECT(merkle4_root_multi(ctxs, leaves) == merkle4_root(leaves));
        bool threw = false;
        try {
            merkle4_root_multi(ctxs, random_scalars(12, 4 * 100));  // 100 leaves per device: not 4^k
        } catch (const std::invalid_argument&) {
            threw = true;
        }
        EXPECT(threw);
    }
    // ---- the library's RCCL communicator at one rank (real backend) and the forest entry point, device-resident data ----
    {
        const std::size_t n_leaves = 4096, per_tree = 64;  // one 4^6-leaf tree = a forest of 64 trees of 4^3 leaves + one 64-leaf tree on top
        auto leaves = random_scalars(0xf0e, n_leaves);
        void *d_leaves = nullptr, *d_root = nullptr, *d_roots = nullptr;
        EXPECT(hipMalloc(&d_leaves, 32 * n_leaves) == 0 && hipMalloc(&d_root, 32) == 0 && hipMalloc(&d_roots, 32 * n_leaves / per_tree) == 0);
        EXPECT(hipMemcpy(d_leaves, leaves.data(), 32 * n_leaves, 1) == 0);
        Context c(0);
        {
            Comm comm(c, Comm::unique_id(), 0, 1);
            EXPECT(comm.rank() == 0 && comm.size() == 1);
            comm.merkle4_root_sharded_device(d_leaves, n_leaves, d_root);
            comm.check();  // ABI 8: waits for the stream; no peer reported a failed build
            BlsScalar got{};
            EXPECT(hipDeviceSynchronize() == 0 && hipMemcpy(got.data(), d_root, 32, 2) == 0);
            EXPECT(got == merkle4_root(leaves));
            EXPECT(Comm::backend().find("rccl") != std::string::npos);  // resolved at run time: this binary links no librccl
            c.trim();  // the grow-only scratch given back; the communicator stays usable
            comm.merkle4_root_sharded_device(d_leaves, n_leaves, d_root);
            comm.check();
            EXPECT(hipMemcpy(got.data(), d_root, 32, 2) == 0 && got == merkle4_root(leaves));
        }
        {
            Context c1(0);
            std::vector<Context*> one = {&c1};
            auto comms = Comm::create_all(one);
            EXPECT(comms.size() == 1 && comms[0].size() == 1);
            comms[0].merkle4_root_sharded_device(d_leaves, n_leaves, d_root);
            BlsScalar got{};
            EXPECT(hipDeviceSynchronize() == 0 && hipMemcpy(got.data(), d_root, 32, 2) == 0);
            EXPECT(got == merkle4_root(leaves));
        }
        merkle4_forest_device(d_leaves, n_leaves / per_tree, per_tree, d_roots, c);
        std::vector<BlsScalar> roots(n_leaves / per_tree);
        EXPECT(hipDeviceSynchronize() == 0 && hipMemcpy(roots.data(), d_roots, 32 * roots.size(), 2) == 0);
        for (std::size_t t = 0; t < roots.size(); ++t)
            EXPECT(roots[t] == merkle4_root(std::vector<BlsScalar>(leaves.begin() + t * per_tree, leaves.begin() + (t + 1) * per_tree)));
        EXPECT(merkle4_root(roots) == merkle4_root(leaves));
        EXPECT(merkle4_forest(leaves, per_tree, c) == roots);  // the host-buffer twin
        hipFree(d_leaves);
        hipFree(d_root);
        hipFree(d_roots);
    }
    // ---- encrypt / decrypt (src/encryption.rs:62-95; tests/encryption.rs properties), both call sequences ----
    for (int variant : {P252_CRYPT_STREAM, P252_CRYPT_DUPLEX}) {
        for (std::size_t len : {std::size_t(2), std::size_t(21), std::size_t(42)}) {
            const std::size_t n = 17;
            auto msgs = random_scalars(100 + len, n * len), secrets = random_scalars(200 + len, 2 * n), nonces = random_scalars(300 + len, n);
            auto cipher = encrypt_batch(msgs, len, secrets, nonces, variant);
            EXPECT(cipher.size() == n * (len + 1));
            std::vector<BlsScalar> exp(n * (len + 1));
            const BlsScalar tag = encryption_tag(len, variant);
            for (std::size_t i = 0; i < n; ++i)
                p252o_encrypt_v(variant, tag.data(), msgs[i * len].data(), len, secrets[2 * i].data(), nonces[i].data(), exp[i * (len + 1)].data());
            EXPECT(cipher == exp);
            std::vector<std::uint8_t> ok;
            EXPECT(decrypt_batch(cipher, len, secrets, nonces, ok, variant) == msgs);
            EXPECT(std::all_of(ok.begin(), ok.end(), [](std::uint8_t v) { return v == 1; }));
            auto wrong = nonces;
            std::swap(wrong[0], wrong[1]);
            decrypt_batch(cipher, len, secrets, wrong, ok, variant);
            EXPECT(ok[0] == 0 && ok[1] == 0 && ok[2] == 1);  // wrong nonce -> DecryptionFailed for exactly those items
            std::vector<BlsScalar> one(cipher.begin(), cipher.begin() + len + 1);
            EXPECT(decrypt(one, secrets[0], secrets[1], nonces[0], variant) == std::vector<BlsScalar>(msgs.begin(), msgs.begin() + len));
            bool failed = false;
            try {
                decrypt(one, secrets[1], secrets[0], nonces[0], variant);
            } catch (const DecryptionFailed&) {
                failed = true;
            }
            EXPECT(failed);
        }
    }
    std::printf(failures ? "C++ HOST API: %d FAILURES\n" : "C++ HOST API: ALL PASSED\n", failures);
    return failures ? 1 : 0;
}
