// poseidon252.hpp — C++17 host-side mirror of dusk_poseidon::{Domain, Hash} over the C ABI
// (include/poseidon252_hip.h).  Header-only; link with -lposeidon252_hip.
//
// The reference (dusk-poseidon 0.42, Rust) is compiled code and no Rust toolchain exists in the build
// image, so the host side above the C ABI is provided in C++ (and in Python, poseidon252_amd/hash.py)
// with the reference's names, argument meaning and error behaviour:
//
//   reference (src/hash.rs)                       here
//   ------------------------------------------    -----------------------------------------------
//   enum Domain {Merkle4,Merkle2,Encryption,Other}  enum class Domain            (hash.rs:21-36)
//   impl From<Domain> for u64                       domain_separator(Domain)     (hash.rs:38-56)
//   Hash::new / output_len / update / finalize      Hash(...) / same names       (hash.rs:98-155)
//   Hash::finalize_truncated / digest[_truncated]   same names                   (hash.rs:164-210)
//   panic!("io-pattern should be valid")            throws IoPatternError        (hash.rs:124-137)
//   —                                               HashBatch: n messages, one kernel launch
//
// A BlsScalar is 4 little-endian u64 Montgomery limbs (a * 2^256 mod p), exactly the reference's
// memory layout, so buffers are interchangeable with a Rust &[BlsScalar].
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "poseidon252_hip.h"

namespace dusk_poseidon_hip {

using BlsScalar = std::array<uint64_t, 4>;     // Montgomery limbs, as dusk_bls12_381::BlsScalar
using JubJubRaw = std::array<uint64_t, 4>;     // argument of JubJubScalar::from_raw (hash.rs:180)
constexpr std::size_t HADES_WIDTH = P252_HADES_WIDTH;  // src/lib.rs:17

enum class Domain : int {  // declaration order of src/hash.rs:21-36
    Merkle4 = P252_DOMAIN_MERKLE4,
    Merkle2 = P252_DOMAIN_MERKLE2,
    Encryption = P252_DOMAIN_ENCRYPTION,
    Other = P252_DOMAIN_OTHER,
};

// dusk_poseidon::Error (src/error.rs:9-29) — the variants the hash path can produce.  Where the
// reference panics (`.expect`, hash.rs:131-154) this mirror throws.
struct IoPatternError : std::logic_error {
    enum Kind { IOPatternViolation, InvalidIOPattern } kind;
    IoPatternError(Kind k, const std::string& what) : std::logic_error(what), kind(k) {}
};
struct DeviceError : std::runtime_error {  // no HIP device / HIP failure: there is no CPU fallback
    using std::runtime_error::runtime_error;
};

inline uint64_t domain_separator(Domain d) {  // From<Domain> for u64
    uint64_t sep = 0;
    p252_domain_separator(static_cast<int>(d), &sep);
    return sep;
}

namespace detail {
inline void check(int rc, p252_ctx* ctx, const char* what) {
    if (rc == P252_OK) return;
    const std::string msg = std::string(what) + ": " + (ctx ? p252_last_error(ctx) : "");
    if (rc == P252_ERR_IO_PATTERN_VIOLATION)
        throw IoPatternError(IoPatternError::IOPatternViolation, "io-pattern should be valid: IOPatternViolation " + msg);
    if (rc == P252_ERR_INVALID_IO_PATTERN)
        throw IoPatternError(IoPatternError::InvalidIOPattern, "at this point the io-pattern is valid: InvalidIOPattern " + msg);
    if (rc == P252_ERR_INVALID_ARGUMENT) throw std::invalid_argument(msg);
    throw DeviceError(msg + " (rc " + std::to_string(rc) + ")");
}
}  // namespace detail

// One p252_ctx, bound to one HIP device (one process/thread per GPU).
class Context {
  public:
    explicit Context(int device = 0) {
        // argument lists have changed between library versions under unchanged names: never call into another interface
        if (p252_abi_version() != P252_ABI_VERSION)
            throw DeviceError("libposeidon252_hip.so implements ABI version " + std::to_string(p252_abi_version()) +
                              ", this header is version " + std::to_string(P252_ABI_VERSION));
        p252_ctx* c = nullptr;
        const int rc = p252_create(device, &c);
        if (rc != P252_OK) throw DeviceError(std::string("p252_create: ") + p252_last_error(nullptr));
        ctx_.reset(c, p252_destroy);
    }
    p252_ctx* get() const { return ctx_.get(); }
    // clears every scratch buffer the context owns (zeroize, Cargo.toml:14); the host-buffer encrypt / decrypt calls and the
    // destructor do so themselves
    void wipe() { detail::check(p252_wipe(ctx_.get()), ctx_.get(), "Context::wipe"); }
    // gives the grow-only scratch back (waits for the device, wipes, frees); the next call allocates what it needs again
    void trim() { detail::check(p252_trim(ctx_.get()), ctx_.get(), "Context::trim"); }
    static Context& default_context() {
        static Context c(0);
        return c;
    }

  private:
    std::shared_ptr<p252_ctx> ctx_;
};

// Page-locks a caller-owned buffer for its lifetime (p252_host_register / p252_host_unregister): host-buffer calls on it
// copy at PCIe speed instead of page-locking it on every call.  The buffer must outlive this object.
class HostRegistration {
  public:
    HostRegistration(void* p, std::size_t bytes) : p_(p) {
        if (p252_host_register(p, bytes) != P252_OK) throw DeviceError("p252_host_register failed");
    }
    ~HostRegistration() { (void)p252_host_unregister(p_); }
    HostRegistration(const HostRegistration&) = delete;
    HostRegistration& operator=(const HostRegistration&) = delete;

  private:
    void* p_;
};

// io_pattern() of src/hash.rs:62-85 plus dusk-safe's validation; throws like Hash::finalize panics
inline void check_io_pattern(Domain d, const std::vector<std::size_t>& absorb_lens, std::size_t output_len) {
    detail::check(p252_check_io_pattern(static_cast<int>(d), absorb_lens.data(), absorb_lens.size(), output_len), nullptr,
                  "io_pattern");
}

// Safe::tag for this io-pattern (scalar.rs:29-31).  UNPINNED recipe: pass the value
// from the real crates through the `tag` parameters instead when it is available.
inline BlsScalar compute_tag(Domain d, const std::vector<std::size_t>& absorb_lens, std::size_t output_len) {
    BlsScalar t{};
    detail::check(p252_tag(static_cast<int>(d), absorb_lens.data(), absorb_lens.size(), output_len, t.data()), nullptr, "tag");
    return t;
}

// ---- Hash: one message, absorbed in chunks, squeezed once (src/hash.rs:87-211) ----
class Hash {
  public:
    explicit Hash(Domain domain, Context& ctx = Context::default_context()) : domain_(domain), ctx_(ctx) {}  // Hash::new
    static Hash new_(Domain domain) { return Hash(domain); }

    // hash.rs:111-115: honoured only for Domain::Other and output_len > 0
    void output_len(std::size_t n) {
        if (domain_ == Domain::Other && n > 0) output_len_ = n;
    }
    // hash.rs:118-120: the slice is borrowed, never copied or modified; it must outlive finalize()
    void update(const BlsScalar* input, std::size_t len) { input_.emplace_back(input, len); }
    void update(const std::vector<BlsScalar>& input) { update(input.data(), input.size()); }

    // hash.rs:128-155.  Throws IoPatternError where the reference panics.
    std::vector<BlsScalar> finalize() const { return run(false); }
    // hash.rs:164-183: the digest kernel's output stage canonicalises, masks to 250 bits and stores the raw limbs
    // JubJubScalar::from_raw receives — one launch (p252_hash_batch_truncated)
    std::vector<JubJubRaw> finalize_truncated() const { return run(true); }

  private:
    std::vector<BlsScalar> run(bool truncated) const {
        std::vector<std::size_t> lens;
        std::size_t total = 0;
        for (auto& c : input_) {
            lens.push_back(c.second);
            total += c.second;
        }
        check_io_pattern(domain_, lens, output_len_);
        const BlsScalar tag = has_tag_ ? tag_ : compute_tag(domain_, lens, output_len_);
        std::vector<BlsScalar> msg;
        msg.reserve(total);
        for (auto& c : input_) msg.insert(msg.end(), c.first, c.first + c.second);
        std::vector<BlsScalar> out(output_len_);
        detail::check((truncated ? p252_hash_batch_truncated : p252_hash_batch)(ctx_.get(), tag.data(), msg[0].data(), total, output_len_,
                                                                                out[0].data(), 1),
                      ctx_.get(), truncated ? "Hash::finalize_truncated" : "Hash::finalize");
        return out;
    }

  public:
    // hash.rs:191-195, 203-210
    static std::vector<BlsScalar> digest(Domain domain, const std::vector<BlsScalar>& input) {
        Hash h(domain);
        h.update(input);
        return h.finalize();
    }
    static std::vector<JubJubRaw> digest_truncated(Domain domain, const std::vector<BlsScalar>& input) {
        Hash h(domain);
        h.update(input);
        return h.finalize_truncated();
    }
    // inject the capacity element computed by the real crates (BlsScalar::hash_to_scalar)
    void set_tag(const BlsScalar& tag) {
        tag_ = tag;
        has_tag_ = true;
    }

  private:
    Domain domain_;
    Context& ctx_;
    std::vector<std::pair<const BlsScalar*, std::size_t>> input_;
    std::size_t output_len_ = 1;
    BlsScalar tag_{};
    bool has_tag_ = false;
};

// ---- HashBatch: n independent messages with one io-pattern in one kernel launch.  Per item the
// result equals Hash::digest(domain, item): same validation, same tag, same output order. ----
class HashBatch {
  public:
    HashBatch(Domain domain, std::size_t item_len, std::size_t output_len = 1, Context& ctx = Context::default_context())
        : domain_(domain), item_len_(item_len), output_len_((domain == Domain::Other && output_len > 0) ? output_len : 1), ctx_(ctx) {
        check_io_pattern(domain_, {item_len_}, output_len_);
        tag_ = compute_tag(domain_, {item_len_}, output_len_);
    }
    void set_tag(const BlsScalar& tag) { tag_ = tag; }
    const BlsScalar& tag() const { return tag_; }
    std::size_t output_len() const { return output_len_; }

    // host buffers: input.size() must be a multiple of item_len
    std::vector<BlsScalar> digest(const std::vector<BlsScalar>& input) const {
        if (item_len_ == 0 || input.size() % item_len_) throw std::invalid_argument("HashBatch::digest: ragged input");
        const std::size_t n = input.size() / item_len_;
        std::vector<BlsScalar> out(n * output_len_);
        if (n)
            detail::check(p252_hash_batch(ctx_.get(), tag_.data(), input[0].data(), item_len_, output_len_, out[0].data(), n),
                          ctx_.get(), "HashBatch::digest");
        return out;
    }
    // Hash::digest_truncated per item (hash.rs:203-210), truncated inside the digest kernel: one launch
    std::vector<JubJubRaw> digest_truncated(const std::vector<BlsScalar>& input) const {
        if (item_len_ == 0 || input.size() % item_len_) throw std::invalid_argument("HashBatch::digest_truncated: ragged input");
        const std::size_t n = input.size() / item_len_;
        std::vector<JubJubRaw> out(n * output_len_);
        if (n)
            detail::check(p252_hash_batch_truncated(ctx_.get(), tag_.data(), input[0].data(), item_len_, output_len_, out[0].data(), n),
                          ctx_.get(), "HashBatch::digest_truncated");
        return out;
    }
    // device buffers, asynchronous on `stream` (a hipStream_t)
    void digest_device(const void* d_in, void* d_out, std::size_t n, void* stream = nullptr) const {
        detail::check(p252_hash_batch_device(ctx_.get(), tag_.data(), d_in, item_len_, output_len_, d_out, n, stream), ctx_.get(),
                      "HashBatch::digest_device");
    }
    void digest_truncated_device(const void* d_in, void* d_out_raw, std::size_t n, void* stream = nullptr) const {
        detail::check(p252_hash_batch_truncated_device(ctx_.get(), tag_.data(), d_in, item_len_, output_len_, d_out_raw, n, stream),
                      ctx_.get(), "HashBatch::digest_truncated_device");
    }

  private:
    Domain domain_;
    std::size_t item_len_, output_len_;
    Context& ctx_;
    BlsScalar tag_{};
};

// Arity-4 Merkle root over Hash::digest(Domain::Merkle4, ..) nodes (empty slots = zero, hash.rs:22-26)
inline BlsScalar merkle4_root(const std::vector<BlsScalar>& leaves, Context& ctx = Context::default_context()) {
    const BlsScalar tag = compute_tag(Domain::Merkle4, {4}, 1);
    BlsScalar root{};
    if (leaves.empty()) throw std::invalid_argument("merkle4_root: no leaves");
    detail::check(p252_merkle4_tree(ctx.get(), tag.data(), leaves[0].data(), leaves.size(), root.data(), nullptr), ctx.get(),
                  "merkle4_root");
    return root;
}

// ---- several GPUs from one process: one Context per device, sharded inside the library (p252_*_multi) ----
// digests[i*output_len..] == Hash::digest(domain, item i); contiguous shards, no inter-GPU dependence
inline std::vector<BlsScalar> digest_multi(const std::vector<Context*>& ctxs, const HashBatch& hb, std::size_t item_len,
                                           const std::vector<BlsScalar>& input) {
    if (ctxs.empty() || item_len == 0 || input.size() % item_len) throw std::invalid_argument("digest_multi: bad arguments");
    std::vector<p252_ctx*> raw;
    for (Context* c : ctxs) raw.push_back(c->get());
    const std::size_t n = input.size() / item_len;
    std::vector<BlsScalar> out(n * hb.output_len());
    if (n)
        detail::check(p252_hash_batch_multi(raw.data(), raw.size(), hb.tag().data(), input[0].data(), item_len, hb.output_len(),
                                            out[0].data(), n),
                      raw[0], "digest_multi");
    return out;
}
// root of the arity-4 tree over ctxs.size() * 4^k leaves: one complete subtree per device, 32-byte roots gathered on the host
// (host leaves: they stream in through each context's staging lanes; device-resident shards: p252_merkle4_tree_multi_device, RCCL)
inline BlsScalar merkle4_root_multi(const std::vector<Context*>& ctxs, const std::vector<BlsScalar>& leaves) {
    if (ctxs.empty() || leaves.empty()) throw std::invalid_argument("merkle4_root_multi: bad arguments");
    std::vector<p252_ctx*> raw;
    for (Context* c : ctxs) raw.push_back(c->get());
    const BlsScalar tag = compute_tag(Domain::Merkle4, {4}, 1);
    BlsScalar root{};
    detail::check(p252_merkle4_tree_multi(raw.data(), raw.size(), tag.data(), leaves[0].data(), leaves.size(), root.data()), raw[0],
                  "merkle4_root_multi");
    return root;
}

// ---- RCCL communicator of the library (p252_comm_*): the constants are broadcast and validated when it is created, the
// sharded tree build all-gathers its 32-byte subtree roots on the stream.  One Comm per Context.  One process per GPU:
// rank 0 calls Comm::unique_id() and hands the bytes to the other ranks, every rank constructs Comm(ctx, id, rank, world).
// One process, several GPUs: Comm::create_all(contexts on distinct devices). ----
class Comm {
  public:
    using Id = std::array<unsigned char, P252_COMM_ID_BYTES>;
    static Id unique_id() {
        Id id{};
        detail::check(p252_comm_unique_id(id.data(), id.size()), nullptr, "p252_comm_unique_id");
        return id;
    }
    Comm(Context& ctx, const Id& id, int rank, int world) : ctx_(ctx) {
        p252_comm* c = nullptr;
        detail::check(p252_comm_create_rank(ctx.get(), id.data(), id.size(), rank, world, &c), ctx.get(), "p252_comm_create_rank");
        comm_.reset(c, p252_comm_destroy);
    }
    static std::vector<Comm> create_all(const std::vector<Context*>& ctxs) {
        if (ctxs.empty()) throw std::invalid_argument("Comm::create_all: no contexts");
        std::vector<p252_ctx*> raw;
        for (Context* c : ctxs) raw.push_back(c->get());
        std::vector<p252_comm*> out(raw.size(), nullptr);
        detail::check(p252_comm_create_all(raw.data(), raw.size(), out.data()), raw[0], "p252_comm_create_all");
        std::vector<Comm> v;
        for (std::size_t t = 0; t < raw.size(); ++t) v.push_back(Comm(*ctxs[t], out[t]));
        return v;
    }
    int rank() const { return p252_comm_rank(comm_.get()); }
    int size() const { return p252_comm_size(comm_.get()); }
    // this rank's 4^k device-resident leaves -> root over ALL ranks' leaves in d_root (32 B, device), asynchronously on
    // `stream`; collective (every rank calls it)
    void merkle4_root_sharded_device(const void* d_leaves, std::size_t n_leaves_local, void* d_root, void* stream = nullptr) {
        const BlsScalar tag = compute_tag(Domain::Merkle4, {4}, 1);
        detail::check(p252_merkle4_tree_sharded_device(comm_.get(), tag.data(), d_leaves, n_leaves_local, d_root, stream), ctx_.get(),
                      "merkle4_root_sharded_device");
    }
    // waits for `stream`, then throws if a sharded build of this communicator met a failed peer since the last check (that build's
    // root is all-ones on every healthy rank): p252_comm_check
    void check(void* stream = nullptr) { detail::check(p252_comm_check(comm_.get(), stream), ctx_.get(), "Comm::check"); }
    // the RCCL shared object the library resolved for this process at run time (p252_comm_backend)
    static std::string backend() {
        char path[4096] = {0};
        detail::check(p252_comm_backend(path, sizeof path), nullptr, "p252_comm_backend");
        return path;
    }
    p252_comm* get() const { return comm_.get(); }

  private:
    Comm(Context& ctx, p252_comm* c) : ctx_(ctx) { comm_.reset(c, p252_comm_destroy); }
    Context ctx_;  // (shared ownership: the context outlives its communicator)
    std::shared_ptr<p252_comm> comm_;
};

// Roots of the leaves.size() / leaves_per_tree independent complete trees stored tree-major in HOST memory
// (p252_merkle4_forest: the first level is hashed while the leaves stream in through the staging lanes, the upper levels once)
inline std::vector<BlsScalar> merkle4_forest(const std::vector<BlsScalar>& leaves, std::size_t leaves_per_tree,
                                             Context& ctx = Context::default_context()) {
    if (leaves_per_tree == 0 || leaves.size() % leaves_per_tree) throw std::invalid_argument("merkle4_forest: not whole trees");
    const BlsScalar tag = compute_tag(Domain::Merkle4, {4}, 1);
    std::vector<BlsScalar> roots(leaves.size() / leaves_per_tree);
    if (!roots.empty())
        detail::check(p252_merkle4_forest(ctx.get(), tag.data(), leaves[0].data(), roots.size(), leaves_per_tree, roots[0].data()), ctx.get(),
                      "merkle4_forest");
    return roots;
}

// A forest of roots.size() independent complete trees of leaves_per_tree = 4^k device-resident leaves each (tree-major):
// one launch per level across all trees (p252_merkle4_forest_device); d_roots receives n_trees scalars.
inline void merkle4_forest_device(const void* d_leaves, std::size_t n_trees, std::size_t leaves_per_tree, void* d_roots,
                                  Context& ctx = Context::default_context(), void* d_levels = nullptr, void* stream = nullptr) {
    const BlsScalar tag = compute_tag(Domain::Merkle4, {4}, 1);
    detail::check(p252_merkle4_forest_device(ctx.get(), tag.data(), d_leaves, n_trees, leaves_per_tree, d_roots, d_levels, stream), ctx.get(),
                  "merkle4_forest_device");
}

// `Opening::verify` of the downstream poseidon-merkle consumer (AGENTS.md:62-66) for n device-resident arity-4 openings against ONE
// root: d_ok[i] = 1 iff opening i re-hashes to *d_root (p252_merkle4_verify_batch_device); layouts as p252_merkle4_path_batch_device.
inline void merkle4_verify_batch_device(const void* d_leaves, const void* d_siblings, const void* d_positions, std::size_t depth,
                                        const void* d_root, void* d_ok, std::size_t n, Context& ctx = Context::default_context(),
                                        void* stream = nullptr) {
    const BlsScalar tag = compute_tag(Domain::Merkle4, {4}, 1);
    detail::check(p252_merkle4_verify_batch_device(ctx.get(), tag.data(), d_leaves, d_siblings, d_positions, depth, d_root, d_ok, n, stream),
                  ctx.get(), "merkle4_verify_batch_device");
}

// ---- dusk_poseidon::encrypt / decrypt (src/encryption.rs:62-95), batched; `variant` = P252_CRYPT_STREAM (default) or
// P252_CRYPT_DUPLEX — the construction is UNPINNED.  secrets[i] = {shared.get_u(), shared.get_v()}. ----
struct DecryptionFailed : std::runtime_error {  // dusk_poseidon::Error::DecryptionFailed (src/error.rs:27-29)
    DecryptionFailed() : std::runtime_error("DecryptionFailed") {}
};
inline BlsScalar encryption_tag(std::size_t message_len, int variant = P252_CRYPT_STREAM) {
    BlsScalar t{};
    detail::check(p252_encryption_tag(variant, message_len, t.data()), nullptr, "encryption_tag");
    return t;
}
inline std::vector<BlsScalar> encrypt_batch(const std::vector<BlsScalar>& messages, std::size_t message_len,
                                            const std::vector<BlsScalar>& secrets, const std::vector<BlsScalar>& nonces,
                                            int variant = P252_CRYPT_STREAM, Context& ctx = Context::default_context()) {
    const std::size_t n = nonces.size();
    if (messages.size() != n * message_len || secrets.size() != 2 * n) throw std::invalid_argument("encrypt_batch: sizes");
    const BlsScalar tag = encryption_tag(message_len, variant);
    std::vector<BlsScalar> out(n * (message_len + 1));
    if (n)
        detail::check(p252_encrypt_batch(ctx.get(), variant, tag.data(), messages[0].data(), secrets[0].data(), nonces[0].data(), message_len,
                                         out[0].data(), n),
                      ctx.get(), "encrypt_batch");
    return out;
}
// returns the messages; ok[i] == 0 marks an item whose MAC did not verify (its message is unspecified)
inline std::vector<BlsScalar> decrypt_batch(const std::vector<BlsScalar>& ciphers, std::size_t message_len,
                                            const std::vector<BlsScalar>& secrets, const std::vector<BlsScalar>& nonces,
                                            std::vector<std::uint8_t>& ok, int variant = P252_CRYPT_STREAM,
                                            Context& ctx = Context::default_context()) {
    const std::size_t n = nonces.size();
    if (ciphers.size() != n * (message_len + 1) || secrets.size() != 2 * n) throw std::invalid_argument("decrypt_batch: sizes");
    const BlsScalar tag = encryption_tag(message_len, variant);
    std::vector<BlsScalar> out(n * message_len);
    ok.assign(n, 0);
    if (n)
        detail::check(p252_decrypt_batch(ctx.get(), variant, tag.data(), ciphers[0].data(), secrets[0].data(), nonces[0].data(), message_len,
                                         out[0].data(), ok.data(), n),
                      ctx.get(), "decrypt_batch");
    return out;
}
// single message, the reference's call shape: throws DecryptionFailed like decrypt() returns Err
inline std::vector<BlsScalar> decrypt(const std::vector<BlsScalar>& cipher, const BlsScalar& secret_u, const BlsScalar& secret_v,
                                      const BlsScalar& nonce, int variant = P252_CRYPT_STREAM, Context& ctx = Context::default_context()) {
    if (cipher.size() < 2) throw IoPatternError(IoPatternError::InvalidIOPattern, "decrypt: empty message");
    std::vector<std::uint8_t> ok;
    auto m = decrypt_batch(cipher, cipher.size() - 1, {secret_u, secret_v}, {nonce}, ok, variant, ctx);
    if (!ok[0]) throw DecryptionFailed();
    return m;
}

// ---- the canonical byte format: BlsScalar::to_bytes / from_bytes (src/hades/round_constants.rs:66-67) ----
using ScalarBytes = std::array<std::uint8_t, 32>;  // little-endian bytes of the canonical value
inline std::vector<ScalarBytes> to_bytes(const std::vector<BlsScalar>& scalars) {
    std::vector<ScalarBytes> out(scalars.size());
    if (!scalars.empty()) detail::check(p252_to_bytes(scalars[0].data(), out[0].data(), scalars.size()), nullptr, "to_bytes");
    return out;
}
// ok[i] == 0 where the value is not below the modulus (BlsScalar::from_bytes returns an error there)
inline std::vector<BlsScalar> from_bytes(const std::vector<ScalarBytes>& bytes, std::vector<std::uint8_t>& ok) {
    std::vector<BlsScalar> out(bytes.size());
    ok.assign(bytes.size(), 0);
    if (!bytes.empty()) detail::check(p252_from_bytes(bytes[0].data(), out[0].data(), ok.data(), bytes.size()), nullptr, "from_bytes");
    return out;
}

}  // namespace dusk_poseidon_hip
