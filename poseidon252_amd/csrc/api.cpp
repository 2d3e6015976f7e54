// api.cpp — C ABI of libposeidon252_hip.so (include/poseidon252_hip.h).
// Host side only: context / device-memory management, io-pattern rules mirroring src/hash.rs,
// launch sequencing.  All hashing runs in kernels.hip; there is no CPU path.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/poseidon252_hip.h"
#include "blake2b.hpp"
#include "ctx.hpp"
#include "hades29.hpp"
#include "kernels.h"
#include "openings.h"
#include "tables.hpp"
#include "_gen/assets.inc"

using namespace p252;
using namespace p252host;

static std::string g_create_err;
static std::mutex g_mu;

int p252host::fail(p252_ctx* ctx, int code, const std::string& msg) {
    if (ctx)
        ctx->err = msg;
    else {
        std::lock_guard<std::mutex> lk(g_mu);
        g_create_err = msg;
    }
    return code;
}

int p252host::ensure(p252_ctx* ctx, void** buf, size_t* cap, size_t need) {
    if (need <= *cap) return P252_OK;
    if (*buf) HIP_TRY(ctx, hipFree(*buf));
    *buf = nullptr;
    *cap = 0;
    HIP_TRY(ctx, hipMalloc(buf, need));
    *cap = need;
    return P252_OK;
}

int p252host::level_set(p252_ctx* ctx, hipStream_t st, size_t need0, size_t need1, p252_ctx::LevelSet** out) {
    p252_ctx::LevelSet* set = nullptr;
    for (auto& s : ctx->lvl)
        if (s.st == st) set = &s;
    if (!set && ctx->lvl.size() < p252_ctx::MAX_LEVEL_SETS) {
        if (ctx->lvl.capacity() < p252_ctx::MAX_LEVEL_SETS) ctx->lvl.reserve(p252_ctx::MAX_LEVEL_SETS);  // (pointers into it stay valid)
        ctx->lvl.emplace_back();
        set = &ctx->lvl.back();
        set->st = st;
        const hipError_t e = hipEventCreateWithFlags(&set->done, hipEventDisableTiming);
        if (e != hipSuccess) {  // never leave a half-made pair (done == nullptr) in the list
            ctx->lvl.pop_back();
            return fail(ctx, P252_ERR_HIP, std::string("hipEventCreateWithFlags: ") + hipGetErrorString(e));
        }
    }
    if (!set) {  // more streams than pairs: take over the least recently used pair, behind its last build
        set = &ctx->lvl[0];
        for (auto& s : ctx->lvl)
            if (s.stamp < set->stamp) set = &s;
        if (set->stamp) HIP_TRY(ctx, hipStreamWaitEvent(st, set->done, 0));
        set->st = st;
    }
    // (growing frees the old buffer: hipFree waits for the device, so no queued launch still reads it)
    int rc = ensure(ctx, &set->buf[0], &set->cap[0], need0);
    if (rc == P252_OK) rc = ensure(ctx, &set->buf[1], &set->cap[1], need1);
    if (rc) return rc;
    *out = set;
    return P252_OK;
}

int p252host::level_set_done(p252_ctx* ctx, p252_ctx::LevelSet* set) {
    set->stamp = ++ctx->lvl_clock;
    HIP_TRY(ctx, hipEventRecord(set->done, set->st));
    return P252_OK;
}

namespace {
// Records the pair's event on EVERY way out of a builder — also when a launch failed half way: the launches before it are queued and
// touch the pair, and a later takeover by another stream waits on this event (the early returns used to skip it and
// left the event of an OLDER build for the next takeover).  finish() on the success path returns the record's own status.
struct LevelSetGuard {
    p252_ctx* ctx;
    p252_ctx::LevelSet* set = nullptr;
    explicit LevelSetGuard(p252_ctx* c) : ctx(c) {}
    int finish() {
        p252_ctx::LevelSet* s = set;
        set = nullptr;
        return s ? p252host::level_set_done(ctx, s) : P252_OK;
    }
    ~LevelSetGuard() {
        if (!set) return;
        const std::string msg = ctx->err;  // (the failure being reported stays the message)
        (void)p252host::level_set_done(ctx, set);
        ctx->err = msg;
    }
};
}  // namespace

const std::vector<int32_t>& p252host::host_tables() {
    static const std::vector<int32_t> tab = [] {
        HadesTables T;
        derive_tables(ARC_BIN, MDS_BIN, T);
        // the kernels' schedule rests on mds.bin being R/(i+j+5) (tables.hpp): never run it on anything else
        return T.int_ok ? encode_tables29(T) : std::vector<int32_t>();
    }();
    return tab;
}

// device-resident scalar arrays are read and written with 16-byte accesses (a BlsScalar array from hipMalloc, or any
// 32-byte-multiple offset into one, qualifies); anything else would fault on the GPU, so it is refused here
static bool misaligned(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; }
static const char* const ALIGN_MSG = "device scalar arrays must be 16-byte aligned";

// kernel argument for the tag: the scalar itself and lane 0's first-round S-box output (the digest kernels start there)
static TagArg tag_arg(const uint64_t tag[4]) {
    TagArg t;
    std::memcpy(t.w, tag, 32);
    const E29 x0 = hades_pre0(from_mont4(t.w), host_tables().data());
    for (int k = 0; k < NL; ++k) t.x0[k] = x0.d[k];
    return t;
}

// runs f(t) for every context index on its own host thread (t = 0 on the calling thread); first failure wins
template <class F>
static int for_each_ctx(p252_ctx* const* ctxs, size_t n_ctx, F&& f) {
    std::vector<int> rcs(n_ctx, P252_OK);
    std::vector<std::thread> pool;
    for (size_t t = 1; t < n_ctx; ++t) pool.emplace_back([&, t] { rcs[t] = f(t); });
    rcs[0] = f(0);
    for (auto& th : pool) th.join();
    for (size_t t = 0; t < n_ctx; ++t)
        if (rcs[t] != P252_OK) {
            if (t != 0 && ctxs[0]) ctxs[0]->err = "context " + std::to_string(t) + ": " + ctxs[t]->err;  // p252_last_error(ctxs[0])
            return rcs[t];
        }
    return P252_OK;
}

// ---- secret hygiene (the reference builds its scalars with `zeroize`, Cargo.toml:14, and dusk-safe zeroizes the sponge state
// when a sponge finishes; src/encryption.rs:62-95 hands shared secrets, nonces and plaintexts through it).  The kernels keep
// every sponge state in registers; what outlives a HOST-buffer encrypt / decrypt call is the library's own copy of the
// caller's arrays: the context's device scratch and the page-locked staging lanes.  wipe_span / wipe_lanes clear exactly
// those at the end of such a call; wipe_all clears every buffer the context owns (p252_wipe, p252_destroy). ----
static hipError_t wipe_span(void* d, size_t bytes) { return (d && bytes) ? hipMemsetAsync(d, 0, bytes, nullptr) : hipSuccess; }

// dirty_only: just the bytes calls may have written since the last wipe (staged_run keeps the extent per slot) — what the per-call
// wipe of a host-buffer encrypt / decrypt needs; the full capacity otherwise (p252_wipe, p252_trim, p252_destroy)
static hipError_t wipe_lanes(p252_ctx* ctx, bool dirty_only = false) {
    hipError_t first = hipSuccess;
    for (auto& l : ctx->lanes)
        for (auto& sl : l.slot) {
            const size_t in_b = dirty_only ? sl.in_dirty : sl.in_cap, out_b = dirty_only ? sl.out_dirty : sl.out_cap;
            if (sl.h_in && in_b) std::memset(sl.h_in, 0, in_b);
            if (sl.h_out && out_b) std::memset(sl.h_out, 0, out_b);
            hipError_t e = wipe_span(sl.d_in, in_b);
            if (e == hipSuccess) e = wipe_span(sl.d_out, out_b);
            if (e != hipSuccess && first == hipSuccess) first = e;
            sl.in_dirty = sl.out_dirty = 0;
        }
    return first;
}

static hipError_t wipe_all(p252_ctx* ctx) {
    hipError_t e = hipDeviceSynchronize();  // nothing queued may still read or write what is cleared below
    auto keep = [&](hipError_t r) {
        if (r != hipSuccess && e == hipSuccess) e = r;
    };
    keep(wipe_span(ctx->d_in, ctx->d_in_cap));
    keep(wipe_span(ctx->d_out, ctx->d_out_cap));
    for (auto& set : ctx->lvl)
        for (int i = 0; i < 2; ++i) keep(wipe_span(set.buf[i], set.cap[i]));
    keep(wipe_span(ctx->d_prog, ctx->d_prog_cap));
    ctx->prog_variant = -1;  // (the call table is gone: the next encrypt / decrypt uploads it again)
    ctx->prog_len = 0;
    keep(wipe_lanes(ctx));
    keep(hipStreamSynchronize(nullptr));
    return e;
}

// non-zero bytes of a device buffer, read back 16 MiB at a time (diagnostics only)
static hipError_t count_nonzero_device(const void* d, size_t bytes, uint64_t* total) {
    if (!d || !bytes) return hipSuccess;
    std::vector<uint64_t> host((size_t)2 << 20);
    for (size_t off = 0; off < bytes; off += host.size() * 8) {
        const size_t cnt = bytes - off < host.size() * 8 ? bytes - off : host.size() * 8;
        host[(cnt - 1) / 8] = 0;
        const hipError_t e = hipMemcpy(host.data(), static_cast<const char*>(d) + off, cnt, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return e;
        const unsigned char* b = reinterpret_cast<const unsigned char*>(host.data());
        for (size_t i = 0; i < cnt; ++i) *total += b[i] != 0;
    }
    return hipSuccess;
}

extern "C" {

int p252_abi_version(void) { return P252_ABI_VERSION; }

const char* p252_version(void) { return "poseidon252_hip 0.8 (gfx950; 9x29-bit limbs; integer MDS + integer ARMA recurrence; 32-bit Montgomery quotient digits; lane-group kernels for small batches)"; }

int p252_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int p252_create(int device_id, p252_ctx** out) {
    if (!out) return fail(nullptr, P252_ERR_INVALID_ARGUMENT, "p252_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0)
        return fail(nullptr, P252_ERR_NO_DEVICE,
                    std::string("no HIP device available (") + hipGetErrorString(e) +
                        "); this library has no CPU fallback");
    if (device_id < 0 || device_id >= n) return fail(nullptr, P252_ERR_INVALID_ARGUMENT, "p252_create: bad device id");
    p252_ctx* ctx = new p252_ctx();
    ctx->device = device_id;
    ctx->h_tab = host_tables();
    if (ctx->h_tab.empty()) {
        delete ctx;
        return fail(nullptr, P252_ERR_INVALID_ARGUMENT, "p252_create: the built-in mds.bin is not the Cauchy matrix R/(i+j+5) the kernels' schedule requires");
    }
    hipError_t e2 = hipSetDevice(device_id);
    if (e2 == hipSuccess) e2 = hipMalloc((void**)&ctx->d_tab, ctx->h_tab.size() * sizeof(int32_t));
    if (e2 == hipSuccess)
        e2 = hipMemcpy(ctx->d_tab, ctx->h_tab.data(), ctx->h_tab.size() * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e2 != hipSuccess) {
        std::string msg = std::string("p252_create: ") + hipGetErrorString(e2);
        delete ctx;
        return fail(nullptr, P252_ERR_HIP, msg);
    }
    *out = ctx;
    return P252_OK;
}

// gives back what the grow-only scratch has grown to (after a 2^32-leaf root-only build the context kept 40 GiB
// until p252_destroy).  Wiped first, like p252_wipe; the constant table, the communicator and its 100 bytes stay.
static void free_scratch(p252_ctx* ctx) {
    if (ctx->d_in) (void)hipFree(ctx->d_in);
    if (ctx->d_out) (void)hipFree(ctx->d_out);
    ctx->d_in = ctx->d_out = nullptr;
    ctx->d_in_cap = ctx->d_out_cap = 0;
    for (auto& s : ctx->lvl) {
        for (int i = 0; i < 2; ++i)
            if (s.buf[i]) (void)hipFree(s.buf[i]);
        if (s.done) (void)hipEventDestroy(s.done);
    }
    ctx->lvl.clear();
    if (ctx->d_prog) (void)hipFree(ctx->d_prog);
    ctx->d_prog = nullptr;
    ctx->d_prog_cap = 0;
    ctx->prog_variant = -1;
    ctx->prog_len = 0;
    for (auto& l : ctx->lanes) {
        if (l.st) (void)hipStreamDestroy(l.st);
        for (auto& sl : l.slot) {
            if (sl.h_in) (void)hipHostFree(sl.h_in);
            if (sl.h_out) (void)hipHostFree(sl.h_out);
            if (sl.d_in) (void)hipFree(sl.d_in);
            if (sl.d_out) (void)hipFree(sl.d_out);
            if (sl.done) (void)hipEventDestroy(sl.done);
        }
    }
    ctx->lanes.clear();
}

void p252_destroy(p252_ctx* ctx) {
    if (!ctx) return;
    release_ctx_comm(ctx);
    (void)hipSetDevice(ctx->device);
    (void)wipe_all(ctx);  // nothing a call left in the context's scratch or staging survives it (zeroize, Cargo.toml:14)
    if (ctx->d_tab) (void)hipFree(ctx->d_tab);
    free_scratch(ctx);
    for (int i = 0; i < 3; ++i)
        if (ctx->streams[i]) (void)hipStreamDestroy(ctx->streams[i]);
    delete ctx;
}

const char* p252_last_error(const p252_ctx* ctx) {
    if (ctx) return ctx->err.c_str();
    std::lock_guard<std::mutex> lk(g_mu);
    return g_create_err.c_str();
}

int p252_sync(p252_ctx* ctx, void* hip_stream) {
    if (!ctx) return P252_ERR_INVALID_ARGUMENT;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize((hipStream_t)hip_stream));
    return comm_take_failure(ctx);  // a peer's failed local build in a sharded tree on this stream surfaces here (comm.cpp)
}

int p252_trim(p252_ctx* ctx) {
    if (!ctx) return P252_ERR_INVALID_ARGUMENT;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, wipe_all(ctx));  // (starts with a device synchronisation: nothing queued still uses what is freed below)
    free_scratch(ctx);
    return P252_OK;
}

int p252_wipe(p252_ctx* ctx) {
    if (!ctx) return P252_ERR_INVALID_ARGUMENT;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, wipe_all(ctx));
    return P252_OK;
}

int p252_scratch_residue(p252_ctx* ctx, uint64_t* nonzero_bytes) {
    if (!ctx || !nonzero_bytes) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "scratch_residue: NULL argument");
    *nonzero_bytes = 0;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipDeviceSynchronize());
    HIP_TRY(ctx, count_nonzero_device(ctx->d_in, ctx->d_in_cap, nonzero_bytes));
    HIP_TRY(ctx, count_nonzero_device(ctx->d_out, ctx->d_out_cap, nonzero_bytes));
    for (auto& set : ctx->lvl)
        for (int i = 0; i < 2; ++i) HIP_TRY(ctx, count_nonzero_device(set.buf[i], set.cap[i], nonzero_bytes));
    // (d_prog, the sponge-call table of the last (variant, length), holds nothing of the caller's: not counted)
    for (auto& l : ctx->lanes)
        for (auto& sl : l.slot) {
            HIP_TRY(ctx, count_nonzero_device(sl.d_in, sl.in_cap, nonzero_bytes));
            HIP_TRY(ctx, count_nonzero_device(sl.d_out, sl.out_cap, nonzero_bytes));
            for (size_t i = 0; sl.h_in && i < sl.in_cap; ++i) *nonzero_bytes += static_cast<const unsigned char*>(sl.h_in)[i] != 0;
            for (size_t i = 0; sl.h_out && i < sl.out_cap; ++i) *nonzero_bytes += static_cast<const unsigned char*>(sl.h_out)[i] != 0;
        }
    return P252_OK;
}

// ------------------------------------------------------------------------------------------
// device-buffer entry points
// ------------------------------------------------------------------------------------------
int p252_permute_batch_device(p252_ctx* ctx, const void* d_states, void* d_out, size_t n, void* hip_stream) {
    if (!ctx) return P252_ERR_INVALID_ARGUMENT;
    if (n == 0) return P252_OK;
    if (!d_states || !d_out) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "permute: NULL buffer");
    if (misaligned(d_states) || misaligned(d_out)) return fail(ctx, P252_ERR_INVALID_ARGUMENT, ALIGN_MSG);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_permute(ctx->d_tab, d_states, d_out, n, (hipStream_t)hip_stream));
    return P252_OK;
}

// trunc250: Hash::finalize_truncated (hash.rs:164-183) — the digest kernels' output stage canonicalises, masks to 250 bits and
// stores the raw limbs JubJubScalar::from_raw receives: one launch, no BlsScalar round trip through HBM
static int hash_batch_device_impl(p252_ctx* ctx, const uint64_t tag[4], const void* d_in, size_t in_len,
                                  size_t out_len, void* d_out, size_t n, void* hip_stream, bool trunc250) {
    if (!ctx) return P252_ERR_INVALID_ARGUMENT;
    // dusk-safe rejects io-patterns with an empty absorb or squeeze -> Hash::finalize panics (hash.rs:134-137)
    if (in_len == 0 || out_len == 0)
        return fail(ctx, P252_ERR_INVALID_IO_PATTERN, "hash: in_len and out_len must be > 0");
    if (in_len > 0x7fffffffu || out_len > 0x7fffffffu)
        return fail(ctx, P252_ERR_INVALID_ARGUMENT, "hash: length too large");
    if (n == 0) return P252_OK;
    if (!tag || !d_in || !d_out) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "hash: NULL buffer");
    if (misaligned(d_in) || misaligned(d_out)) return fail(ctx, P252_ERR_INVALID_ARGUMENT, ALIGN_MSG);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)hip_stream;
    if (in_len == 4 && out_len == 1)
        HIP_TRY(ctx, launch_merkle4(ctx->d_tab, tag_arg(tag), d_in, 4 * n, d_out, n, st, 4, 0, trunc250));
    else if (in_len == 2 && out_len == 1)  // Merkle2-shaped digests take the single-permutation kernel too
        HIP_TRY(ctx, launch_merkle4(ctx->d_tab, tag_arg(tag), d_in, 2 * n, d_out, n, st, 2, 0, trunc250));
    else
        HIP_TRY(ctx, launch_sponge(ctx->d_tab, tag_arg(tag), d_in, (unsigned)in_len, (unsigned)out_len, d_out, n, st, trunc250));
    return P252_OK;
}

int p252_hash_batch_device(p252_ctx* ctx, const uint64_t tag[4], const void* d_in, size_t in_len,
                           size_t out_len, void* d_out, size_t n, void* hip_stream) {
    return hash_batch_device_impl(ctx, tag, d_in, in_len, out_len, d_out, n, hip_stream, false);
}

int p252_hash_batch_truncated_device(p252_ctx* ctx, const uint64_t tag[4], const void* d_in, size_t in_len,
                                     size_t out_len, void* d_out_raw, size_t n, void* hip_stream) {
    return hash_batch_device_impl(ctx, tag, d_in, in_len, out_len, d_out_raw, n, hip_stream, true);
}

static size_t levels_len(size_t n_leaves, size_t arity) {
    size_t total = 0, c = n_leaves;
    while (c > 1) {
        c = (c + arity - 1) / arity;
        total += c;
    }
    return total;
}
size_t p252_merkle4_levels_len(size_t n_leaves) { return levels_len(n_leaves, 4); }
size_t p252_merkle2_levels_len(size_t n_leaves) { return levels_len(n_leaves, 2); }

extern "C++" {
int p252host::merkle_tree_device(p252_ctx* ctx, unsigned arity, const uint64_t tag[4], const void* d_leaves, size_t n_leaves,
                              void* d_root, void* d_levels, void* hip_stream) {
    if (!ctx) return P252_ERR_INVALID_ARGUMENT;
    if (n_leaves == 0) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "merkle_tree: n_leaves must be > 0");
    if (!tag || !d_leaves || !d_root) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "merkle_tree: NULL buffer");
    if (misaligned(d_leaves) || misaligned(d_levels)) return fail(ctx, P252_ERR_INVALID_ARGUMENT, ALIGN_MSG);  // d_root: plain copy
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)hip_stream;
    const TagArg t = tag_arg(tag);
    const char* cur = static_cast<const char*>(d_leaves);
    size_t cur_n = n_leaves;
    char* lv = static_cast<char*>(d_levels);
    LevelSetGuard guard(ctx);
    p252_ctx::LevelSet*& set = guard.set;
    if (!d_levels && n_leaves > 1) {  // ping-pong in context-owned scratch: the pair of THIS stream (ctx.hpp)
        const size_t l1 = (n_leaves + arity - 1) / arity, l2 = (l1 + arity - 1) / arity;
        int rc = level_set(ctx, st, l1 * 32, l2 * 32, &set);
        if (rc) return rc;
    }
    // A build with wide levels (more nodes than the chip has lanes at one wave per SIMD: 65,536) keeps every SIMD
    // loaded through its narrow levels by computing them redundantly (kernels.hip k_merkle4_pad: the next wide launch
    // would otherwise start at a dipped clock).  P252_TREE_PAD_LANES overrides the lane count; 0 = off.
    static const long pad_env = [] {
        const char* e = std::getenv("P252_TREE_PAD_LANES");
        return e ? std::atol(e) : -1L;
    }();
    const size_t chip_lanes = 65536;
    const size_t first_level = (n_leaves + arity - 1) / arity;
    const size_t pad = pad_env >= 0 ? (size_t)pad_env : (first_level > chip_lanes ? chip_lanes : 0);
    int parity = 0;
    while (cur_n > 1) {  // a single leaf is its own root: an arity^k-leaf tree costs exactly k levels
        const size_t next_n = (cur_n + arity - 1) / arity;
        char* next = d_levels ? lv : static_cast<char*>(set->buf[parity]);
        HIP_TRY(ctx, launch_merkle4(ctx->d_tab, t, cur, cur_n, next, next_n, st, arity, pad));
        cur = next;
        cur_n = next_n;
        if (d_levels) lv += next_n * 32;
        parity ^= 1;
    }
    HIP_TRY(ctx, hipMemcpyAsync(d_root, cur, 32, hipMemcpyDeviceToDevice, st));
    return guard.finish();
}
}  // extern "C++"

int p252_merkle4_tree_device(p252_ctx* ctx, const uint64_t tag[4], const void* d_leaves, size_t n_leaves,
                             void* d_root, void* d_levels, void* hip_stream) {
    return merkle_tree_device(ctx, 4, tag, d_leaves, n_leaves, d_root, d_levels, hip_stream);
}

// A forest of n_trees independent complete arity-4 trees of leaves_per_tree = 4^k leaves each, tree-major in d_leaves
// (the shape of the downstream poseidon-merkle consumers, AGENTS.md:62-66: many small trees).  Level l of ALL trees is one
// array of n_trees * 4^(k-l) nodes, tree-major — so the forest is the first k levels of the level-by-level reduction of
// the concatenated leaves, and each level is ONE launch across all trees: the narrow upper levels of many small trees
// fill the chip together instead of each paying one wave's latency per tree.
static int forest_device(p252_ctx* ctx, unsigned arity, const uint64_t tag[4], const void* d_leaves, size_t n_trees, size_t leaves_per_tree,
                         void* d_roots, void* d_levels, void* hip_stream) {
    if (!ctx) return P252_ERR_INVALID_ARGUMENT;
    if (n_trees == 0) return P252_OK;
    const bool pow_ok = arity == 4 ? power_of_4(leaves_per_tree) : (leaves_per_tree != 0 && (leaves_per_tree & (leaves_per_tree - 1)) == 0);
    if (!pow_ok) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "merkle_forest: leaves_per_tree must be arity^k");
    if (n_trees > (SIZE_MAX / 32) / leaves_per_tree) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "merkle_forest: size overflow");
    if (!tag || !d_leaves || !d_roots) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "merkle_forest: NULL buffer");
    if (misaligned(d_leaves) || misaligned(d_roots) || misaligned(d_levels)) return fail(ctx, P252_ERR_INVALID_ARGUMENT, ALIGN_MSG);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)hip_stream;
    size_t cur_n = n_trees * leaves_per_tree;
    if (leaves_per_tree == 1) {  // a single leaf is its own root (as in p252_merkle4_tree_device)
        HIP_TRY(ctx, hipMemcpyAsync(d_roots, d_leaves, cur_n * 32, hipMemcpyDeviceToDevice, st));
        return P252_OK;
    }
    LevelSetGuard guard(ctx);
    p252_ctx::LevelSet*& set = guard.set;
    if (!d_levels) {  // ping-pong in context-owned scratch, this stream's pair (the last level goes straight to d_roots)
        int rc = level_set(ctx, st, cur_n / arity * 32, cur_n / arity / arity * 32 + 32, &set);
        if (rc) return rc;
    }
    const TagArg t = tag_arg(tag);
    const char* cur = static_cast<const char*>(d_leaves);
    char* lv = static_cast<char*>(d_levels);
    // narrow levels after wide ones are computed redundantly on every SIMD, as in the tree builder (clock dip)
    const size_t pad = cur_n / arity > 65536 ? 65536 : 0;
    int parity = 0;
    while (cur_n > n_trees) {
        const size_t next_n = cur_n / arity;
        char* next = next_n == n_trees ? static_cast<char*>(d_roots) : (d_levels ? lv : static_cast<char*>(set->buf[parity]));
        HIP_TRY(ctx, launch_merkle4(ctx->d_tab, t, cur, cur_n, next, next_n, st, arity, pad));
        if (d_levels) {
            if (next_n == n_trees) HIP_TRY(ctx, hipMemcpyAsync(lv, d_roots, next_n * 32, hipMemcpyDeviceToDevice, st));  // levels hold the roots too
            lv += next_n * 32;
        }
        cur = next;
        cur_n = next_n;
        parity ^= 1;
    }
    return guard.finish();
}

int p252_merkle4_forest_device(p252_ctx* ctx, const uint64_t tag[4], const void* d_leaves, size_t n_trees, size_t leaves_per_tree,
                               void* d_roots, void* d_levels, void* hip_stream) {
    return forest_device(ctx, 4, tag, d_leaves, n_trees, leaves_per_tree, d_roots, d_levels, hip_stream);
}

int p252_merkle2_forest_device(p252_ctx* ctx, const uint64_t tag[4], const void* d_leaves, size_t n_trees, size_t leaves_per_tree,
                               void* d_roots, void* d_levels, void* hip_stream) {
    return forest_device(ctx, 2, tag, d_leaves, n_trees, leaves_per_tree, d_roots, d_levels, hip_stream);
}

int p252_merkle2_tree_device(p252_ctx* ctx, const uint64_t tag[4], const void* d_leaves, size_t n_leaves,
                             void* d_root, void* d_levels, void* hip_stream) {
    return merkle_tree_device(ctx, 2, tag, d_leaves, n_leaves, d_root, d_levels, hip_stream);
}

// ------------------------------------------------------------------------------------------
// host-buffer entry points (synchronous)
// ------------------------------------------------------------------------------------------
static bool is_pinned(const void* p) {
    hipPointerAttribute_t a;
    const bool is = hipPointerGetAttributes(&a, p) == hipSuccess && a.type == hipMemoryTypeHost;
    (void)hipGetLastError();
    return is;
}

// Pageable caller memory (a Rust Vec<BlsScalar>, a numpy array): page-locking it per call costs as much as the transfer
// (round 1: 1.9e8 digests/s against 4.0e8 from pinned memory).  So a large host-buffer call is split into chunks of items
// handled by a few LANES: a worker thread with its own stream and TWO slots (page-locked staging pair + device pair +
// event).  Per chunk: memcpy in -> H2D -> kernel -> D2H (all asynchronous on the lane's stream) and, one chunk later,
// memcpy out — the host copies of chunk c+1 overlap the DMA and kernel of chunk c, the lanes overlap each other, and
// nothing the caller owns is ever registered.  Events are blocking-sync: a waiting worker sleeps instead of spinning (the
// benchmark box grants 16 CPUs; spinning workers get the whole process throttled).  The host copies themselves are not
// the limit (EPYC 9575F: memcpy 30 GB/s per thread, bench_tools/ntcopy.cpp; non-temporal stores changed nothing end to
// end) — in-flight depth is: round 2's first version (one slot per lane, spinning waits, 8-12 lanes) reached 2.0-3.1e8.
extern "C++" {  // (templates below; the enclosing block is extern "C")
struct HostSpan {  // one per-item array of a batched call: item i occupies bytes [i * stride, (i + 1) * stride)
    const char* src;  // input arrays: caller memory to read
    char* dst;        // output arrays: caller memory to write
    size_t stride;
};

// CPUs this process may actually use: min(affinity mask, cgroup quota — v2 cpu.max, else v1 cfs_quota_us / cfs_period_us,
// as bench.py's usable_cpus() reads them).  The benchmark box shows 256 logical CPUs and grants 16.
static double cpu_budget() {
    static const double cpus = [] {
        double c = (double)std::thread::hardware_concurrency();
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) c = (double)CPU_COUNT(&set);
        if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota> <period>" or "max <period>"
            char q[32];
            double period = 0;
            if (std::fscanf(f, "%31s %lf", q, &period) == 2 && std::strcmp(q, "max") != 0 && period > 0) {
                const double quota = std::atof(q) / period;
                if (quota > 0 && quota < c) c = quota;
            }
            std::fclose(f);
        } else {  // cgroup v1: quota in microseconds per period, -1 = unlimited
            double quota = -1, period = 0;
            if (FILE* fq = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
                if (std::fscanf(fq, "%lf", &quota) != 1) quota = -1;
                std::fclose(fq);
            }
            if (FILE* fp = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (std::fscanf(fp, "%lf", &period) != 1) period = 0;
                std::fclose(fp);
            }
            if (quota > 0 && period > 0 && quota / period < c) c = quota / period;
        }
        return c < 1 ? 1.0 : c;
    }();
    return cpus;
}

static int staging_lanes_env() {  // P252_HOST_LANES: 0 = not set
    static const int lanes = [] {
        if (const char* e = std::getenv("P252_HOST_LANES")) {
            const int v = std::atoi(e);
            return v < 1 ? 1 : (v > 32 ? 32 : v);
        }
        return 0;
    }();
    return lanes;
}

// staging lanes of ONE context driven on its own
static int staging_lanes_wanted() {
    if (staging_lanes_env()) return staging_lanes_env();
    // Three lanes x two slots keep the PCIe link busy; MORE lanes are slower on the benchmark box (16 CPUs of cgroup
    // budget, 256 visible): 3 lanes 3.6-3.8e8 digests/s, 6 lanes 3.1-3.3e8, 12 lanes 2.0-2.2e8
    // (profiles/r02_host_path.txt).  Two when the process may use fewer than four CPUs.
    return cpu_budget() < 4 ? 2 : 3;
}

// staging lanes per context when n_ctx contexts are driven at once (p252_*_multi: one driver thread per context, which
// is that context's first lane, plus its extra lanes): the contexts SHARE the CPU budget — 8 contexts x 3 lanes under a
// 16-CPU quota is the 24-worker configuration the single-context sweep measured at half speed.
// A lane is host-memcpy work: ONE worker moves 1.66e8 digests/s (26.5 GB/s, profiles/r03_host_path_multi.txt: one context,
// one lane), a GPU takes 4.1e8, so a GPU wants 2.5 workers and the whole budget is worth using: lanes = floor(CPUs / n_ctx),
// at most the single-context optimum of 3, at least 1 — never more workers than CPUs (they sleep on blocking-sync events
// while their chunks are in flight; only the copies cost CPU time).
static int staging_lanes_per_ctx(size_t n_ctx) {
    if (staging_lanes_env()) return staging_lanes_env();
    if (n_ctx <= 1) return staging_lanes_wanted();
    int per = (int)(cpu_budget() / (double)n_ctx);
    return per < 1 ? 1 : (per > 3 ? 3 : per);
}

// Runs `launch(d_in[], d_out[], first_item, count, stream)` over n items in chunks of `chunk`, streaming the input arrays
// in and the output arrays out through the staging lanes.  d_in[a] / d_out[a] are the device copies of chunk-local slices
// of ins[a] / outs[a] (256-byte aligned).  `outs` may be empty (results stay on the device: the launch writes them itself).
template <class Launch>
static int staged_run(p252_ctx* ctx, size_t n, size_t chunk, const std::vector<HostSpan>& ins, const std::vector<HostSpan>& outs,
                      Launch&& launch) {
    const size_t n_chunks = (n + chunk - 1) / chunk;
    const int lanes_wanted = ctx->lane_budget > 0 ? ctx->lane_budget : staging_lanes_wanted();
    const int n_lanes = (int)(n_chunks < (size_t)lanes_wanted ? n_chunks : (size_t)lanes_wanted);
    if ((int)ctx->lanes.size() < n_lanes) ctx->lanes.resize(n_lanes);
    auto layout = [&](const std::vector<HostSpan>& spans, std::vector<size_t>& offs) {  // sub-buffer offsets, total bytes
        size_t total = 0;
        for (const HostSpan& sp : spans) {
            offs.push_back(total);
            total += (chunk * sp.stride + 255) & ~(size_t)255;
        }
        return total ? total : (size_t)256;
    };
    std::vector<size_t> in_off, out_off;
    const size_t in_chunk_b = layout(ins, in_off), out_chunk_b = layout(outs, out_off);
    for (int l = 0; l < n_lanes; ++l) {
        p252_ctx::Lane& L = ctx->lanes[l];
        if (!L.st) HIP_TRY(ctx, hipStreamCreateWithFlags(&L.st, hipStreamNonBlocking));
        for (auto& S : L.slot) {
            if (!S.done) HIP_TRY(ctx, hipEventCreateWithFlags(&S.done, hipEventBlockingSync | hipEventDisableTiming));
            if (S.in_cap < in_chunk_b) {
                if (S.h_in) (void)hipHostFree(S.h_in);
                if (S.d_in) (void)hipFree(S.d_in);
                S.h_in = S.d_in = nullptr;
                S.in_cap = 0;
                HIP_TRY(ctx, hipHostMalloc(&S.h_in, in_chunk_b, hipHostMallocDefault));
                HIP_TRY(ctx, hipMalloc(&S.d_in, in_chunk_b));
                S.in_cap = in_chunk_b;
            }
            if (S.out_cap < out_chunk_b) {
                if (S.h_out) (void)hipHostFree(S.h_out);
                if (S.d_out) (void)hipFree(S.d_out);
                S.h_out = S.d_out = nullptr;
                S.out_cap = 0;
                HIP_TRY(ctx, hipHostMalloc(&S.h_out, out_chunk_b, hipHostMallocDefault));
                HIP_TRY(ctx, hipMalloc(&S.d_out, out_chunk_b));
                S.out_cap = out_chunk_b;
            }
            if (S.in_dirty < in_chunk_b) S.in_dirty = in_chunk_b;
            if (S.out_dirty < out_chunk_b) S.out_dirty = out_chunk_b;
        }
    }
    std::atomic<size_t> next{0};
    std::atomic<int> status{P252_OK};
    std::mutex err_mu;
    std::string err;
    auto work = [&](int l) {
        p252_ctx::Lane& L = ctx->lanes[l];
        auto bad = [&](const char* what, hipError_t e) {
            std::lock_guard<std::mutex> lk(err_mu);
            if (status.load() == P252_OK) {
                status.store(P252_ERR_HIP);
                err = std::string(what) + ": " + hipGetErrorString(e);
            }
        };
        hipError_t e = hipSetDevice(ctx->device);
        if (e != hipSuccess) return bad("hipSetDevice", e);
        long pending[2] = {-1, -1};  // chunk whose results sit (or will sit) in the slot's h_out
        auto retire = [&](int k) {    // wait for slot k's chunk and hand its outputs to the caller
            if (pending[k] < 0) return true;
            const hipError_t w = hipEventSynchronize(L.slot[k].done);
            if (w != hipSuccess) { bad("event sync", w); return false; }
            const size_t off = (size_t)pending[k] * chunk, cnt = n - off < chunk ? n - off : chunk;
            for (size_t a = 0; a < outs.size(); ++a)
                std::memcpy(outs[a].dst + off * outs[a].stride, static_cast<char*>(L.slot[k].h_out) + out_off[a], cnt * outs[a].stride);
            pending[k] = -1;
            return true;
        };
        std::vector<const void*> d_in(ins.size());
        std::vector<void*> d_out(outs.size());
        int k = 0;
        for (;;) {
            const size_t c = next.fetch_add(1);
            if (c >= n_chunks || status.load() != P252_OK) break;
            if (!retire(k)) return;
            p252_ctx::Slot& S = L.slot[k];
            const size_t off = c * chunk, cnt = n - off < chunk ? n - off : chunk;
            for (size_t a = 0; a < ins.size(); ++a) {
                char* h = static_cast<char*>(S.h_in) + in_off[a];
                char* d = static_cast<char*>(S.d_in) + in_off[a];
                std::memcpy(h, ins[a].src + off * ins[a].stride, cnt * ins[a].stride);
                e = hipMemcpyAsync(d, h, cnt * ins[a].stride, hipMemcpyHostToDevice, L.st);
                if (e != hipSuccess) return bad("H2D", e);
                d_in[a] = d;
            }
            for (size_t a = 0; a < outs.size(); ++a) d_out[a] = static_cast<char*>(S.d_out) + out_off[a];
            e = launch(d_in.data(), d_out.data(), off, cnt, L.st);
            if (e != hipSuccess) return bad("kernel launch", e);
            for (size_t a = 0; a < outs.size(); ++a) {
                e = hipMemcpyAsync(static_cast<char*>(S.h_out) + out_off[a], d_out[a], cnt * outs[a].stride, hipMemcpyDeviceToHost, L.st);
                if (e != hipSuccess) return bad("D2H", e);
            }
            e = hipEventRecord(S.done, L.st);
            if (e != hipSuccess) return bad("event record", e);
            pending[k] = (long)c;
            k ^= 1;
        }
        // drain, older chunk first (slot k holds the older one)
        if (!retire(k)) return;
        retire(k ^ 1);
    };
    std::vector<std::thread> pool;
    for (int l = 1; l < n_lanes; ++l) pool.emplace_back(work, l);
    work(0);
    for (auto& t : pool) t.join();
    if (status.load() != P252_OK) {
        for (int l = 0; l < n_lanes; ++l) (void)hipStreamSynchronize(ctx->lanes[l].st);  // never leave work in flight on the staging buffers
        return fail(ctx, status.load(), err);
    }
    return P252_OK;
}

// items per chunk so that a chunk moves about P252_HOST_CHUNK_MB (default 8) MiB, a multiple of the block size
static size_t staging_chunk_items(size_t bytes_per_item) {
    static const size_t chunk_bytes_target = [] {
        const char* e = std::getenv("P252_HOST_CHUNK_MB");
        const int mb = e ? std::atoi(e) : 8;
        return (size_t)(mb < 1 ? 1 : (mb > 256 ? 256 : mb)) << 20;
    }();
    size_t chunk = chunk_bytes_target / (bytes_per_item ? bytes_per_item : 1);
    if (chunk < 4096) chunk = 4096;
    return chunk & ~(size_t)255;
}

// n sponge hashes through the staging lanes.  d_resident_out != NULL: the outputs stay on the device (item i at
// d_resident_out + i * out_len * 32) — the first level of a tree built from host leaves; `out` is then unused.
static int hash_batch_staged(p252_ctx* ctx, const uint64_t tag[4], const uint64_t* in, size_t in_len, size_t out_len,
                             uint64_t* out, size_t n, size_t chunk, char* d_resident_out = nullptr, bool trunc250 = false) {
    const TagArg targ = tag_arg(tag);
    const bool single = in_len == 4 && out_len == 1, pair = in_len == 2 && out_len == 1;
    std::vector<HostSpan> ins = {{reinterpret_cast<const char*>(in), nullptr, in_len * 32}}, outs;
    if (!d_resident_out) outs.push_back({nullptr, reinterpret_cast<char*>(out), out_len * 32});
    return staged_run(ctx, n, chunk, ins, outs, [&](const void* const* d_in, void* const* d_out, size_t off, size_t cnt, hipStream_t st) {
        void* d_dst = d_resident_out ? static_cast<void*>(d_resident_out + off * out_len * 32) : d_out[0];
        if (single) return launch_merkle4(ctx->d_tab, targ, d_in[0], 4 * cnt, d_dst, cnt, st, 4, 0, trunc250);
        if (pair) return launch_merkle4(ctx->d_tab, targ, d_in[0], 2 * cnt, d_dst, cnt, st, 2, 0, trunc250);
        return launch_sponge(ctx->d_tab, targ, d_in[0], (unsigned)in_len, (unsigned)out_len, d_dst, cnt, st, trunc250);
    });
}
}  // extern "C++"

int p252_permute_batch(p252_ctx* ctx, const uint64_t* states, uint64_t* out, size_t n) {
    if (!ctx) return P252_ERR_INVALID_ARGUMENT;
    if (n == 0) return P252_OK;
    if (!states || !out) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "permute: NULL buffer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t bytes = n * P252_HADES_WIDTH * 32;
    {
        const size_t chunk = staging_chunk_items(P252_HADES_WIDTH * 32);
        if (n >= 2 * chunk && !(is_pinned(states) && is_pinned(out)))  // large pageable batch: through the staging lanes
            return staged_run(ctx, n, chunk, {{reinterpret_cast<const char*>(states), nullptr, P252_HADES_WIDTH * 32}},
                              {{nullptr, reinterpret_cast<char*>(out), P252_HADES_WIDTH * 32}},
                              [&](const void* const* d_in, void* const* d_out, size_t, size_t cnt, hipStream_t st) {
                                  return launch_permute(ctx->d_tab, d_in[0], d_out[0], cnt, st);
                              });
    }
    int rc = ensure(ctx, &ctx->d_in, &ctx->d_in_cap, bytes);
    if (rc) return rc;
    rc = ensure(ctx, &ctx->d_out, &ctx->d_out_cap, bytes);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpy(ctx->d_in, states, bytes, hipMemcpyHostToDevice));
    rc = p252_permute_batch_device(ctx, ctx->d_in, ctx->d_out, n, nullptr);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpy(out, ctx->d_out, bytes, hipMemcpyDeviceToHost));
    return P252_OK;
}


static int hash_batch_host_impl(p252_ctx* ctx, const uint64_t tag[4], const uint64_t* in, size_t in_len, size_t out_len,
                                uint64_t* out, size_t n, bool trunc250) {
    if (!ctx) return P252_ERR_INVALID_ARGUMENT;
    if (in_len == 0 || out_len == 0)
        return fail(ctx, P252_ERR_INVALID_IO_PATTERN, "hash: in_len and out_len must be > 0");
    if (in_len > 0x7fffffffu || out_len > 0x7fffffffu) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "hash: length too large");
    if (n == 0) return P252_OK;
    if (!tag || !in || !out) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "hash: NULL buffer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t in_bytes = n * in_len * 32, out_bytes = n * out_len * 32;
    // P252_HOST_PIPELINE (developer switch): 0 = one serial copy each way, anything else (default) = pipelined
    static const int mode = [] {
        const char* e = std::getenv("P252_HOST_PIPELINE");
        return e ? std::atoi(e) : 2;
    }();
    const size_t chunk = staging_chunk_items((in_len > out_len ? in_len : out_len) * 32);
    if (mode == 0 || n < 2 * chunk) {
        int rc = ensure(ctx, &ctx->d_in, &ctx->d_in_cap, in_bytes);
        if (rc) return rc;
        rc = ensure(ctx, &ctx->d_out, &ctx->d_out_cap, out_bytes);
        if (rc) return rc;
        HIP_TRY(ctx, hipMemcpy(ctx->d_in, in, in_bytes, hipMemcpyHostToDevice));
        rc = hash_batch_device_impl(ctx, tag, ctx->d_in, in_len, out_len, ctx->d_out, n, nullptr, trunc250);
        if (rc) return rc;
        HIP_TRY(ctx, hipMemcpy(out, ctx->d_out, out_bytes, hipMemcpyDeviceToHost));
        return P252_OK;
    }
    // caller memory that is not page-locked on BOTH sides goes through the library's own staging lanes
    if (!is_pinned(in) || !is_pinned(out)) return hash_batch_staged(ctx, tag, in, in_len, out_len, out, n, chunk, nullptr, trunc250);
    // page-locked on both sides (p252_host_alloc / p252_host_register): zero-copy DMA, chunks round-robin over 3
    // streams so that the H2D copy of chunk c+1, the kernel of chunk c and the D2H copy of chunk c-1 overlap
    int rc = ensure(ctx, &ctx->d_in, &ctx->d_in_cap, in_bytes);
    if (rc) return rc;
    rc = ensure(ctx, &ctx->d_out, &ctx->d_out_cap, out_bytes);
    if (rc) return rc;
    for (int i = 0; i < 3; ++i)
        if (!ctx->streams[i]) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->streams[i], hipStreamNonBlocking));
    int status = P252_OK;
    std::string err;
    size_t c = 0;
    for (size_t off = 0; off < n && status == P252_OK; off += chunk, ++c) {
        const size_t cnt = n - off < chunk ? n - off : chunk;
        hipStream_t st = ctx->streams[c % 3];
        const char* h_in = reinterpret_cast<const char*>(in) + off * in_len * 32;
        char* h_out = reinterpret_cast<char*>(out) + off * out_len * 32;
        char* d_in = static_cast<char*>(ctx->d_in) + off * in_len * 32;
        char* d_out = static_cast<char*>(ctx->d_out) + off * out_len * 32;
        hipError_t e = hipMemcpyAsync(d_in, h_in, cnt * in_len * 32, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) {
            int r2 = hash_batch_device_impl(ctx, tag, d_in, in_len, out_len, d_out, cnt, st, trunc250);
            if (r2) { status = r2; err = ctx->err; break; }
            e = hipMemcpyAsync(h_out, d_out, cnt * out_len * 32, hipMemcpyDeviceToHost, st);
        }
        if (e != hipSuccess) { status = P252_ERR_HIP; err = std::string("pipelined copy: ") + hipGetErrorString(e); }
    }
    for (int i = 0; i < 3; ++i) {
        hipError_t e = hipStreamSynchronize(ctx->streams[i]);
        if (e != hipSuccess && status == P252_OK) { status = P252_ERR_HIP; err = std::string("stream sync: ") + hipGetErrorString(e); }
    }
    if (status != P252_OK) return fail(ctx, status, err);
    return P252_OK;
}

int p252_hash_batch(p252_ctx* ctx, const uint64_t tag[4], const uint64_t* in, size_t in_len, size_t out_len,
                    uint64_t* out, size_t n) {
    return hash_batch_host_impl(ctx, tag, in, in_len, out_len, out, n, false);
}

int p252_hash_batch_truncated(p252_ctx* ctx, const uint64_t tag[4], const uint64_t* in, size_t in_len, size_t out_len,
                              uint64_t* out_raw, size_t n) {
    return hash_batch_host_impl(ctx, tag, in, in_len, out_len, out_raw, n, true);
}

static int merkle_tree_host(p252_ctx* ctx, unsigned arity, const uint64_t tag[4], const uint64_t* leaves, size_t n_leaves,
                            uint64_t root[4], uint64_t* levels) {
    if (!ctx) return P252_ERR_INVALID_ARGUMENT;
    if (n_leaves == 0) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "merkle_tree: n_leaves must be > 0");
    if (!tag || !leaves || !root) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "merkle_tree: NULL buffer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t leaf_bytes = n_leaves * 32;
    const size_t lvl_bytes = levels_len(n_leaves, arity) * 32;
    int rc = ensure(ctx, &ctx->d_out, &ctx->d_out_cap, (levels ? lvl_bytes : 0) + 32);
    if (rc) return rc;
    char* d_root = static_cast<char*>(ctx->d_out);
    char* d_levels = levels ? d_root + 32 : nullptr;
    const size_t n_l1 = n_leaves / arity;
    const size_t chunk_nodes = (((size_t)8 << 20) / (arity * 32)) & ~(size_t)255;  // 8 MiB of leaves per chunk
    if (n_leaves % arity == 0 && n_l1 >= 4 * chunk_nodes && !is_pinned(leaves)) {
        // Big tree from pageable host memory (512 MiB of leaves at 2^24): the upload would take longer than the whole build.
        // The first level is hashed chunk by chunk WHILE the leaves stream in through the staging lanes (its nodes stay on
        // the device); the remaining levels run as usual.  The leaves themselves are never resident as a whole.
        rc = ensure(ctx, &ctx->d_in, &ctx->d_in_cap, n_l1 * 32);
        if (rc) return rc;
        char* d_l1 = levels ? d_levels : static_cast<char*>(ctx->d_in);
        rc = hash_batch_staged(ctx, tag, leaves, arity, 1, nullptr, n_l1, chunk_nodes, d_l1);
        if (rc) return rc;
        rc = merkle_tree_device(ctx, arity, tag, d_l1, n_l1, d_root, levels ? d_levels + n_l1 * 32 : nullptr, nullptr);
        if (rc) return rc;
    } else {
        rc = ensure(ctx, &ctx->d_in, &ctx->d_in_cap, leaf_bytes);
        if (rc) return rc;
        HIP_TRY(ctx, hipMemcpy(ctx->d_in, leaves, leaf_bytes, hipMemcpyHostToDevice));
        rc = merkle_tree_device(ctx, arity, tag, ctx->d_in, n_leaves, d_root, d_levels, nullptr);
        if (rc) return rc;
    }
    HIP_TRY(ctx, hipMemcpy(root, d_root, 32, hipMemcpyDeviceToHost));  // (also orders the download below behind the build)
    if (levels && lvl_bytes) {
        const size_t n_sc = lvl_bytes / 32, chunk = staging_chunk_items(32);
        if (n_sc >= 2 * chunk && !is_pinned(levels)) {
            // all levels of a big tree (171 MiB at 2^24 leaves) into pageable memory: chunk by chunk through the staging lanes
            rc = staged_run(ctx, n_sc, chunk, {}, {{nullptr, reinterpret_cast<char*>(levels), 32}},
                            [&](const void* const*, void* const* d_out, size_t off, size_t cnt, hipStream_t st) {
                                return hipMemcpyAsync(d_out[0], d_levels + off * 32, cnt * 32, hipMemcpyDeviceToDevice, st);
                            });
            if (rc) return rc;
        } else {
            HIP_TRY(ctx, hipMemcpy(levels, d_levels, lvl_bytes, hipMemcpyDeviceToHost));
        }
    }
    return P252_OK;
}

int p252_merkle4_tree(p252_ctx* ctx, const uint64_t tag[4], const uint64_t* leaves, size_t n_leaves,
                      uint64_t root[4], uint64_t* levels) {
    return merkle_tree_host(ctx, 4, tag, leaves, n_leaves, root, levels);
}

int p252_merkle2_tree(p252_ctx* ctx, const uint64_t tag[4], const uint64_t* leaves, size_t n_leaves,
                      uint64_t root[4], uint64_t* levels) {
    return merkle_tree_host(ctx, 2, tag, leaves, n_leaves, root, levels);
}

// Forest from HOST leaves (pageable memory is fine).  Large forests: the FIRST level — three quarters of all permutations — is
// hashed chunk by chunk while the leaves stream in through the staging lanes (8 MiB chunks; its nodes stay on the device, the
// leaves are never resident as a whole), then the upper levels run ONCE, one launch per level across all trees, so their
// latency-bound launches are paid once per forest and not once per chunk (a first version that built a whole forest per
// 32-MiB chunk ran at 2.5e8 perm/s: every chunk paid the five narrow levels).  Trees are independent: no exchange.
int p252_merkle4_forest(p252_ctx* ctx, const uint64_t tag[4], const uint64_t* leaves, size_t n_trees, size_t leaves_per_tree, uint64_t* roots) {
    if (!ctx) return P252_ERR_INVALID_ARGUMENT;
    if (n_trees == 0) return P252_OK;
    if (!power_of_4(leaves_per_tree)) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "merkle_forest: leaves_per_tree must be arity^k");
    if (n_trees > (SIZE_MAX / 32) / leaves_per_tree) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "merkle_forest: size overflow");
    if (!tag || !leaves || !roots) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "merkle_forest: NULL buffer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t n_leaves = n_trees * leaves_per_tree, n_l1 = n_leaves / 4;
    const size_t chunk_nodes = (((size_t)8 << 20) / (4 * 32)) & ~(size_t)255;  // 8 MiB of leaves per chunk, as the tree's host path
    int rc = ensure(ctx, &ctx->d_out, &ctx->d_out_cap, n_trees * 32);
    if (rc) return rc;
    if (leaves_per_tree >= 4 && n_l1 >= 4 * chunk_nodes && !is_pinned(leaves)) {
        rc = ensure(ctx, &ctx->d_in, &ctx->d_in_cap, n_l1 * 32);
        if (rc) return rc;
        rc = hash_batch_staged(ctx, tag, leaves, 4, 1, nullptr, n_l1, chunk_nodes, static_cast<char*>(ctx->d_in));
        if (rc) return rc;
        rc = forest_device(ctx, 4, tag, ctx->d_in, n_trees, leaves_per_tree / 4, ctx->d_out, nullptr, nullptr);
    } else {  // small forest (or page-locked leaves: the DMA runs at PCIe speed anyway): one upload, one build
        rc = ensure(ctx, &ctx->d_in, &ctx->d_in_cap, n_leaves * 32);
        if (rc) return rc;
        HIP_TRY(ctx, hipMemcpy(ctx->d_in, leaves, n_leaves * 32, hipMemcpyHostToDevice));
        rc = forest_device(ctx, 4, tag, ctx->d_in, n_trees, leaves_per_tree, ctx->d_out, nullptr, nullptr);
    }
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpy(roots, ctx->d_out, n_trees * 32, hipMemcpyDeviceToHost));
    return P252_OK;
}

// page-locked host memory for callers that want the host-buffer entry points at PCIe speed
void* p252_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}

void p252_host_free(void* p) {
    if (p) (void)hipHostFree(p);
}

int p252_host_register(void* p, size_t bytes) {
    if (!p || bytes == 0) return P252_ERR_INVALID_ARGUMENT;
    if (hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess) {
        (void)hipGetLastError();
        return P252_ERR_HIP;
    }
    return P252_OK;
}

int p252_host_unregister(void* p) {
    if (!p) return P252_ERR_INVALID_ARGUMENT;
    if (hipHostUnregister(p) != hipSuccess) {
        (void)hipGetLastError();
        return P252_ERR_HIP;
    }
    return P252_OK;
}

// ------------------------------------------------------------------------------------------
// SURVEY §8(f) "next" rows: truncated outputs and batched Merkle openings
// ------------------------------------------------------------------------------------------
int p252_truncate250_device(p252_ctx* ctx, const void* d_scalars, void* d_out_raw, size_t n, void* hip_stream) {
    if (!ctx) return P252_ERR_INVALID_ARGUMENT;
    if (n == 0) return P252_OK;
    if (!d_scalars || !d_out_raw) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "truncate250: NULL buffer");
    if (misaligned(d_scalars) || misaligned(d_out_raw)) return fail(ctx, P252_ERR_INVALID_ARGUMENT, ALIGN_MSG);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_truncate250(d_scalars, d_out_raw, n, (hipStream_t)hip_stream));
    return P252_OK;
}

// ---- incremental update of a stored arity-4 tree (the poseidon-merkle consumer changes leaves of a tree it
// keeps): the k leaves are written into d_leaves, then every level re-hashes the (at most k) nodes above them, in place in
// d_levels (layout of p252_merkle4_tree_device: the levels above the leaves, bottom-up); d_root (optional) gets the new root.
static int merkle4_update_device(p252_ctx* ctx, const uint64_t tag[4], void* d_leaves, size_t n_leaves, void* d_levels,
                                 const void* d_indices, const void* d_new_leaves, size_t k, void* d_root, void* d_n_bad, void* hip_stream) {
    if (!ctx) return P252_ERR_INVALID_ARGUMENT;
    if (n_leaves == 0 || n_leaves > 0xffffffffULL) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "merkle4_update: n_leaves must be in 1 .. 2^32 - 1");
    if (!tag || !d_leaves || (n_leaves > 1 && !d_levels)) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "merkle4_update: NULL buffer");
    if (k && (!d_indices || !d_new_leaves)) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "merkle4_update: NULL update list");
    if (misaligned(d_leaves) || misaligned(d_levels) || misaligned(d_new_leaves)) return fail(ctx, P252_ERR_INVALID_ARGUMENT, ALIGN_MSG);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)hip_stream;
    const TagArg t = tag_arg(tag);
    HIP_TRY(ctx, launch_scatter_scalars(d_indices, d_new_leaves, d_leaves, k, n_leaves, d_n_bad, st));
    const char* cur = static_cast<const char*>(d_leaves);
    size_t cur_n = n_leaves;
    char* lv = static_cast<char*>(d_levels);
    unsigned shift = 2;
    while (cur_n > 1) {
        const size_t next_n = (cur_n + 3) / 4;
        HIP_TRY(ctx, launch_merkle4_update(ctx->d_tab, t, d_indices, shift, cur, cur_n, lv, k, st));
        cur = lv;
        cur_n = next_n;
        lv += next_n * 32;
        shift += 2;
    }
    if (d_root) HIP_TRY(ctx, hipMemcpyAsync(d_root, cur, 32, hipMemcpyDeviceToDevice, st));
    return P252_OK;
}

int p252_merkle4_update_device(p252_ctx* ctx, const uint64_t tag[4], void* d_leaves, size_t n_leaves, void* d_levels,
                               const void* d_indices, const void* d_new_leaves, size_t k, void* d_root, void* hip_stream) {
    return merkle4_update_device(ctx, tag, d_leaves, n_leaves, d_levels, d_indices, d_new_leaves, k, d_root, nullptr, hip_stream);
}

int p252_merkle4_update_checked_device(p252_ctx* ctx, const uint64_t tag[4], void* d_leaves, size_t n_leaves, void* d_levels,
                                       const void* d_indices, const void* d_new_leaves, size_t k, void* d_root, void* d_n_bad,
                                       void* hip_stream) {
    if (!d_n_bad) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "merkle4_update_checked: d_n_bad is NULL");
    if ((reinterpret_cast<uintptr_t>(d_n_bad) & 3u) != 0) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "merkle4_update_checked: d_n_bad must be 4-byte aligned");
    return merkle4_update_device(ctx, tag, d_leaves, n_leaves, d_levels, d_indices, d_new_leaves, k, d_root, d_n_bad, hip_stream);
}

// ---- the canonical byte format of a scalar (BlsScalar::to_bytes / from_bytes; the reference round-trips its round
// constants through the pair, src/hades/round_constants.rs:66-67, and reads its KAT inputs with from_hex_str, src/hades.rs:131)
int p252_to_bytes_device(p252_ctx* ctx, const void* d_scalars, void* d_bytes, size_t n, void* hip_stream) {
    if (!ctx) return P252_ERR_INVALID_ARGUMENT;
    if (n == 0) return P252_OK;
    if (!d_scalars || !d_bytes) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "to_bytes: NULL buffer");
    if (misaligned(d_scalars) || misaligned(d_bytes)) return fail(ctx, P252_ERR_INVALID_ARGUMENT, ALIGN_MSG);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_to_canonical(d_scalars, d_bytes, n, (hipStream_t)hip_stream));
    return P252_OK;
}

int p252_from_bytes_device(p252_ctx* ctx, const void* d_bytes, void* d_scalars, void* d_ok, size_t n, void* hip_stream) {
    if (!ctx) return P252_ERR_INVALID_ARGUMENT;
    if (n == 0) return P252_OK;
    if (!d_scalars || !d_bytes) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "from_bytes: NULL buffer");
    if (misaligned(d_scalars) || misaligned(d_bytes)) return fail(ctx, P252_ERR_INVALID_ARGUMENT, ALIGN_MSG);
    static const Digits9 r2 = [] {
        Digits9 d;
        encode_balanced29(FrHost::pow2(517), d.d);  // v * 2^517 / 2^261 = v * 2^256: the Montgomery form
        return d;
    }();
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_from_canonical(d_bytes, d_scalars, d_ok, n, r2, (hipStream_t)hip_stream));
    return P252_OK;
}

int p252_merkle4_path_batch_device(p252_ctx* ctx, const uint64_t tag[4], const void* d_leaves, const void* d_siblings,
                                   const void* d_positions, size_t depth, void* d_roots, size_t n, void* hip_stream) {
    if (!ctx) return P252_ERR_INVALID_ARGUMENT;
    if (n == 0) return P252_OK;
    if (!tag || !d_leaves || !d_roots || (depth && (!d_siblings || !d_positions)))
        return fail(ctx, P252_ERR_INVALID_ARGUMENT, "merkle4_path: NULL buffer");
    if (depth > 0xffffu) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "merkle4_path: depth too large");
    if (misaligned(d_leaves) || misaligned(d_roots) || (depth && misaligned(d_siblings))) return fail(ctx, P252_ERR_INVALID_ARGUMENT, ALIGN_MSG);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_merkle4_path(ctx->d_tab, tag_arg(tag), d_leaves, d_siblings, d_positions, (unsigned)depth,
                                     d_roots, n, (hipStream_t)hip_stream));
    return P252_OK;
}

// arity-2 twin of p252_merkle4_path_batch_device (Domain::Merkle2 nodes; pass the Merkle2 tag): siblings[n][depth], positions 0..1
int p252_merkle2_path_batch_device(p252_ctx* ctx, const uint64_t tag[4], const void* d_leaves, const void* d_siblings,
                                   const void* d_positions, size_t depth, void* d_roots, size_t n, void* hip_stream) {
    if (!ctx) return P252_ERR_INVALID_ARGUMENT;
    if (n == 0) return P252_OK;
    if (!tag || !d_leaves || !d_roots || (depth && (!d_siblings || !d_positions)))
        return fail(ctx, P252_ERR_INVALID_ARGUMENT, "merkle2_path: NULL buffer");
    if (depth > 0xffffu) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "merkle2_path: depth too large");
    if (misaligned(d_leaves) || misaligned(d_roots) || (depth && misaligned(d_siblings))) return fail(ctx, P252_ERR_INVALID_ARGUMENT, ALIGN_MSG);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_merkle2_path(ctx->d_tab, tag_arg(tag), d_leaves, d_siblings, d_positions, (unsigned)depth, d_roots, n, (hipStream_t)hip_stream));
    return P252_OK;
}

// Opening::verify in bulk: re-hash (either arity) into this stream's scratch, then one byte per opening
static int verify_batch_device(p252_ctx* ctx, unsigned arity, const uint64_t tag[4], const void* d_leaves, const void* d_siblings,
                               const void* d_positions, size_t depth, const void* d_root, void* d_ok, size_t n, void* hip_stream) {
    if (!ctx) return P252_ERR_INVALID_ARGUMENT;
    if (n == 0) return P252_OK;
    const std::string who = arity == 4 ? "merkle4_verify" : "merkle2_verify";
    if (!d_root || !d_ok) return fail(ctx, P252_ERR_INVALID_ARGUMENT, who + ": NULL buffer");
    if (misaligned(d_root)) return fail(ctx, P252_ERR_INVALID_ARGUMENT, ALIGN_MSG);
    if (n > (SIZE_MAX / 32)) return fail(ctx, P252_ERR_INVALID_ARGUMENT, who + ": size overflow");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)hip_stream;
    LevelSetGuard guard(ctx);
    p252_ctx::LevelSet*& set = guard.set;  // the recomputed roots live in the scratch pair of THIS stream (ctx.hpp): n x 32 bytes
    int rc = level_set(ctx, st, n * 32, 0, &set);
    if (rc) return rc;
    rc = arity == 4 ? p252_merkle4_path_batch_device(ctx, tag, d_leaves, d_siblings, d_positions, depth, set->buf[0], n, hip_stream)
                    : p252_merkle2_path_batch_device(ctx, tag, d_leaves, d_siblings, d_positions, depth, set->buf[0], n, hip_stream);
    if (rc == P252_OK) {
        const hipError_t e = launch_compare_roots(set->buf[0], d_root, d_ok, n, st);
        if (e != hipSuccess) rc = fail(ctx, P252_ERR_HIP, who + ": " + hipGetErrorString(e));
    }
    const std::string msg = ctx->err;
    const int rc2 = guard.finish();
    if (rc) ctx->err = msg;
    return rc ? rc : rc2;
}

int p252_merkle4_verify_batch_device(p252_ctx* ctx, const uint64_t tag[4], const void* d_leaves, const void* d_siblings,
                                     const void* d_positions, size_t depth, const void* d_root, void* d_ok, size_t n, void* hip_stream) {
    return verify_batch_device(ctx, 4, tag, d_leaves, d_siblings, d_positions, depth, d_root, d_ok, n, hip_stream);
}

int p252_merkle2_verify_batch_device(p252_ctx* ctx, const uint64_t tag[4], const void* d_leaves, const void* d_siblings,
                                     const void* d_positions, size_t depth, const void* d_root, void* d_ok, size_t n, void* hip_stream) {
    return verify_batch_device(ctx, 2, tag, d_leaves, d_siblings, d_positions, depth, d_root, d_ok, n, hip_stream);
}

size_t p252_merkle2_depth(size_t n_leaves) {
    size_t d = 0;
    for (size_t c = n_leaves; c > 1; c = (c + 1) / 2) ++d;
    return d;
}

// depth of the arity-4 tree over n_leaves = number of levels above the leaves (a 4^k-leaf tree: k; a single leaf: 0)
size_t p252_merkle4_depth(size_t n_leaves) {
    size_t d = 0;
    for (size_t c = n_leaves; c > 1; c = (c + 3) / 4) ++d;
    return d;
}

// Openings of a STORED tree, extracted on the device (openings.hip: pure data movement): for each of the k leaf positions the
// leaf, the three siblings per level and the position bytes, in exactly the layout p252_merkle4_path_batch_device takes — build
// (all levels) -> extract -> verify never leaves the GPU.
static int openings_device(p252_ctx* ctx, unsigned arity, const void* d_leaves, size_t n_leaves, const void* d_levels, const void* d_indices, size_t k,
                           void* d_leaves_out, void* d_siblings, void* d_positions, void* d_n_bad, void* hip_stream) {
    if (!ctx) return P252_ERR_INVALID_ARGUMENT;
    if (k == 0) return P252_OK;
    const std::string who = arity == 4 ? "merkle4_openings" : "merkle2_openings";
    if (n_leaves == 0 || n_leaves > 0xffffffffu) return fail(ctx, P252_ERR_INVALID_ARGUMENT, who + ": n_leaves must be in 1 .. 2^32 - 1 (positions are uint32)");
    const size_t depth = arity == 4 ? p252_merkle4_depth(n_leaves) : p252_merkle2_depth(n_leaves);
    if (!d_leaves || !d_indices || !d_leaves_out || (depth && (!d_levels || !d_siblings || !d_positions)))
        return fail(ctx, P252_ERR_INVALID_ARGUMENT, who + ": NULL buffer");
    if (misaligned(d_leaves) || misaligned(d_levels) || misaligned(d_leaves_out) || misaligned(d_siblings)) return fail(ctx, P252_ERR_INVALID_ARGUMENT, ALIGN_MSG);
    if ((reinterpret_cast<uintptr_t>(d_indices) & 3u) || (reinterpret_cast<uintptr_t>(d_n_bad) & 3u))
        return fail(ctx, P252_ERR_INVALID_ARGUMENT, who + ": indices / counter must be 4-byte aligned");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)hip_stream;
    if (d_n_bad) HIP_TRY(ctx, hipMemsetAsync(d_n_bad, 0, 4, st));
    HIP_TRY(ctx, (arity == 4 ? launch_merkle4_openings : launch_merkle2_openings)(d_leaves, n_leaves, d_levels, d_indices, k, (unsigned)depth, d_leaves_out, d_siblings,
                                                                                       d_positions, d_n_bad, st));
    return P252_OK;
}

int p252_merkle4_openings_device(p252_ctx* ctx, const void* d_leaves, size_t n_leaves, const void* d_levels, const void* d_indices, size_t k,
                                 void* d_leaves_out, void* d_siblings, void* d_positions, void* d_n_bad, void* hip_stream) {
    return openings_device(ctx, 4, d_leaves, n_leaves, d_levels, d_indices, k, d_leaves_out, d_siblings, d_positions, d_n_bad, hip_stream);
}

int p252_merkle2_openings_device(p252_ctx* ctx, const void* d_leaves, size_t n_leaves, const void* d_levels, const void* d_indices, size_t k,
                                 void* d_leaves_out, void* d_siblings, void* d_positions, void* d_n_bad, void* hip_stream) {
    return openings_device(ctx, 2, d_leaves, n_leaves, d_levels, d_indices, k, d_leaves_out, d_siblings, d_positions, d_n_bad, hip_stream);
}

int p252_merkle4_path_batch(p252_ctx* ctx, const uint64_t tag[4], const uint64_t* leaves, const uint64_t* siblings,
                            const uint8_t* positions, size_t depth, uint64_t* roots, size_t n) {
    if (!ctx) return P252_ERR_INVALID_ARGUMENT;
    if (n == 0) return P252_OK;
    if (!tag || !leaves || !roots || (depth && (!siblings || !positions)))
        return fail(ctx, P252_ERR_INVALID_ARGUMENT, "merkle4_path: NULL buffer");
    for (size_t i = 0; i < n * depth; ++i)
        if (positions[i] > 3) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "merkle4_path: position outside 0..3");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (depth > 0xffffu) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "merkle4_path: depth too large");
    if (depth) {
        const size_t chunk = staging_chunk_items(64 + depth * 97);
        if (n >= 2 * chunk) {  // large batch: leaves, sibling blocks and positions stream through the staging lanes
            const TagArg targ = tag_arg(tag);
            return staged_run(ctx, n, chunk,
                              {{reinterpret_cast<const char*>(leaves), nullptr, 32},
                               {reinterpret_cast<const char*>(siblings), nullptr, depth * 96},
                               {reinterpret_cast<const char*>(positions), nullptr, depth}},
                              {{nullptr, reinterpret_cast<char*>(roots), 32}},
                              [&](const void* const* d_in, void* const* d_out, size_t, size_t cnt, hipStream_t st) {
                                  return launch_merkle4_path(ctx->d_tab, targ, d_in[0], d_in[1], d_in[2], (unsigned)depth, d_out[0], cnt, st);
                              });
        }
    }
    const size_t leaf_b = n * 32, sib_b = n * depth * 96, pos_b = (n * depth + 15) & ~(size_t)15;
    int rc = ensure(ctx, &ctx->d_in, &ctx->d_in_cap, leaf_b + sib_b + pos_b + 16);
    if (rc) return rc;
    rc = ensure(ctx, &ctx->d_out, &ctx->d_out_cap, leaf_b);
    if (rc) return rc;
    char* base = static_cast<char*>(ctx->d_in);
    HIP_TRY(ctx, hipMemcpy(base, leaves, leaf_b, hipMemcpyHostToDevice));
    if (depth) {
        HIP_TRY(ctx, hipMemcpy(base + leaf_b, siblings, sib_b, hipMemcpyHostToDevice));
        HIP_TRY(ctx, hipMemcpy(base + leaf_b + sib_b, positions, n * depth, hipMemcpyHostToDevice));
    }
    rc = p252_merkle4_path_batch_device(ctx, tag, base, base + leaf_b, base + leaf_b + sib_b, depth, ctx->d_out, n, nullptr);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpy(roots, ctx->d_out, leaf_b, hipMemcpyDeviceToHost));
    return P252_OK;
}

// ---- encryption row (src/encryption.rs:62-95 -> dusk_safe::encrypt / decrypt) ----
// The sponge-call sequence dusk_safe::encrypt makes, as a table (one word per call: kind << 29 | len; kinds as in
// kernels.hip k_crypt).  Two candidates (include/poseidon252_hip.h): P252_CRYPT_STREAM — squeeze all `len` masks, then
// absorb the whole message — and P252_CRYPT_DUPLEX — squeeze / absorb per chunk of <= 4.  Adding a third is one more
// branch here: the kernel, the tag and the oracle all derive from the table.
static std::vector<uint32_t> crypt_program(int variant, size_t len) {
    std::vector<uint32_t> prog;
    auto call = [&](uint32_t kind, size_t n) { prog.push_back((kind << 29) | (uint32_t)n); };
    call(0, 2);  // Absorb(2): shared secret (u, v)
    call(1, 1);  // Absorb(1): nonce
    if (variant == P252_CRYPT_STREAM) {
        call(2, len);  // Squeeze(len): masks
        call(3, len);  // Absorb(len): message
    } else {
        for (size_t left = len; left;) {
            const size_t c = left < 4 ? left : 4;
            call(2, c);
            call(3, c);
            left -= c;
        }
    }
    call(4, 1);  // Squeeze(1): MAC
    return prog;
}

static bool crypt_variant_ok(int variant) { return variant == P252_CRYPT_STREAM || variant == P252_CRYPT_DUPLEX; }

// the sponge-call table of (variant, len) on the device (kept until another one is asked for)
static int prepare_prog(p252_ctx* ctx, int variant, size_t len) {
    if (ctx->prog_variant == variant && ctx->prog_len == len) return P252_OK;
    const std::vector<uint32_t> prog = crypt_program(variant, len);
    // the previous program may still be read by a kernel in flight on another stream: drain before replacing it
    HIP_TRY(ctx, hipDeviceSynchronize());
    int rc = ensure(ctx, (void**)&ctx->d_prog, &ctx->d_prog_cap, prog.size() * sizeof(uint32_t));
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpy(ctx->d_prog, prog.data(), prog.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    ctx->prog_variant = variant;
    ctx->prog_len = len;
    ctx->prog_calls = (unsigned)prog.size();
    return P252_OK;
}

static int crypt_device(p252_ctx* ctx, int variant, bool decrypt, const uint64_t tag[4], const void* d_in, const void* d_secrets,
                        const void* d_nonces, size_t len, void* d_out, void* d_ok, size_t n, void* hip_stream) {
    if (!ctx) return P252_ERR_INVALID_ARGUMENT;
    if (!crypt_variant_ok(variant)) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "encrypt/decrypt: unknown variant");
    if (len == 0) return fail(ctx, P252_ERR_INVALID_IO_PATTERN, "encrypt/decrypt: empty message");
    if (len >= 0x1ffffff0u) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "encrypt/decrypt: message too long");
    if (n == 0) return P252_OK;
    if (!tag || !d_in || !d_secrets || !d_nonces || !d_out || (decrypt && !d_ok))
        return fail(ctx, P252_ERR_INVALID_ARGUMENT, "encrypt/decrypt: NULL buffer");
    if (misaligned(d_in) || misaligned(d_secrets) || misaligned(d_nonces) || misaligned(d_out)) return fail(ctx, P252_ERR_INVALID_ARGUMENT, ALIGN_MSG);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)hip_stream;
    int rc = prepare_prog(ctx, variant, len);
    if (rc) return rc;
    HIP_TRY(ctx, launch_crypt(decrypt, ctx->d_tab, tag_arg(tag), d_in, d_secrets, d_nonces, (unsigned)len, d_out, d_ok, n, ctx->d_prog,
                              ctx->prog_calls, st));
    return P252_OK;
}

int p252_encrypt_batch_device(p252_ctx* ctx, int variant, const uint64_t tag[4], const void* d_messages, const void* d_secrets,
                              const void* d_nonces, size_t len, void* d_ciphers, size_t n, void* hip_stream) {
    return crypt_device(ctx, variant, false, tag, d_messages, d_secrets, d_nonces, len, d_ciphers, nullptr, n, hip_stream);
}

int p252_decrypt_batch_device(p252_ctx* ctx, int variant, const uint64_t tag[4], const void* d_ciphers, const void* d_secrets,
                              const void* d_nonces, size_t len, void* d_messages, void* d_ok, size_t n, void* hip_stream) {
    return crypt_device(ctx, variant, true, tag, d_ciphers, d_secrets, d_nonces, len, d_messages, d_ok, n, hip_stream);
}

static int crypt_host_run(p252_ctx* ctx, int variant, bool decrypt, const uint64_t tag[4], const uint64_t* in, const uint64_t* secrets,
                          const uint64_t* nonces, size_t len, uint64_t* out, uint8_t* ok, size_t n, size_t* used_in, size_t* used_out,
                          bool* used_lanes) {
    if (!ctx) return P252_ERR_INVALID_ARGUMENT;
    if (!crypt_variant_ok(variant)) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "encrypt/decrypt: unknown variant");
    if (len == 0) return fail(ctx, P252_ERR_INVALID_IO_PATTERN, "encrypt/decrypt: empty message");
    if (n == 0) return P252_OK;
    if (!tag || !in || !secrets || !nonces || !out || (decrypt && !ok))
        return fail(ctx, P252_ERR_INVALID_ARGUMENT, "encrypt/decrypt: NULL buffer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    {
        const size_t in_stride = (decrypt ? len + 1 : len) * 32, out_stride = (decrypt ? len : len + 1) * 32;
        const size_t chunk = staging_chunk_items(in_stride + out_stride + 96);
        if (n >= 2 * chunk && len < 0x1ffffff0u) {  // large batch: through the staging lanes, chunk by chunk
            // the call table is uploaded once, here; the lanes then only launch (no shared state, no lock, and a failure
            // keeps its own HIP error string)
            int rc0 = prepare_prog(ctx, variant, len);
            if (rc0) return rc0;
            const TagArg targ = tag_arg(tag);
            const uint32_t* d_prog = ctx->d_prog;
            const unsigned n_calls = ctx->prog_calls;
            std::vector<HostSpan> ins = {{reinterpret_cast<const char*>(in), nullptr, in_stride},
                                         {reinterpret_cast<const char*>(secrets), nullptr, 64},
                                         {reinterpret_cast<const char*>(nonces), nullptr, 32}};
            std::vector<HostSpan> outs = {{nullptr, reinterpret_cast<char*>(out), out_stride}};
            if (decrypt) outs.push_back({nullptr, reinterpret_cast<char*>(ok), 1});
            *used_lanes = true;
            return staged_run(ctx, n, chunk, ins, outs,
                              [&](const void* const* d_in, void* const* d_out, size_t, size_t cnt, hipStream_t st) {
                                  return launch_crypt(decrypt, ctx->d_tab, targ, d_in[0], d_in[1], d_in[2], (unsigned)len, d_out[0],
                                                      decrypt ? d_out[1] : nullptr, cnt, d_prog, n_calls, st);
                              });
        }
    }
    const size_t in_b = n * (decrypt ? len + 1 : len) * 32, out_b = n * (decrypt ? len : len + 1) * 32;
    const size_t sec_b = n * 64, non_b = n * 32, ok_b = (n + 15) & ~(size_t)15;
    int rc = ensure(ctx, &ctx->d_in, &ctx->d_in_cap, in_b + sec_b + non_b);
    if (rc) return rc;
    rc = ensure(ctx, &ctx->d_out, &ctx->d_out_cap, out_b + ok_b);
    if (rc) return rc;
    char* di = static_cast<char*>(ctx->d_in);
    char* dout = static_cast<char*>(ctx->d_out);
    *used_in = in_b + sec_b + non_b;
    *used_out = out_b + ok_b;
    HIP_TRY(ctx, hipMemcpy(di, in, in_b, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(di + in_b, secrets, sec_b, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(di + in_b + sec_b, nonces, non_b, hipMemcpyHostToDevice));
    rc = crypt_device(ctx, variant, decrypt, tag, di, di + in_b, di + in_b + sec_b, len, dout, dout + out_b, n, nullptr);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpy(out, dout, out_b, hipMemcpyDeviceToHost));
    if (decrypt) HIP_TRY(ctx, hipMemcpy(ok, dout + out_b, n, hipMemcpyDeviceToHost));
    return P252_OK;
}

// the host-buffer encrypt / decrypt: whatever the call copied into library-owned memory — shared secrets, nonces, plaintexts
// and ciphertexts in the context's device scratch or in the staging lanes (page-locked host + device chunks) — is cleared
// before it returns, on success and on failure (the reference: zeroize, Cargo.toml:14; dusk-safe zeroizes a finished sponge)
static int crypt_host(p252_ctx* ctx, int variant, bool decrypt, const uint64_t tag[4], const uint64_t* in, const uint64_t* secrets,
                      const uint64_t* nonces, size_t len, uint64_t* out, uint8_t* ok, size_t n) {
    size_t used_in = 0, used_out = 0;
    bool used_lanes = false;
    const int rc = crypt_host_run(ctx, variant, decrypt, tag, in, secrets, nonces, len, out, ok, n, &used_in, &used_out, &used_lanes);
    if (ctx && (used_in || used_out || used_lanes)) {
        const std::string msg = ctx->err;
        // nothing of THIS call may still be in flight on what is cleared: the call ran on the staging lanes' streams (non-blocking
        // streams) or on the null stream — those are synchronised, not the device (other contexts and streams keep running),
        // and only the bytes the lanes may have written are cleared, not their whole capacity
        hipError_t e = hipSuccess;
        for (auto& l : ctx->lanes)
            if (used_lanes && l.st && e == hipSuccess) e = hipStreamSynchronize(l.st);
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
        if (e == hipSuccess) e = wipe_span(ctx->d_in, used_in);
        if (e == hipSuccess) e = wipe_span(ctx->d_out, used_out);
        if (e == hipSuccess && used_lanes) e = wipe_lanes(ctx, /*dirty_only=*/true);
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
        if (e != hipSuccess && rc == P252_OK) return fail(ctx, P252_ERR_HIP, std::string("encrypt/decrypt: wiping the scratch failed: ") + hipGetErrorString(e));
        ctx->err = msg;
    }
    return rc;
}

int p252_encrypt_batch(p252_ctx* ctx, int variant, const uint64_t tag[4], const uint64_t* messages, const uint64_t* secrets,
                       const uint64_t* nonces, size_t len, uint64_t* ciphers, size_t n) {
    return crypt_host(ctx, variant, false, tag, messages, secrets, nonces, len, ciphers, nullptr, n);
}

int p252_decrypt_batch(p252_ctx* ctx, int variant, const uint64_t tag[4], const uint64_t* ciphers, const uint64_t* secrets,
                       const uint64_t* nonces, size_t len, uint64_t* messages, uint8_t* ok, size_t n) {
    return crypt_host(ctx, variant, true, tag, ciphers, secrets, nonces, len, messages, ok, n);
}

// shared by p252_tag and p252_encryption_tag: tag = hash_to_scalar(io-words (BE u32) || domain separator (BE u64)),
// hash_to_scalar = BLAKE2b-512 read as a 512-bit little-endian integer mod p.  UNPINNED recipe (header).
static void tag_from_words(const std::vector<uint32_t>& words, uint64_t sep, uint64_t tag_out[4]) {
    std::vector<uint8_t> buf;
    for (uint32_t w : words)
        for (int b = 0; b < 4; ++b) buf.push_back((uint8_t)(w >> (24 - 8 * b)));
    for (int b = 0; b < 8; ++b) buf.push_back((uint8_t)(sep >> (56 - 8 * b)));
    uint8_t h[64];
    blake2b_512(buf.data(), buf.size(), h);
    uint64_t lo[4], hi[4];
    for (int k = 0; k < 4; ++k) {
        lo[k] = u64_from_buffer(h, 8 * k);
        hi[k] = u64_from_buffer(h, 32 + 8 * k);
    }
    FrHost r = FrHost::from_raw(lo) + FrHost::from_raw(hi) * FrHost::pow2(256);
    std::memcpy(tag_out, r.l, 32);
}

// UNPINNED: tag of the encryption io-pattern of `variant` (adjacent calls of one kind aggregated, as dusk-safe does)
int p252_encryption_tag(int variant, size_t message_len, uint64_t tag_out[4]) {
    if (!tag_out || !crypt_variant_ok(variant)) return P252_ERR_INVALID_ARGUMENT;
    if (message_len == 0) return P252_ERR_INVALID_IO_PATTERN;
    if (message_len >= 0x1ffffff0ULL) return P252_ERR_INVALID_ARGUMENT;
    std::vector<uint32_t> words;
    int prev_absorb = -1;
    for (uint32_t c : crypt_program(variant, message_len)) {
        const uint32_t kind = c >> 29, n = c & 0x1fffffffu;
        const int is_absorb = kind == 0 || kind == 1 || kind == 3;
        if (!words.empty() && is_absorb == prev_absorb)
            words.back() += n;
        else
            words.push_back((is_absorb ? 0x80000000u : 0u) | n);
        prev_absorb = is_absorb;
    }
    uint64_t sep = 0;
    p252_domain_separator(P252_DOMAIN_ENCRYPTION, &sep);
    tag_from_words(words, sep, tag_out);
    return P252_OK;
}

// ------------------------------------------------------------------------------------------
// multi-device entry points (SURVEY §8(b) sketch, §8(e)): an array of contexts, one per GPU.  Shards are contiguous and
// independent — no inter-GPU dependence, no collective on the data path; the only exchange of the sharded tree is the
// 32-byte subtree root of every device, gathered by the host.  One host thread per context for the duration of the call.
// ------------------------------------------------------------------------------------------
extern "C++" {
bool p252host::power_of_4(size_t v) {
    if (v == 0 || (v & (v - 1))) return false;
    int tz = 0;
    while (!((v >> tz) & 1)) ++tz;
    return (tz & 1) == 0;
}
}  // extern "C++"

extern "C++" {
int p252host::check_ctxs(p252_ctx* const* ctxs, size_t n_ctx) {
    if (!ctxs || n_ctx == 0) return P252_ERR_INVALID_ARGUMENT;
    for (size_t t = 0; t < n_ctx; ++t)
        if (!ctxs[t]) return P252_ERR_INVALID_ARGUMENT;
    for (size_t a = 0; a < n_ctx; ++a)  // a context is used by one thread at a time: no duplicates
        for (size_t b = a + 1; b < n_ctx; ++b)
            if (ctxs[a] == ctxs[b]) return fail(ctxs[0], P252_ERR_INVALID_ARGUMENT, "multi: the same context appears twice");
    // one context per GPU is the point of these entry points: when the node has at least n_ctx devices, two contexts on the
    // same device are a caller's mistake (every shard would land on one GPU).  Fewer devices than contexts — the
    // single-GPU test configuration — is allowed.  P252_MULTI_ALLOW_SHARED_DEVICE=1 lifts the check.
    static const bool allow_shared = [] {
        const char* e = std::getenv("P252_MULTI_ALLOW_SHARED_DEVICE");
        return e && e[0] == '1';
    }();
    if (!allow_shared && (size_t)p252_device_count() >= n_ctx)
        for (size_t a = 0; a < n_ctx; ++a)
            for (size_t b = a + 1; b < n_ctx; ++b)
                if (ctxs[a]->device == ctxs[b]->device)
                    return fail(ctxs[0], P252_ERR_INVALID_ARGUMENT,
                                "multi: contexts " + std::to_string(a) + " and " + std::to_string(b) + " are bound to the same device although the node has one per context");
    return P252_OK;
}
}  // extern "C++"

// for the duration of one multi call every context works within its share of the CPU budget
struct LaneBudgetScope {
    p252_ctx* const* ctxs;
    size_t n;
    LaneBudgetScope(p252_ctx* const* c, size_t n_ctx) : ctxs(c), n(n_ctx) {
        const int per = staging_lanes_per_ctx(n_ctx);
        for (size_t t = 0; t < n; ++t) ctxs[t]->lane_budget = per;
    }
    ~LaneBudgetScope() {
        for (size_t t = 0; t < n; ++t) ctxs[t]->lane_budget = 0;
    }
};

int p252_hash_batch_multi(p252_ctx* const* ctxs, size_t n_ctx, const uint64_t tag[4], const uint64_t* in, size_t in_len,
                          size_t out_len, uint64_t* out, size_t n) {
    int rc = check_ctxs(ctxs, n_ctx);
    if (rc) return rc;
    if (in_len == 0 || out_len == 0) return fail(ctxs[0], P252_ERR_INVALID_IO_PATTERN, "hash: in_len and out_len must be > 0");
    if (n == 0) return P252_OK;
    if (!tag || !in || !out) return fail(ctxs[0], P252_ERR_INVALID_ARGUMENT, "hash: NULL buffer");
    // contiguous shards, sizes differing by at most one item
    const size_t base = n / n_ctx, rem = n % n_ctx;
    LaneBudgetScope budget(ctxs, n_ctx);
    return for_each_ctx(ctxs, n_ctx, [&](size_t t) {
        const size_t lo = t * base + (t < rem ? t : rem), cnt = base + (t < rem ? 1 : 0);
        return p252_hash_batch(ctxs[t], tag, in + lo * in_len * 4, in_len, out_len, out + lo * out_len * 4, cnt);
    });
}

int p252_hash_batch_multi_device(p252_ctx* const* ctxs, size_t n_ctx, const uint64_t tag[4], const void* const* d_in, size_t in_len,
                                 size_t out_len, void* const* d_out, const size_t* n_per_ctx, void* const* hip_streams) {
    int rc = check_ctxs(ctxs, n_ctx);
    if (rc) return rc;
    if (!d_in || !d_out || !n_per_ctx) return fail(ctxs[0], P252_ERR_INVALID_ARGUMENT, "hash_multi_device: NULL array");
    for (size_t t = 0; t < n_ctx; ++t) {  // launches are asynchronous: a plain loop, no threads needed
        rc = p252_hash_batch_device(ctxs[t], tag, d_in[t], in_len, out_len, d_out[t], n_per_ctx[t], hip_streams ? hip_streams[t] : nullptr);
        if (rc) {
            if (t) ctxs[0]->err = "context " + std::to_string(t) + ": " + ctxs[t]->err;
            return rc;
        }
    }
    return P252_OK;
}

static int tree_top(p252_ctx* ctx0, const uint64_t tag[4], const std::vector<uint64_t>& roots, size_t n_ctx, uint64_t root[4]) {
    if (n_ctx == 1) {
        std::memcpy(root, roots.data(), 32);
        return P252_OK;
    }
    return p252_merkle4_tree(ctx0, tag, roots.data(), n_ctx, root, nullptr);  // <= log4(n_ctx) + 1 tiny levels, zero-padded
}

int p252_merkle4_tree_multi(p252_ctx* const* ctxs, size_t n_ctx, const uint64_t tag[4], const uint64_t* leaves, size_t n_leaves,
                            uint64_t root[4]) {
    int rc = check_ctxs(ctxs, n_ctx);
    if (rc) return rc;
    if (!tag || !leaves || !root) return fail(ctxs[0], P252_ERR_INVALID_ARGUMENT, "merkle_tree_multi: NULL buffer");
    if (n_leaves == 0 || n_leaves % n_ctx || !power_of_4(n_leaves / n_ctx))
        return fail(ctxs[0], P252_ERR_INVALID_ARGUMENT,
                    "merkle_tree_multi: every device must own a complete subtree (n_leaves = n_ctx * 4^k)");
    const size_t m = n_leaves / n_ctx;
    std::vector<uint64_t> roots(4 * n_ctx);
    LaneBudgetScope budget(ctxs, n_ctx);
    rc = for_each_ctx(ctxs, n_ctx, [&](size_t t) { return p252_merkle4_tree(ctxs[t], tag, leaves + t * m * 4, m, &roots[4 * t], nullptr); });
    if (rc) return rc;
    return tree_top(ctxs[0], tag, roots, n_ctx, root);
}

int p252_merkle4_tree_multi_device(p252_ctx* const* ctxs, size_t n_ctx, const uint64_t tag[4], const void* const* d_leaves,
                                   size_t leaves_per_ctx, uint64_t root[4]) {
    int rc = check_ctxs(ctxs, n_ctx);
    if (rc) return rc;
    if (!tag || !d_leaves || !root) return fail(ctxs[0], P252_ERR_INVALID_ARGUMENT, "merkle_tree_multi_device: NULL buffer");
    if (!power_of_4(leaves_per_ctx))
        return fail(ctxs[0], P252_ERR_INVALID_ARGUMENT, "merkle_tree_multi_device: every device must own a complete subtree (4^k leaves)");
    // RCCL path (contexts on distinct devices; the communicator is created on first use and kept): subtrees, ONE grouped
    // ncclAllGather of the 32-byte roots on the devices' streams, top levels on every device — the root is device-resident
    // everywhere and only its 32 bytes come back to the host, from ctxs[0]
    bool used_rccl = false;
    std::vector<void*> d_top(n_ctx, nullptr);
    for (size_t t = 0; t < n_ctx && rc == P252_OK; ++t) {
        p252_ctx* c = ctxs[t];
        if (hipSetDevice(c->device) != hipSuccess) { rc = fail(c, P252_ERR_HIP, "hipSetDevice"); break; }
        rc = ensure(c, &c->d_out, &c->d_out_cap, 32);
        d_top[t] = c->d_out;
    }
    if (rc == P252_OK) rc = tree_multi_device_rccl(ctxs, n_ctx, tag, d_leaves, leaves_per_ctx, d_top.data(), nullptr, &used_rccl);
    if (rc) {
        if (ctxs[0]->err.empty()) ctxs[0]->err = "merkle_tree_multi_device failed on another context";
        return rc;
    }
    if (used_rccl) {
        for (size_t t = 1; t < n_ctx; ++t) {  // the call is synchronous: every device has its copy of the root when it returns
            HIP_TRY(ctxs[t], hipSetDevice(ctxs[t]->device));
            HIP_TRY(ctxs[t], hipStreamSynchronize(nullptr));
        }
        HIP_TRY(ctxs[0], hipSetDevice(ctxs[0]->device));
        HIP_TRY(ctxs[0], hipMemcpy(root, d_top[0], 32, hipMemcpyDeviceToHost));
        return P252_OK;
    }
    // HOST-GATHER path — only when the contexts share a device (RCCL wants one device per rank: the single-GPU test
    // configuration) or P252_MULTI_HOST_GATHER=1: every device reduces its resident subtree (asynchronous launches on its
    // own default stream) ...
    std::vector<void*> d_roots(n_ctx, nullptr);
    for (size_t t = 0; t < n_ctx && rc == P252_OK; ++t) {
        p252_ctx* c = ctxs[t];
        if (hipSetDevice(c->device) != hipSuccess) { rc = fail(c, P252_ERR_HIP, "hipSetDevice"); break; }
        rc = ensure(c, &c->d_out, &c->d_out_cap, 32);
        if (rc == P252_OK) rc = merkle_tree_device(c, 4, tag, d_leaves[t], leaves_per_ctx, c->d_out, nullptr, nullptr);
        d_roots[t] = c->d_out;
    }
    // ... then the only exchange step of the path: 32 bytes per device to the host
    std::vector<uint64_t> roots(4 * n_ctx);
    for (size_t t = 0; t < n_ctx && rc == P252_OK; ++t) {
        p252_ctx* c = ctxs[t];
        if (hipSetDevice(c->device) != hipSuccess || hipMemcpy(&roots[4 * t], d_roots[t], 32, hipMemcpyDeviceToHost) != hipSuccess)
            rc = fail(c, P252_ERR_HIP, "merkle_tree_multi_device: root copy failed");
    }
    if (rc) {
        if (ctxs[0]->err.empty()) ctxs[0]->err = "merkle_tree_multi_device failed on another context";
        return rc;
    }
    return tree_top(ctxs[0], tag, roots, n_ctx, root);
}

// ------------------------------------------------------------------------------------------
// measurement aids
// ------------------------------------------------------------------------------------------
int p252_clock_probe_device(p252_ctx* ctx, void* d_out6, unsigned spin_us, void* hip_stream) {
    if (!ctx) return P252_ERR_INVALID_ARGUMENT;
    if (!d_out6 || (reinterpret_cast<uintptr_t>(d_out6) & 7u)) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "clock_probe: d_out6 must be an 8-byte aligned device buffer of 6 x uint64");
    if (spin_us > 1000000u) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "clock_probe: spin_us > 1 s");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_clock_probe(d_out6, spin_us * 100u, (hipStream_t)hip_stream));  // real-time counter: 100 MHz
    return P252_OK;
}

int p252_staging_lanes(size_t n_ctx) { return staging_lanes_per_ctx(n_ctx ? n_ctx : 1); }

// ------------------------------------------------------------------------------------------
// constant-table exchange
// ------------------------------------------------------------------------------------------
size_t p252_tables_size(void) { return (size_t)Tab29Layout::TOTAL * sizeof(int32_t); }

int p252_tables_export(p252_ctx* ctx, void* host_buf, size_t len) {
    if (!ctx || !host_buf || len != p252_tables_size()) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "tables_export: bad args");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpy(host_buf, ctx->d_tab, len, hipMemcpyDeviceToHost));
    return P252_OK;
}

int p252_tables_import(p252_ctx* ctx, const void* host_buf, size_t len) {
    if (!ctx || !host_buf || len != p252_tables_size()) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "tables_import: bad args");
    // The table is a pure function of arc.bin / mds.bin, which are compiled in: anything else than what this library
    // derives itself (with its Cauchy-structure and column-bound checks) is a corrupted or mismatched broadcast and
    // would give silently wrong digests — refuse it.
    if (std::memcmp(host_buf, host_tables().data(), len) != 0)
        return fail(ctx, P252_ERR_INVALID_ARGUMENT, "tables_import: the table differs from the one this library derives from its arc.bin / mds.bin");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemcpy(ctx->d_tab, host_buf, len, hipMemcpyHostToDevice));
    std::memcpy(ctx->h_tab.data(), host_buf, len);
    return P252_OK;
}

// ------------------------------------------------------------------------------------------
// host helpers mirroring src/hash.rs
// ------------------------------------------------------------------------------------------
int p252_domain_separator(int domain, uint64_t* sep_out) {  // hash.rs:38-56
    if (!sep_out) return P252_ERR_INVALID_ARGUMENT;
    switch (domain) {
        case P252_DOMAIN_MERKLE4: *sep_out = 0x000000000000000fULL; return P252_OK;
        case P252_DOMAIN_MERKLE2: *sep_out = 0x0000000000000003ULL; return P252_OK;
        case P252_DOMAIN_ENCRYPTION: *sep_out = 0x0000000100000000ULL; return P252_OK;
        case P252_DOMAIN_OTHER: *sep_out = 0; return P252_OK;
        default: return P252_ERR_INVALID_ARGUMENT;
    }
}

int p252_check_io_pattern(int domain, const size_t* absorb_lens, size_t n_absorbs, size_t out_len) {
    if (domain < 0 || domain > 3) return P252_ERR_INVALID_ARGUMENT;
    if (n_absorbs && !absorb_lens) return P252_ERR_INVALID_ARGUMENT;
    size_t total = 0;
    for (size_t i = 0; i < n_absorbs; ++i) total += absorb_lens[i];
    // hash.rs:70-78
    if (domain == P252_DOMAIN_MERKLE2 && (total != 2 || out_len != 1)) return P252_ERR_IO_PATTERN_VIOLATION;
    if (domain == P252_DOMAIN_MERKLE4 && (total != 4 || out_len != 1)) return P252_ERR_IO_PATTERN_VIOLATION;
    // dusk-safe: a pattern must start with an absorb, end with a squeeze, and hold no zero-length call
    if (n_absorbs == 0 || out_len == 0) return P252_ERR_INVALID_IO_PATTERN;
    for (size_t i = 0; i < n_absorbs; ++i)
        if (absorb_lens[i] == 0) return P252_ERR_INVALID_IO_PATTERN;
    return P252_OK;
}

int p252_tag(int domain, const size_t* absorb_lens, size_t n_absorbs, size_t out_len, uint64_t tag_out[4]) {
    if (!tag_out) return P252_ERR_INVALID_ARGUMENT;
    int rc = p252_check_io_pattern(domain, absorb_lens, n_absorbs, out_len);
    if (rc) return rc;
    uint64_t absorbed = 0;
    for (size_t i = 0; i < n_absorbs; ++i) absorbed += absorb_lens[i];
    if (absorbed >= 0x80000000ULL || out_len >= 0x80000000ULL) return P252_ERR_INVALID_ARGUMENT;
    // tag input: aggregated io-pattern words (big-endian u32, absorb flagged by the top bit), then
    // the domain separator as a big-endian u64
    uint64_t sep = 0;
    p252_domain_separator(domain, &sep);
    tag_from_words({0x80000000u | (uint32_t)absorbed, (uint32_t)out_len}, sep, tag_out);
    return P252_OK;
}

int p252_truncate250(const uint64_t* scalars, uint64_t* out_raw, size_t n) {  // hash.rs:164-183
    if (n && (!scalars || !out_raw)) return P252_ERR_INVALID_ARGUMENT;
    for (size_t i = 0; i < n; ++i) {
        FrHost::from_limbs(scalars + 4 * i).to_canonical(out_raw + 4 * i);
        out_raw[4 * i + 3] &= 0x03ffffffffffffffULL;
    }
    return P252_OK;
}

// host-side twins of the two conversions (one Montgomery multiplication per scalar: not worth a transfer)
int p252_to_bytes(const uint64_t* scalars, uint8_t* bytes, size_t n) {
    if (n && (!scalars || !bytes)) return P252_ERR_INVALID_ARGUMENT;
    for (size_t i = 0; i < n; ++i) {
        uint64_t c[4];
        FrHost::from_limbs(scalars + 4 * i).to_canonical(c);
        for (int k = 0; k < 32; ++k) bytes[32 * i + k] = (uint8_t)(c[k / 8] >> (8 * (k % 8)));
    }
    return P252_OK;
}

int p252_from_bytes(const uint8_t* bytes, uint64_t* scalars, uint8_t* ok, size_t n) {
    if (n && (!bytes || !scalars)) return P252_ERR_INVALID_ARGUMENT;
    for (size_t i = 0; i < n; ++i) {
        uint64_t v[4] = {0, 0, 0, 0};
        for (int k = 0; k < 32; ++k) v[k / 8] |= (uint64_t)bytes[32 * i + k] << (8 * (k % 8));
        if (ok) ok[i] = FrHost::geq_p(v) ? 0 : 1;
        while (FrHost::geq_p(v)) FrHost::sub_p(v);  // (a 256-bit value is below 3p)
        std::memcpy(scalars + 4 * i, FrHost::from_raw(v).l, 32);
    }
    return P252_OK;
}

}  // extern "C"
