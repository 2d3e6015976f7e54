// comm.cpp — RCCL inside the library (north_star: "RCCL broadcast of constants over xGMI", "RCCL gather
// of roots").  The hashing path has no data-path collective: digests and complete subtrees are independent.  What is
// exchanged is (i) the constant table, once, when a communicator is created — broadcast from rank 0 and VALIDATED against
// the table every rank derives itself, exactly as p252_tables_import validates — and (ii) the 32-byte subtree root of every
// rank in the sharded tree build: ONE ncclAllGather on the rank's stream, between the subtree's last launch and the top
// levels' first, so that the root lands device-resident on every GPU with no host round trip.
//
// Two ways to make a communicator, for the two ways the reference's callers would drive several GPUs:
//   p252_comm_create_rank   one process (or thread) per GPU: rank 0 calls p252_comm_unique_id, hands the 128 bytes to the
//                           others by whatever means the host program has (an env store, MPI, torch.distributed, a file),
//                           every rank calls create_rank with its own context             -> ncclCommInitRank
//   p252_comm_create_all    one process, an array of contexts on distinct devices        -> ncclCommInitAll
// The p252_*_multi_device entry points create the second kind themselves on first use.
//
// xGMI is point-to-point and these messages are 32 bytes x world: the exchange is pure latency (a few microseconds), never
// bandwidth; there is nothing to bucket or overlap.  It is on the stream so that the HOST never waits for it.
//
// RCCL itself is NOT linked: rccl_dyn.hpp resolves the ten entry points used here on the first call that needs them, preferring a
// copy the process already holds (torch's, the application's).  No RCCL -> P252_ERR_COMM from these entry points only.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "ctx.hpp"
#include "openings.h"
#include "rccl_dyn.hpp"

using namespace p252host;

static_assert(P252_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "P252_COMM_ID_BYTES must be RCCL's unique-id size");

struct p252_comm {
    ncclComm_t nccl = nullptr;
    p252_ctx* ctx = nullptr;
    int rank = 0, world = 1;
    void* d_sub = nullptr;    // this rank's subtree root (32 B)
    void* d_roots = nullptr;  // world x 32 B: the gathered roots, in rank order
    void* d_top = nullptr;    // 32 B: the root over the gathered roots (when the caller passes no output pointer)
    void* d_tab_in = nullptr;  // creation only: where the broadcast constant table lands before it is validated (freed afterwards)
    // d_sub / d_roots / d_top are ONE set per communicator and the sharded builds are asynchronous on a caller-chosen stream: a
    // build on another stream than the previous one first waits on the event recorded behind that one (collectives of one
    // communicator are ordered anyway — every rank must issue them in the same order), so two builds never share the buffers
    hipEvent_t done = nullptr;
    hipStream_t last_st = nullptr;
    bool used = false;
    // communicators of one p252_comm_create_all share this list (rank order): collectives issued for all of them by one
    // host thread go inside one ncclGroupStart / ncclGroupEnd
    std::shared_ptr<std::vector<p252_comm*>> clique;
    bool owned_by_ctx = false;  // created lazily by a p252_*_multi_device call: destroyed with its context
    // a peer's failed local build (k_poison_if_peer_failed, merkle2.hip): 1 + its rank, written by the device into host-mapped memory
    // behind the top levels of a sharded build; read and cleared by the host at p252_comm_check / p252_sync
    unsigned* h_fail = nullptr;
    unsigned* d_fail = nullptr;
};

// the RCCL entry points (rccl_dyn.hpp): set by the first successful need_rccl() and never changed afterwards — every p252_comm
// that exists was made through it, so code that holds a communicator may use R without asking again
// (an atomic pointer: ranks driven by one thread each — one process, several GPUs — may make their first call at the same time; they all
// store the same address, p252rccl::api() itself is serialised)
static std::atomic<const p252rccl::Api*> R{nullptr};

static int need_rccl(p252_ctx* ctx) {
    if (R.load(std::memory_order_acquire)) return P252_OK;
    std::string why;
    const p252rccl::Api* a = p252rccl::api(&why);
    if (!a) return fail(ctx, P252_ERR_COMM, why);
    R.store(a, std::memory_order_release);
    return P252_OK;
}

#define NCCL_TRY(ctx, expr)                                                                         \
    do {                                                                                            \
        ncclResult_t r_ = (expr);                                                                   \
        if (r_ != ncclSuccess)                                                                      \
            return fail(ctx, P252_ERR_COMM, std::string(#expr) + ": " + R.load()->GetErrorString(r_));    \
    } while (0)

namespace {

// everything a communicator allocates, BEFORE the first collective (ncclCommInitRank included): a rank that cannot allocate
// fails before its peers are blocked in a collective it would then never enter
int alloc_buffers(p252_comm* c) {
    p252_ctx* ctx = c->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMalloc(&c->d_sub, 32));
    HIP_TRY(ctx, hipMalloc(&c->d_roots, (size_t)c->world * 32));
    HIP_TRY(ctx, hipMalloc(&c->d_top, 32));
    HIP_TRY(ctx, hipMalloc(&c->d_tab_in, host_tables().size() * sizeof(int32_t)));
    HIP_TRY(ctx, hipEventCreateWithFlags(&c->done, hipEventDisableTiming));
    HIP_TRY(ctx, hipHostMalloc((void**)&c->h_fail, sizeof(unsigned), hipHostMallocMapped));
    *c->h_fail = 0;
    HIP_TRY(ctx, hipHostGetDevicePointer((void**)&c->d_fail, c->h_fail, 0));
    return P252_OK;
}

// the buffers' previous user, if it ran on another stream, is waited for on the device; the host never blocks
int comm_enter(p252_comm* c, hipStream_t st) {
    if (c->used && c->last_st != st) HIP_TRY(c->ctx, hipStreamWaitEvent(st, c->done, 0));
    return P252_OK;
}

int comm_leave(p252_comm* c, hipStream_t st) {
    c->used = true;
    c->last_st = st;
    HIP_TRY(c->ctx, hipEventRecord(c->done, st));
    return P252_OK;
}

void free_comm(p252_comm* c, bool abort) {
    if (!c) return;
    if (c->ctx) (void)hipSetDevice(c->ctx->device);
    if (c->nccl && R.load()) (void)(abort ? R.load()->CommAbort(c->nccl) : R.load()->CommDestroy(c->nccl));
    if (c->d_sub) (void)hipFree(c->d_sub);
    if (c->d_roots) (void)hipFree(c->d_roots);
    if (c->d_top) (void)hipFree(c->d_top);
    if (c->d_tab_in) (void)hipFree(c->d_tab_in);
    if (c->done) (void)hipEventDestroy(c->done);
    if (c->h_fail) (void)hipHostFree(c->h_fail);
    if (c->ctx && c->ctx->comm == c) c->ctx->comm = nullptr;
    if (c->clique)
        for (auto& p : *c->clique)
            if (p == c) p = nullptr;
    delete c;
}

// the constant table of rank `root`, broadcast over RCCL into a scratch buffer on every rank of `comms` (all of them
// driven by this thread: one group), compared with the table this library derives from its own arc.bin / mds.bin, then
// installed.  A mismatch is a corrupted or mismatched broadcast (another build of the library on another rank): refused.
int broadcast_and_validate(const std::vector<p252_comm*>& comms, int root) {
    const std::vector<int32_t>& ref = host_tables();
    const size_t bytes = ref.size() * sizeof(int32_t);
    std::vector<void*> scratch(comms.size(), nullptr);
    int rc = P252_OK;
    for (size_t t = 0; t < comms.size(); ++t) scratch[t] = comms[t]->d_tab_in;  // allocated with the communicator, before any collective
    {
        ncclResult_t r = R.load()->GroupStart();
        for (size_t t = 0; t < comms.size() && r == ncclSuccess; ++t) {
            p252_ctx* ctx = comms[t]->ctx;
            (void)hipSetDevice(ctx->device);
            r = R.load()->Broadcast(ctx->d_tab, scratch[t], bytes, ncclUint8, root, comms[t]->nccl, nullptr);
        }
        const ncclResult_t r2 = R.load()->GroupEnd();
        if (r == ncclSuccess) r = r2;
        if (r != ncclSuccess) rc = fail(comms[0]->ctx, P252_ERR_COMM, std::string("ncclBroadcast of the constant table: ") + R.load()->GetErrorString(r));
    }
    std::vector<int32_t> got(ref.size());
    for (size_t t = 0; t < comms.size() && rc == P252_OK; ++t) {
        p252_ctx* ctx = comms[t]->ctx;
        if (hipSetDevice(ctx->device) != hipSuccess || hipMemcpy(got.data(), scratch[t], bytes, hipMemcpyDeviceToHost) != hipSuccess) {  // (synchronises the null stream)
            rc = fail(ctx, P252_ERR_HIP, "comm: reading back the broadcast constant table failed");
        } else if (std::memcmp(got.data(), ref.data(), bytes) != 0) {
            rc = fail(ctx, P252_ERR_INVALID_ARGUMENT,
                      "comm: the constant table broadcast by rank " + std::to_string(root) + " differs from the one this library derives from its arc.bin / mds.bin");
        } else if (hipMemcpy(ctx->d_tab, scratch[t], bytes, hipMemcpyDeviceToDevice) != hipSuccess) {
            rc = fail(ctx, P252_ERR_HIP, "comm: installing the broadcast constant table failed");
        }
    }
    for (size_t t = 0; t < comms.size(); ++t)
        if (scratch[t]) {
            (void)hipSetDevice(comms[t]->ctx->device);
            (void)hipFree(scratch[t]);
            comms[t]->d_tab_in = nullptr;
        }
    if (rc != P252_OK && comms[0]->ctx->err.empty()) comms[0]->ctx->err = "comm: constant broadcast failed on another rank";
    return rc;
}

}  // namespace

namespace p252host {

// called by p252_destroy: a communicator the library created for this context goes with it
void release_ctx_comm(p252_ctx* ctx) {
    if (ctx && ctx->comm) {
        if (ctx->comm->owned_by_ctx)
            free_comm(ctx->comm, false);
        else
            ctx->comm->ctx = nullptr, ctx->comm = nullptr;  // the caller's: it outlives the context only as a husk to destroy
    }
}

// the host's side of k_poison_if_peer_failed: called where the host has just synchronised with the stream of a sharded build
// (p252_comm_check, p252_sync).  A set flag is reported once and cleared.
int comm_take_failure(p252_ctx* ctx) {
    if (!ctx || !ctx->comm || !ctx->comm->h_fail) return P252_OK;
    const unsigned f = __atomic_exchange_n(ctx->comm->h_fail, 0u, __ATOMIC_ACQ_REL);
    if (!f) return P252_OK;
    return fail(ctx, P252_ERR_COMM,
                "sharded tree: rank " + std::to_string(f - 1) + " reported a failed local build (its gathered root was the all-ones sentinel); "
                "the root this rank produced for that build has been overwritten with all-ones and must not be used");
}

// the communicators of `ctxs` (rank t = ctxs[t]) when they all belong to ONE clique in exactly this order, else empty
static std::vector<p252_comm*> clique_of(p252_ctx* const* ctxs, size_t n_ctx) {
    std::vector<p252_comm*> v;
    if (!ctxs[0]->comm || !ctxs[0]->comm->clique || ctxs[0]->comm->clique->size() != n_ctx) return v;
    const auto& cl = *ctxs[0]->comm->clique;
    for (size_t t = 0; t < n_ctx; ++t)
        if (!cl[t] || cl[t]->ctx != ctxs[t] || ctxs[t]->comm != cl[t] || cl[t]->rank != (int)t) return v;
    v.assign(cl.begin(), cl.end());
    return v;
}

// why the lazy communicator of the *_multi_device entry points could not be made (empty: not tried / made); read and written under g_lazy_mu
static std::string g_lazy_refused;
static std::mutex g_lazy_mu;
static std::string lazy_refused() {
    std::lock_guard<std::mutex> lk(g_lazy_mu);
    return g_lazy_refused;
}
static void set_lazy_refused(const std::string& why) {
    std::lock_guard<std::mutex> lk(g_lazy_mu);
    g_lazy_refused = why;
}

static bool distinct_devices(p252_ctx* const* ctxs, size_t n_ctx) {
    // (test-only: tests/test_comm_mock_ranks.py links the library against a mock RCCL that accepts several ranks on one device, to
    // run the multi-rank logic on a one-GPU box; real RCCL refuses such a communicator itself)
    static const bool allow_shared = [] {
        const char* e = std::getenv("P252_COMM_ALLOW_SHARED_DEVICE");
        return e && e[0] == '1';
    }();
    if (allow_shared) return true;
    for (size_t a = 0; a < n_ctx; ++a)
        for (size_t b = a + 1; b < n_ctx; ++b)
            if (ctxs[a]->device == ctxs[b]->device) return false;
    return true;
}

static int create_all(p252_ctx* const* ctxs, size_t n_ctx, std::vector<p252_comm*>& out, bool owned) {
    int rc0 = need_rccl(ctxs[0]);
    if (rc0) return rc0;
    for (size_t t = 0; t < n_ctx; ++t)
        if (ctxs[t]->comm) return fail(ctxs[0], P252_ERR_INVALID_ARGUMENT, "comm_create_all: context " + std::to_string(t) + " already belongs to a communicator");
    if (!distinct_devices(ctxs, n_ctx))
        return fail(ctxs[0], P252_ERR_INVALID_ARGUMENT, "comm_create_all: RCCL needs one device per rank; two of the contexts are bound to the same device");
    auto clique = std::make_shared<std::vector<p252_comm*>>(n_ctx, nullptr);
    out.assign(n_ctx, nullptr);
    int rc = P252_OK;
    for (size_t t = 0; t < n_ctx; ++t) {  // the objects and every device allocation first, the collectives after
        p252_comm* c = new p252_comm();
        c->ctx = ctxs[t];
        c->rank = (int)t;
        c->world = (int)n_ctx;
        c->clique = clique;
        c->owned_by_ctx = owned;
        (*clique)[t] = c;
        out[t] = c;
        ctxs[t]->comm = c;
        if (rc == P252_OK) rc = alloc_buffers(c);
        if (rc != P252_OK && t != 0 && ctxs[0]->err.empty()) ctxs[0]->err = "context " + std::to_string(t) + ": " + ctxs[t]->err;
    }
    if (rc == P252_OK) {
        std::vector<int> devs(n_ctx);
        for (size_t t = 0; t < n_ctx; ++t) devs[t] = ctxs[t]->device;
        std::vector<ncclComm_t> nc(n_ctx, nullptr);
        const ncclResult_t r = R.load()->CommInitAll(nc.data(), (int)n_ctx, devs.data());
        if (r != ncclSuccess) rc = fail(ctxs[0], P252_ERR_COMM, std::string("ncclCommInitAll: ") + R.load()->GetErrorString(r));
        for (size_t t = 0; t < n_ctx && rc == P252_OK; ++t) out[t]->nccl = nc[t];
    }
    if (rc == P252_OK) rc = broadcast_and_validate(out, 0);
    if (rc == P252_OK) set_lazy_refused("");  // (a creation that works — the caller's own included — lifts a remembered refusal)
    if (rc != P252_OK) {
        const std::string msg = ctxs[0]->err;
        for (auto*& c : out) {
            free_comm(c, true);
            c = nullptr;
        }
        out.clear();
        ctxs[0]->err = msg;
    }
    return rc;
}

// Sharded tree over the ranks of one single-process clique, everything device-resident and asynchronous: every device
// reduces its subtree on its stream, ONE grouped ncclAllGather moves the n roots (32 B each) to every device, every device
// hashes the top levels (<= log4(n) + 1 tiny launches, zero-padded per hash.rs:22-26).  d_root_out[t] (may be NULL)
// receives the root on device t.
static int tree_clique(const std::vector<p252_comm*>& comms, const uint64_t tag[4], const void* const* d_leaves, size_t leaves_per_ctx,
                       void* const* d_root_out, void* const* hip_streams) {
    const size_t n = comms.size();
    auto stream_of = [&](size_t t) { return (hipStream_t)(hip_streams ? hip_streams[t] : nullptr); };
    int rc = P252_OK;
    for (size_t t = 0; t < n && rc == P252_OK; ++t) {
        (void)hipSetDevice(comms[t]->ctx->device);
        rc = comm_enter(comms[t], stream_of(t));
        if (rc == P252_OK) rc = merkle_tree_device(comms[t]->ctx, 4, tag, d_leaves[t], leaves_per_ctx, comms[t]->d_sub, nullptr, hip_streams ? hip_streams[t] : nullptr);
    }
    if (rc == P252_OK) {  // (one host thread drives every rank: a failure above means NO rank has entered the collective yet)
        ncclResult_t r = R.load()->GroupStart();
        for (size_t t = 0; t < n && r == ncclSuccess; ++t) {
            (void)hipSetDevice(comms[t]->ctx->device);
            r = R.load()->AllGather(comms[t]->d_sub, comms[t]->d_roots, 32, ncclUint8, comms[t]->nccl, stream_of(t));
        }
        const ncclResult_t r2 = R.load()->GroupEnd();
        if (r == ncclSuccess) r = r2;
        if (r != ncclSuccess) rc = fail(comms[0]->ctx, P252_ERR_COMM, std::string("ncclAllGather of the subtree roots: ") + R.load()->GetErrorString(r));
    }
    for (size_t t = 0; t < n && rc == P252_OK; ++t) {
        void* dst = (d_root_out && d_root_out[t]) ? d_root_out[t] : comms[t]->d_top;
        rc = merkle_tree_device(comms[t]->ctx, 4, tag, comms[t]->d_roots, n, dst, nullptr, hip_streams ? hip_streams[t] : nullptr);
        if (rc == P252_OK) rc = comm_leave(comms[t], stream_of(t));
    }
    if (rc != P252_OK && comms[0]->ctx->err.empty()) comms[0]->ctx->err = "sharded tree failed on another context";
    return rc;
}

// used by api.cpp's p252_merkle4_tree_multi_device: RCCL path when the contexts sit on distinct devices (the communicator
// is created on first use and kept), nullptr-result (-> host gather) when they share a device — RCCL refuses two ranks on
// one GPU, and that is the single-GPU test configuration — or when P252_MULTI_HOST_GATHER=1 asks for it.
int tree_multi_device_rccl(p252_ctx* const* ctxs, size_t n_ctx, const uint64_t tag[4], const void* const* d_leaves, size_t leaves_per_ctx,
                           void* const* d_root_out, void* const* hip_streams, bool* used_rccl, std::string* unavailable_why) {
    *used_rccl = false;
    static const bool host_gather = [] {
        const char* e = std::getenv("P252_MULTI_HOST_GATHER");
        return e && e[0] == '1';
    }();
    if (n_ctx == 1 && !(ctxs[0]->comm && !ctxs[0]->comm->owned_by_ctx)) {
        // one context: nothing to exchange — no communicator is created (and the context is not tied to a hidden one);
        // *used_rccl = true only says "the root is where d_root_out asked for it"
        if (d_root_out && d_root_out[0]) {
            int rc = merkle_tree_device(ctxs[0], 4, tag, d_leaves[0], leaves_per_ctx, d_root_out[0], nullptr, hip_streams ? hip_streams[0] : nullptr);
            if (rc) return rc;
        }
        *used_rccl = true;
        return P252_OK;
    }
    std::vector<p252_comm*> comms = clique_of(ctxs, n_ctx);
    if (comms.empty()) {
        if (host_gather || !distinct_devices(ctxs, n_ctx)) return P252_OK;
        static std::mutex mu;  // one lazy creation at a time
        std::lock_guard<std::mutex> lk(mu);
        // a lazy creation that failed once is not retried on every call (ncclCommInitAll takes 0.1-1 s to refuse): the reason
        // is kept for p252_merkle4_tree_multi_device_resident's message, the caller of this function gathers through the host
        const std::string refused = lazy_refused();
        if (!refused.empty()) {
            if (unavailable_why) *unavailable_why = refused;
            return P252_OK;
        }
        // ABI 5 accepted ANY array of contexts.  A context that already sits in a communicator over ANOTHER array (other count,
        // order or subset — ctxs[:4] after ctxs[:8]) must not make the call fail:
        //  * a communicator the CALLER made (p252_comm_create_rank / _create_all) is the caller's to keep: the roots are gathered
        //    through the host instead (P252_OK with *used_rccl = false);
        //  * communicators this entry point made itself (owned by their contexts) are torn down — every member of each such
        //    clique, RCCL communicators go collectively — and one over the new array is created.
        for (size_t t = 0; t < n_ctx; ++t)
            if (ctxs[t]->comm && !ctxs[t]->comm->owned_by_ctx) return P252_OK;
        for (size_t t = 0; t < n_ctx; ++t)
            if (ctxs[t]->comm) {
                const auto clique = ctxs[t]->comm->clique;  // (keeps the list alive while its members go)
                const std::vector<p252_comm*> members = clique ? *clique : std::vector<p252_comm*>{ctxs[t]->comm};
                for (p252_comm* m : members)
                    if (m && m->ctx) {
                        (void)hipSetDevice(m->ctx->device);
                        (void)hipDeviceSynchronize();  // nothing of the old clique may still be queued
                    }
                for (p252_comm* m : members) free_comm(m, false);
            }
        int rc = create_all(ctxs, n_ctx, comms, /*owned=*/true);
        if (rc) {  // no communicator to be had (no RCCL in the process, ncclCommInitAll refused): the host gather still exists — the caller takes it
            const std::string why = ctxs[0]->err.empty() ? std::string("communicator creation failed") : ctxs[0]->err;
            set_lazy_refused(why);
            if (unavailable_why) *unavailable_why = why;
            ctxs[0]->err.clear();
            return P252_OK;
        }
    }
    *used_rccl = true;
    return tree_clique(comms, tag, d_leaves, leaves_per_ctx, d_root_out, hip_streams);
}

}  // namespace p252host

extern "C" {

int p252_comm_unique_id(void* id_out, size_t len) {
    if (!id_out || len != P252_COMM_ID_BYTES) return fail(nullptr, P252_ERR_INVALID_ARGUMENT, "comm_unique_id: id_out must hold P252_COMM_ID_BYTES bytes");
    int rc = need_rccl(nullptr);
    if (rc) return rc;
    ncclUniqueId id;
    NCCL_TRY(nullptr, R.load()->GetUniqueId(&id));
    std::memcpy(id_out, id.internal, P252_COMM_ID_BYTES);
    return P252_OK;
}

int p252_comm_create_rank(p252_ctx* ctx, const void* id, size_t len, int rank, int world, p252_comm** out) {
    if (!ctx || !out) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "comm_create_rank: NULL argument");
    *out = nullptr;
    if (!id || len != P252_COMM_ID_BYTES) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "comm_create_rank: id must be the P252_COMM_ID_BYTES bytes of p252_comm_unique_id");
    if (world < 1 || rank < 0 || rank >= world) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "comm_create_rank: need 0 <= rank < world");
    if (ctx->comm) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "comm_create_rank: the context already belongs to a communicator");
    int rc = need_rccl(ctx);
    if (rc) return rc;
    // every allocation BEFORE the first collective: a rank that cannot allocate returns here, while its peers have not yet entered
    // anything they would wait in for it; from ncclCommInitRank on, creation is collective — a failure on one rank (reported on
    // that rank) aborts its communicator, and the job must treat it as the failure of all (header: "collective")
    p252_comm* c = new p252_comm();
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    rc = alloc_buffers(c);
    if (rc == P252_OK) {
        ncclUniqueId uid;
        std::memcpy(uid.internal, id, P252_COMM_ID_BYTES);
        const ncclResult_t r = R.load()->CommInitRank(&c->nccl, world, uid, rank);
        if (r != ncclSuccess) rc = fail(ctx, P252_ERR_COMM, std::string("ncclCommInitRank: ") + R.load()->GetErrorString(r));
    }
    if (rc == P252_OK) {
        ctx->comm = c;
        rc = broadcast_and_validate(std::vector<p252_comm*>{c}, 0);
    }
    if (rc != P252_OK) {
        const std::string msg = ctx->err;
        free_comm(c, true);
        ctx->err = msg;
        return rc;
    }
    *out = c;
    return P252_OK;
}

int p252_comm_create_all(p252_ctx* const* ctxs, size_t n_ctx, p252_comm** comms_out) {
    int rc = check_ctxs(ctxs, n_ctx);
    if (rc) return rc;
    if (!comms_out) return fail(ctxs[0], P252_ERR_INVALID_ARGUMENT, "comm_create_all: comms_out is NULL");
    std::vector<p252_comm*> v;
    rc = create_all(ctxs, n_ctx, v, /*owned=*/false);
    for (size_t t = 0; t < n_ctx; ++t) comms_out[t] = rc == P252_OK ? v[t] : nullptr;
    return rc;
}

void p252_comm_destroy(p252_comm* comm) { free_comm(comm, false); }

int p252_comm_check(p252_comm* comm, void* hip_stream) {
    if (!comm || !comm->ctx) return P252_ERR_INVALID_ARGUMENT;
    p252_ctx* ctx = comm->ctx;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize((hipStream_t)hip_stream));
    return comm_take_failure(ctx);
}

int p252_comm_backend(char* path_out, size_t len) {
    const int rc = need_rccl(nullptr);
    if (rc) return rc;
    if (path_out && len) {
        std::strncpy(path_out, R.load()->origin.c_str(), len - 1);
        path_out[len - 1] = 0;
    }
    return P252_OK;
}

int p252_comm_rank(const p252_comm* comm) { return comm ? comm->rank : -1; }
int p252_comm_size(const p252_comm* comm) { return comm ? comm->world : 0; }

int p252_merkle4_tree_sharded_device(p252_comm* comm, const uint64_t tag[4], const void* d_leaves, size_t n_leaves_local, void* d_root,
                                     void* hip_stream) {
    if (!comm || !comm->ctx) return P252_ERR_INVALID_ARGUMENT;
    p252_ctx* ctx = comm->ctx;
    if (!tag || !d_leaves || !d_root) return fail(ctx, P252_ERR_INVALID_ARGUMENT, "merkle_tree_sharded: NULL buffer");
    if (!power_of_4(n_leaves_local))
        return fail(ctx, P252_ERR_INVALID_ARGUMENT, "merkle_tree_sharded: every rank must own a complete subtree (4^k leaves)");
    hipStream_t st = (hipStream_t)hip_stream;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = comm_enter(comm, st);
    // the rank's subtree (zero communication) ...
    if (rc == P252_OK) rc = merkle_tree_device(ctx, 4, tag, d_leaves, n_leaves_local, comm->d_sub, nullptr, hip_stream);
    if (rc) {
        // A LOCAL failure (allocation of the level scratch) must not leave the peers blocked on the stream in a collective this
        // rank never enters: the rank still contributes — an all-ones root, which is no BlsScalar (>= p) — and returns
        // its error.  Its peers find the sentinel on the device (k_poison_if_peer_failed below): their root becomes all-ones too and
        // p252_comm_check / p252_sync report P252_ERR_COMM naming this rank (they used to return a garbage root as P252_OK).
        const std::string msg = ctx->err;
        (void)hipMemsetAsync(comm->d_sub, 0xff, 32, st);
        (void)R.load()->AllGather(comm->d_sub, comm->d_roots, 32, ncclUint8, comm->nccl, st);
        (void)comm_leave(comm, st);
        ctx->err = msg;
        return rc;
    }
    // ... the path's only exchange step, on the same stream: world x 32 bytes to every rank ...
    NCCL_TRY(ctx, R.load()->AllGather(comm->d_sub, comm->d_roots, 32, ncclUint8, comm->nccl, st));
    // ... and the top levels, on every rank (a single rank's "tree over one root" is a copy)
    rc = merkle_tree_device(ctx, 4, tag, comm->d_roots, (size_t)comm->world, d_root, nullptr, hip_stream);
    if (rc == P252_OK && p252::launch_poison_if_peer_failed(comm->d_roots, (unsigned)comm->world, d_root, comm->d_fail, st) != hipSuccess)
        rc = fail(ctx, P252_ERR_HIP, "merkle_tree_sharded: launching the failed-peer check failed");
    const int rc2 = comm_leave(comm, st);
    return rc ? rc : rc2;
}

int p252_merkle4_tree_multi_device_resident(p252_ctx* const* ctxs, size_t n_ctx, const uint64_t tag[4], const void* const* d_leaves,
                                            size_t leaves_per_ctx, void* const* d_root_out, void* const* hip_streams) {
    int rc = check_ctxs(ctxs, n_ctx);
    if (rc) return rc;
    if (!tag || !d_leaves) return fail(ctxs[0], P252_ERR_INVALID_ARGUMENT, "merkle_tree_multi_device_resident: NULL buffer");
    if (!power_of_4(leaves_per_ctx))
        return fail(ctxs[0], P252_ERR_INVALID_ARGUMENT, "merkle_tree_multi_device_resident: every device must own a complete subtree (4^k leaves)");
    bool used = false;
    std::string why;
    rc = tree_multi_device_rccl(ctxs, n_ctx, tag, d_leaves, leaves_per_ctx, d_root_out, hip_streams, &used, &why);
    if (rc) return rc;
    if (!used && !why.empty())  // RCCL itself said no (or is not in the process): its own words, not a guess
        return fail(ctxs[0], P252_ERR_COMM, "merkle_tree_multi_device_resident: no RCCL communicator over the contexts: " + why +
                                                 " (p252_merkle4_tree_multi_device gathers through the host in that case)");
    if (!used)
        return fail(ctxs[0], P252_ERR_COMM,
                    "merkle_tree_multi_device_resident needs an RCCL communicator over the contexts: they share a device (RCCL wants one per rank), "
                    "one of them belongs to a communicator the caller made over another array, or P252_MULTI_HOST_GATHER=1 is set; "
                    "p252_merkle4_tree_multi_device gathers through the host in that case");
    return P252_OK;
}

}  // extern "C"
