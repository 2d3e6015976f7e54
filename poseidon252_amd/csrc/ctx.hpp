// ctx.hpp — the context object and the few host-side helpers shared by api.cpp (the C ABI) and comm.cpp (the RCCL
// communicator of the multi-GPU entry points).  Internal: nothing here is part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/poseidon252_hip.h"

struct p252_comm;

struct p252_ctx {
    int device = -1;
    int32_t* d_tab = nullptr;
    std::vector<int32_t> h_tab;
    // grow-only scratch for the host-buffer entry points and the tree builder
    void* d_in = nullptr;
    size_t d_in_cap = 0;
    void* d_out = nullptr;
    size_t d_out_cap = 0;
    // Root-only tree / forest builds ping-pong two level buffers.  The `_device` builders are asynchronous on a caller-chosen
    // stream, so ONE pair per context would be shared by builds queued on different streams (round 4: silent wrong roots).  Each
    // caller stream therefore owns a pair (up to MAX_LEVEL_SETS); a further stream takes over the least recently used pair after
    // waiting on the event recorded behind that pair's last build — builds on different streams overlap, none ever shares scratch.
    struct LevelSet {
        hipStream_t st = nullptr;
        void* buf[2] = {nullptr, nullptr};
        size_t cap[2] = {0, 0};
        hipEvent_t done = nullptr;  // recorded on `st` behind the last launch that touches buf[]
        uint64_t stamp = 0;         // use counter value of the last build (least recently used = smallest)
    };
    static constexpr size_t MAX_LEVEL_SETS = 4;
    std::vector<LevelSet> lvl;
    uint64_t lvl_clock = 0;
    hipStream_t streams[3] = {nullptr, nullptr, nullptr};  // host-buffer pipeline over caller-pinned memory (created on first use)
    // host-buffer pipeline over PAGEABLE caller memory: library-owned page-locked staging, one lane per worker thread
    // (stream + pinned in/out chunk + device in/out chunk), created on first use and kept
    struct Slot {  // one chunk in flight: page-locked staging pair, device pair, completion event
        void* h_in = nullptr;
        void* h_out = nullptr;
        void* d_in = nullptr;
        void* d_out = nullptr;
        size_t in_cap = 0, out_cap = 0;
        size_t in_dirty = 0, out_dirty = 0;  // bytes calls may have written since the last wipe (<= cap): what a per-call wipe clears
        hipEvent_t done = nullptr;
    };
    struct Lane {  // one worker thread + stream, double-buffered: the host copy of chunk c+1 overlaps the DMA / kernel of chunk c
        hipStream_t st = nullptr;
        Slot slot[2];
    };
    std::vector<Lane> lanes;
    int lane_budget = 0;  // > 0: staging lanes this call may use (set by the p252_*_multi drivers, which share the CPU quota)
    // encryption: the sponge-call program of the last (variant, message_len) used, uploaded once (k_crypt interprets it)
    uint32_t* d_prog = nullptr;
    size_t d_prog_cap = 0;
    int prog_variant = -1;
    size_t prog_len = 0;
    unsigned prog_calls = 0;
    p252_comm* comm = nullptr;  // the communicator this context is a rank of, if any (comm.cpp; not owned)
    std::string err;
};


namespace p252host {
int fail(p252_ctx* ctx, int code, const std::string& msg);
int ensure(p252_ctx* ctx, void** buf, size_t* cap, size_t need);
// the level-scratch pair of a root-only build on `st` (at least need0 / need1 bytes); level_set_done() after the build's last launch
int level_set(p252_ctx* ctx, hipStream_t st, size_t need0, size_t need1, p252_ctx::LevelSet** out);
int level_set_done(p252_ctx* ctx, p252_ctx::LevelSet* set);
const std::vector<int32_t>& host_tables();
bool power_of_4(size_t v);
int check_ctxs(p252_ctx* const* ctxs, size_t n_ctx);
// the level-by-level tree build on one device (api.cpp): asynchronous on hip_stream, root (32 B) written to d_root
int merkle_tree_device(p252_ctx* ctx, unsigned arity, const uint64_t tag[4], const void* d_leaves, size_t n_leaves, void* d_root,
                       void* d_levels, void* hip_stream);
// comm.cpp
void release_ctx_comm(p252_ctx* ctx);  // p252_destroy: a communicator the library created for this context goes with it
// the sharded tree over an array of contexts through RCCL (communicator created on first use); *used_rccl = false and
// P252_OK when the contexts cannot form one (shared device) or P252_MULTI_HOST_GATHER=1: the caller gathers through the host
int tree_multi_device_rccl(p252_ctx* const* ctxs, size_t n_ctx, const uint64_t tag[4], const void* const* d_leaves, size_t leaves_per_ctx,
                           void* const* d_root_out, void* const* hip_streams, bool* used_rccl, std::string* unavailable_why = nullptr);
// a peer's failed local build seen by this context's communicator since the last call (comm.cpp): P252_ERR_COMM once, then cleared
int comm_take_failure(p252_ctx* ctx);
}  // namespace p252host

#define HIP_TRY(ctx, expr)                                                                  \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return p252host::fail(ctx, P252_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
