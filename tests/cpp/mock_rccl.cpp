// mock_rccl.cpp — TEST-ONLY stand-in for the ten RCCL entry points csrc/comm.cpp calls, so that the library's multi-rank logic
// (communicator creation, the validated broadcast of the constants, the all-gather of subtree roots, the top levels on every
// rank, teardown) can run with MORE THAN ONE RANK on a box with ONE GPU.  Real RCCL refuses two ranks on a device and no
// multi-GPU node was ever available to the builder; the suite covers the real backend at one rank (tests/test_comm_forest.py,
// tests/c/abi_smoke.c) and the rank logic through this mock (tests/test_comm_mock_ranks.py), which links the library's own
// objects against this file instead of librccl.  It is not shipped and not part of the product library.
//
// Semantics implemented (what comm.cpp relies on): a communicator = (world, rank); a collective completes when every rank of
// the world has issued it — ranks may be driven by one thread inside ncclGroupStart / ncclGroupEnd (ncclCommInitAll cliques) or
// by one thread each (ncclCommInitRank); the data movement is done with hipMemcpy after synchronising every rank's stream, so
// on return the result is visible to whatever the caller enqueues next (stronger than RCCL's stream ordering, never weaker).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct Op {
    int kind = 0;  // 1 = all-gather, 2 = broadcast
    const void* send = nullptr;
    void* recv = nullptr;
    size_t bytes = 0;
    int root = 0;
    hipStream_t st = nullptr;
};

struct World {
    int n = 0;
    std::mutex mu;
    std::condition_variable cv;
    int registered = 0, posted = 0, alive = 0;
    unsigned long long gen = 0;
    std::vector<Op> ops;
    std::vector<int> dev;
};

}  // namespace

struct ncclComm {
    World* w;
    int rank;
};

namespace {

std::mutex g_mu;
std::map<std::string, World*> g_worlds;
std::atomic<unsigned long long> g_uid{1};

size_t type_size(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        default: return 8;
    }
}

ncclResult_t execute(World* w) {  // called with w->mu held, by the rank that completed the round
    for (int r = 0; r < w->n; ++r) {
        if (hipSetDevice(w->dev[r]) != hipSuccess || hipStreamSynchronize(w->ops[r].st) != hipSuccess) return ncclUnhandledCudaError;
        if (w->ops[r].kind != w->ops[0].kind || w->ops[r].bytes != w->ops[0].bytes) return ncclInvalidUsage;  // mismatched collectives
    }
    for (int t = 0; t < w->n; ++t) {
        char* dst = static_cast<char*>(w->ops[t].recv);
        if (w->ops[0].kind == 1) {
            for (int r = 0; r < w->n; ++r)
                if (hipMemcpy(dst + (size_t)r * w->ops[r].bytes, w->ops[r].send, w->ops[r].bytes, hipMemcpyDeviceToDevice) != hipSuccess) return ncclUnhandledCudaError;
        } else {
            const int root = w->ops[t].root;
            if (root != w->ops[0].root || root < 0 || root >= w->n) return ncclInvalidArgument;
            if (hipMemcpy(dst, w->ops[root].send, w->ops[t].bytes, hipMemcpyDeviceToDevice) != hipSuccess) return ncclUnhandledCudaError;
        }
    }
    return hipDeviceSynchronize() == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

struct Pending {
    ncclComm* c;
    Op op;
};
thread_local int t_depth = 0;
thread_local std::vector<Pending> t_pending;
thread_local ncclResult_t t_err = ncclSuccess;

// post every pending op of this thread, then wait for the rounds that are still open
ncclResult_t flush() {
    std::vector<std::pair<World*, unsigned long long>> waits;
    ncclResult_t rc = ncclSuccess;
    for (Pending& p : t_pending) {
        World* w = p.c->w;
        std::unique_lock<std::mutex> lk(w->mu);
        w->ops[p.c->rank] = p.op;
        if (++w->posted == w->n) {
            const ncclResult_t r = execute(w);
            if (r != ncclSuccess) rc = r;
            w->posted = 0;
            ++w->gen;
            w->cv.notify_all();
        } else {
            waits.push_back({w, w->gen});
        }
    }
    t_pending.clear();
    for (auto& wg : waits) {
        std::unique_lock<std::mutex> lk(wg.first->mu);
        wg.first->cv.wait(lk, [&] { return wg.first->gen != wg.second; });
    }
    return rc;
}

ncclResult_t issue(ncclComm* c, const Op& op) {
    if (!c) return ncclInvalidArgument;
    t_pending.push_back({c, op});
    return t_depth > 0 ? ncclSuccess : flush();
}

}  // namespace

extern "C" {

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "mock rccl error"; }

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    std::memset(id->internal, 0, sizeof id->internal);
    const unsigned long long v = g_uid.fetch_add(1);
    std::memcpy(id->internal, "MOCKRCCL", 8);
    std::memcpy(id->internal + 8, &v, sizeof v);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || rank < 0 || rank >= nranks || std::memcmp(id.internal, "MOCKRCCL", 8) != 0) return ncclInvalidArgument;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return ncclUnhandledCudaError;
    World* w;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        World*& slot = g_worlds[std::string(id.internal, sizeof id.internal)];
        if (!slot) {
            slot = new World();
            slot->n = nranks;
            slot->ops.resize(nranks);
            slot->dev.assign(nranks, 0);
        }
        w = slot;
    }
    std::unique_lock<std::mutex> lk(w->mu);
    if (w->n != nranks) return ncclInvalidArgument;
    w->dev[rank] = dev;
    ++w->alive;
    if (++w->registered == nranks)
        w->cv.notify_all();
    else
        w->cv.wait(lk, [&] { return w->registered >= nranks; });  // collective: returns when every rank has joined
    *comm = new ncclComm{w, rank};
    return ncclSuccess;
}

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
    if (!comms || ndev < 1) return ncclInvalidArgument;
    World* w = new World();
    w->n = ndev;
    w->ops.resize(ndev);
    w->dev.resize(ndev);
    w->registered = w->alive = ndev;
    for (int r = 0; r < ndev; ++r) {
        w->dev[r] = devlist ? devlist[r] : r;
        comms[r] = new ncclComm{w, r};
    }
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    if (!comm) return ncclSuccess;
    World* w = comm->w;
    bool last;
    {
        std::lock_guard<std::mutex> lk(w->mu);
        last = --w->alive == 0;
    }
    delete comm;
    if (last) {
        std::lock_guard<std::mutex> lk(g_mu);
        for (auto it = g_worlds.begin(); it != g_worlds.end(); ++it)
            if (it->second == w) {
                g_worlds.erase(it);
                break;
            }
        delete w;
    }
    return ncclSuccess;
}
ncclResult_t ncclCommAbort(ncclComm_t comm) { return ncclCommDestroy(comm); }

ncclResult_t ncclGroupStart() {
    ++t_depth;
    return ncclSuccess;
}
ncclResult_t ncclGroupEnd() {
    if (t_depth <= 0) return ncclInvalidUsage;
    return --t_depth == 0 ? flush() : ncclSuccess;
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream) {
    Op op;
    op.kind = 1;
    op.send = sendbuff;
    op.recv = recvbuff;
    op.bytes = sendcount * type_size(datatype);
    op.st = stream;
    return issue(comm, op);
}

ncclResult_t ncclBroadcast(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, int root, ncclComm_t comm, hipStream_t stream) {
    Op op;
    op.kind = 2;
    op.send = sendbuff;
    op.recv = recvbuff;
    op.bytes = count * type_size(datatype);
    op.root = root;
    op.st = stream;
    return issue(comm, op);
}

}  // extern "C"
