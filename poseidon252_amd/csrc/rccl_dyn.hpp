// rccl_dyn.hpp — RCCL resolved at run time, on the first call that needs it (comm.cpp).  libposeidon252_hip.so does NOT link librccl:
// a single-GPU hashing deployment loads and runs without RCCL installed, and a process that already holds a copy (PyTorch-ROCm
// bundles one under the SONAME librccl.so.1; an application may link its own) gets THAT copy — one RCCL per process, which is the
// state to be in at first contact with several real ranks.  <rccl/rccl.h> is included for its types only.
//
// Search order (rccl_dyn.cpp), first hit wins:
//   1. $P252_RCCL_PATH                     an explicit file; if set, nothing else is tried (the test suite's mock comes in this way)
//   2. the process's global symbol scope   dlsym(RTLD_DEFAULT): the application linked or LD_PRELOADed RCCL
//   3. a copy already mapped               dlopen(RTLD_NOLOAD) of librccl.so.1 / librccl.so — matches by SONAME, so torch's bundled
//                                          copy (loaded RTLD_LOCAL under `import torch`) is found here
//   4. the loader's search path            dlopen("librccl.so.1"), dlopen("librccl.so")
//   5. $ROCM_PATH/lib, /opt/rocm/lib       librccl.so.1 by absolute path
// Nothing found: p252_comm_* and the RCCL path of the *_multi_device entry points return P252_ERR_COMM with this list in
// p252_last_error; every other entry point is unaffected.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <string>

namespace p252rccl {

struct Api {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string origin;  // where the ten symbols came from (a path, or "the process's global scope")
};

// the resolved table, or nullptr with *why = what was tried and what each attempt said.  A success is kept for the life of the
// process; a failure is not (the next call searches again: the host program may have loaded RCCL in between).  Thread-safe.
const Api* api(std::string* why);

}  // namespace p252rccl
