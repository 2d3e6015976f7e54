// rccl_dyn.cpp — see rccl_dyn.hpp: the ten RCCL entry points comm.cpp calls, looked up with dlopen / dlsym on first use.
#include "rccl_dyn.hpp"

#include <dlfcn.h>

#include <cstdlib>
#include <mutex>

namespace p252rccl {

namespace {

// every symbol out of ONE object (a table mixed from two copies of RCCL would be worse than none)
bool fill(void* h, Api& a, std::string& missing) {
    struct {
        const char* name;
        void** slot;
    } syms[] = {
        {"ncclGetUniqueId", (void**)&a.GetUniqueId},   {"ncclCommInitRank", (void**)&a.CommInitRank}, {"ncclCommInitAll", (void**)&a.CommInitAll},
        {"ncclCommDestroy", (void**)&a.CommDestroy},   {"ncclCommAbort", (void**)&a.CommAbort},       {"ncclGroupStart", (void**)&a.GroupStart},
        {"ncclGroupEnd", (void**)&a.GroupEnd},         {"ncclBroadcast", (void**)&a.Broadcast},       {"ncclAllGather", (void**)&a.AllGather},
        {"ncclGetErrorString", (void**)&a.GetErrorString},
    };
    for (auto& s : syms) {
        *s.slot = dlsym(h, s.name);
        if (!*s.slot) {
            missing = s.name;
            return false;
        }
    }
    return true;
}

std::string path_of(void* fn) {
    Dl_info info;
    return (dladdr(fn, &info) && info.dli_fname) ? std::string(info.dli_fname) : std::string("?");
}

}  // namespace

const Api* api(std::string* why) {
    static std::mutex mu;
    static Api table;
    static bool have = false;
    std::lock_guard<std::mutex> lk(mu);
    if (have) return &table;
    std::string tried;
    auto try_handle = [&](void* h, const std::string& what, bool global_scope = false) {  // (RTLD_DEFAULT is a null handle)
        if (!h && !global_scope) {
            const char* e = dlerror();
            tried += "\n  " + what + ": " + (e ? e : "not loaded");
            return false;
        }
        Api a;
        std::string missing;
        if (!fill(h, a, missing)) {
            tried += "\n  " + what + ": no symbol " + missing;
            return false;
        }
        a.origin = path_of((void*)a.GetUniqueId);
        table = a;
        have = true;
        return true;
    };
    const char* explicit_path = std::getenv("P252_RCCL_PATH");
    if (explicit_path && explicit_path[0]) {
        try_handle(dlopen(explicit_path, RTLD_NOW | RTLD_LOCAL), std::string("P252_RCCL_PATH=") + explicit_path);
    } else {
        bool ok = false;
        if (dlsym(RTLD_DEFAULT, "ncclGetUniqueId")) ok = try_handle(RTLD_DEFAULT, "the process's global symbol scope", true);
        else tried += "\n  the process's global symbol scope: no ncclGetUniqueId";
        for (const char* name : {"librccl.so.1", "librccl.so"})
            if (!ok) ok = try_handle(dlopen(name, RTLD_NOLOAD | RTLD_NOW), std::string("already mapped ") + name);
        for (const char* name : {"librccl.so.1", "librccl.so"})
            if (!ok) ok = try_handle(dlopen(name, RTLD_NOW | RTLD_LOCAL), std::string("dlopen ") + name);
        for (const char* prefix : {(const char*)std::getenv("ROCM_PATH"), "/opt/rocm"})
            if (!ok && prefix && prefix[0]) {
                const std::string p = std::string(prefix) + "/lib/librccl.so.1";
                ok = try_handle(dlopen(p.c_str(), RTLD_NOW | RTLD_LOCAL), "dlopen " + p);
            }
    }
    if (have) return &table;
    if (why) *why = "RCCL is not available to this process (libposeidon252_hip.so resolves it on first use; P252_RCCL_PATH names a file explicitly); tried:" + tried;
    return nullptr;
}

}  // namespace p252rccl
