// th_comm.h — the multi-GPU side of the C ABI (include/tracehip.h "multi-GPU"): one process per GPU, collectives through RCCL.
// RCCL is dlopen'ed on first use instead of linked: a process that already holds a copy (PyTorch ships its own librccl.so with the
// same soname) keeps exactly one, and hosts without RCCL can still load libtracehip.so for single-GPU work.
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <mutex>
#include <string>

namespace th {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

inline RcclApi* rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (api.handle) break;
        }
        if (!api.handle) {
            const char* e = dlerror();
            api.error = std::string("cannot load librccl.so: ") + (e ? e : "unknown error");
            return;
        }
        auto sym = [&](const char* name) {
            void* p = dlsym(api.handle, name);
            if (!p && api.error.empty()) api.error = std::string("librccl.so lacks ") + name;
            return p;
        };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.Reduce = (decltype(api.Reduce))sym("ncclReduce");
        api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    });
    return &api;
}

struct Comm {
    ncclComm_t comm = nullptr;
    int rank = 0, n_ranks = 1;
};

}  // namespace th
