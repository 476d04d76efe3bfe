// RCCL through the C ABI (include/bark_hip.h, "multi-GPU"): the two exchanges of the path — the all-gather of the
// per-rank MLL blocks (forest.py:92-98 evaluates independent samples, so nothing else crosses GPUs) and
// the all-reduce of the mixture moments of the posterior (tree_gps.py:116-131) — as plain entry points, so that a
// binding needs no torch.distributed for them.  librccl is resolved at run time (dlopen, no link-time dependency: the
// library must load on a box without RCCL, and inside a torch process the already loaded copy is the one found).
#include <dlfcn.h>

#include <mutex>

#include "common.h"

namespace {

// the few declarations of <rccl/rccl.h> this file needs (ABI-stable since NCCL 2: opaque communicator, 128-byte id,
// enum values below)
struct UniqueId {
    char internal[128];
};
typedef void *Comm;
typedef int Result;  // 0 == ncclSuccess
constexpr int kFloat64 = 8, kSum = 0, kMax = 2;

struct Api {
    void *handle = nullptr;
    Result (*GetUniqueId)(UniqueId *) = nullptr;
    Result (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
    Result (*CommDestroy)(Comm) = nullptr;
    Result (*AllGather)(const void *, void *, size_t, int, Comm, hipStream_t) = nullptr;
    Result (*AllReduce)(const void *, void *, size_t, int, int, Comm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(Result) = nullptr;
    bool ok = false;
};

Api &api() {
    static Api a;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            a.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (a.handle) break;
        }
        if (!a.handle) return;
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(a.handle, "ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(a.handle, "ncclCommInitRank"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(a.handle, "ncclCommDestroy"));
        a.AllGather = reinterpret_cast<decltype(a.AllGather)>(dlsym(a.handle, "ncclAllGather"));
        a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(a.handle, "ncclAllReduce"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(a.handle, "ncclGetErrorString"));
        a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllGather && a.AllReduce;
    });
    return a;
}

int need_api() {
    if (!api().ok) return bark::fail(BARK_ERR_HIP, "RCCL is not available (librccl.so could not be loaded)");
    return BARK_OK;
}

int check(Result r, const char *what) {
    if (r == 0) return BARK_OK;
    return bark::fail(BARK_ERR_HIP, "%s failed: %s", what, api().GetErrorString ? api().GetErrorString(r) : "RCCL error");
}

}  // namespace

extern "C" {

int bark_comm_unique_id(void *id_out) {
    bark::error_buffer()[0] = 0;
    if (!id_out) return bark::fail(BARK_ERR_ARG, "bark_comm_unique_id: null output");
    int rc = need_api();
    if (rc) return rc;
    return check(api().GetUniqueId(static_cast<UniqueId *>(id_out)), "ncclGetUniqueId");
}

int bark_comm_create(const void *id, int rank, int world, int device, void **comm_out) {
    bark::error_buffer()[0] = 0;
    if (!id || !comm_out || world < 1 || rank < 0 || rank >= world) return bark::fail(BARK_ERR_ARG, "bark_comm_create: bad argument");
    *comm_out = nullptr;
    int rc = need_api();
    if (rc) return rc;
    int prev = 0;
    BARK_HIP_CHECK(hipGetDevice(&prev));
    BARK_HIP_CHECK(hipSetDevice(device));
    Comm c = nullptr;
    rc = check(api().CommInitRank(&c, world, *static_cast<const UniqueId *>(id), rank), "ncclCommInitRank");
    (void)hipSetDevice(prev);  // the caller's current device is not this function's to change (as bark_ctx_create)
    if (rc) return rc;
    *comm_out = c;
    return BARK_OK;
}

void bark_comm_destroy(void *comm) {
    if (comm && api().ok) (void)api().CommDestroy(static_cast<Comm>(comm));
}

int bark_allgather_mll(void *comm, const double *local, int64_t n_local, double *out, void *stream) {
    bark::error_buffer()[0] = 0;
    if (!comm || !local || !out || n_local < 1) return bark::fail(BARK_ERR_ARG, "bark_allgather_mll: bad argument");
    int rc = need_api();
    if (rc) return rc;
    return check(api().AllGather(local, out, (size_t)n_local, kFloat64, static_cast<Comm>(comm), static_cast<hipStream_t>(stream)),
                 "ncclAllGather");
}

int bark_allreduce_f64(void *comm, double *buf, int64_t n, int op_max, void *stream) {
    bark::error_buffer()[0] = 0;
    if (!comm || !buf || n < 1) return bark::fail(BARK_ERR_ARG, "bark_allreduce_f64: bad argument");
    int rc = need_api();
    if (rc) return rc;
    return check(api().AllReduce(buf, buf, (size_t)n, kFloat64, op_max ? kMax : kSum, static_cast<Comm>(comm),
                                 static_cast<hipStream_t>(stream)),
                 "ncclAllReduce");
}

}  // extern "C"
