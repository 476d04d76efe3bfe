// bark_ctx: the per-device context of the C ABI (include/bark_hip.h).
//
// The reference's callers are single-threaded Python; the library is re-entrant per context: helper
// streams, events, the scratch buffer and the categorical-fault flag belong to a bark_ctx, created and destroyed
// explicitly.  The only process-wide state left is the thread-local last-error buffer and the once-per-device
// dynamic-LDS attributes of the kernels (std::call_once).
#include "common.h"

namespace bark {

int check_ctx(const bark_ctx *ctx) {
    if (!ctx) return fail(BARK_ERR_ARG, "null bark_ctx (create one per device with bark_ctx_create)");
    int dev = -1;
    BARK_HIP_CHECK(hipGetDevice(&dev));
    if (dev != ctx->device)
        return fail(BARK_ERR_ARG, "bark_ctx belongs to device %d but the current device is %d", ctx->device, dev);
    return BARK_OK;
}

int ctx_events(bark_ctx *ctx, size_t n) {
    if (!ctx->helper) {
        int lo = 0, hi = 0;
        BARK_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        BARK_HIP_CHECK(hipStreamCreateWithPriority(&ctx->helper, hipStreamNonBlocking, lo));
    }
    if (!ctx->helper2) {
        int lo = 0, hi = 0;
        BARK_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        BARK_HIP_CHECK(hipStreamCreateWithPriority(&ctx->helper2, hipStreamNonBlocking, lo));
    }
    if (!ctx->helper3) {
        int lo = 0, hi = 0;
        BARK_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        BARK_HIP_CHECK(hipStreamCreateWithPriority(&ctx->helper3, hipStreamNonBlocking, lo));
    }
    while (ctx->events.size() < n) {
        hipEvent_t e;
        BARK_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->events.push_back(e);
    }
    return BARK_OK;
}

int ctx_chain_streams(bark_ctx *ctx, size_t n) {
    if (!ctx->chain_fork) BARK_HIP_CHECK(hipEventCreateWithFlags(&ctx->chain_fork, hipEventDisableTiming));
    while (ctx->chain_streams.size() < n) {
        hipStream_t st;
        hipEvent_t ev;
        BARK_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        BARK_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        ctx->chain_streams.push_back(st);
        ctx->chain_done.push_back(ev);
    }
    return BARK_OK;
}

}  // namespace bark

using namespace bark;

extern "C" {

int bark_ctx_create(int device, bark_ctx **out) {
    error_buffer()[0] = 0;
    if (!out) return fail(BARK_ERR_ARG, "bark_ctx_create: null output");
    *out = nullptr;
    int count = 0;
    BARK_HIP_CHECK(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) return fail(BARK_ERR_ARG, "bark_ctx_create: device %d of %d", device, count);
    int prev = 0;
    BARK_HIP_CHECK(hipGetDevice(&prev));
    BARK_HIP_CHECK(hipSetDevice(device));
    bark_ctx *ctx = new bark_ctx();
    ctx->device = device;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&ctx->fault), 256);
    if (e == hipSuccess) e = hipMemset(ctx->fault, 0, 256);
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void **>(&ctx->fault_host), 256, hipHostMallocDefault);
    (void)hipSetDevice(prev);
    if (e != hipSuccess) {
        if (ctx->fault) (void)hipFree(ctx->fault);
        delete ctx;
        return fail(BARK_ERR_HIP, "bark_ctx_create: %s", hipGetErrorString(e));
    }
    *out = ctx;
    return BARK_OK;
}

// Frees everything the context owns.  The caller must have synchronised the streams it used with this context.
void bark_ctx_destroy(bark_ctx *ctx) {
    if (!ctx) return;
    int prev = 0;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(ctx->device);
    if (ctx->helper) {
        (void)hipStreamSynchronize(ctx->helper);
        (void)hipStreamDestroy(ctx->helper);
    }
    if (ctx->helper2) {
        (void)hipStreamSynchronize(ctx->helper2);
        (void)hipStreamDestroy(ctx->helper2);
    }
    if (ctx->helper3) {
        (void)hipStreamSynchronize(ctx->helper3);
        (void)hipStreamDestroy(ctx->helper3);
    }
    for (hipEvent_t e : ctx->events) (void)hipEventDestroy(e);
    for (hipStream_t s : ctx->chain_streams) {
        (void)hipStreamSynchronize(s);
        (void)hipStreamDestroy(s);
    }
    for (hipEvent_t e : ctx->chain_done) (void)hipEventDestroy(e);
    if (ctx->chain_fork) (void)hipEventDestroy(ctx->chain_fork);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->fault) (void)hipFree(ctx->fault);
    if (ctx->fault_host) (void)hipHostFree(ctx->fault_host);
    (void)hipSetDevice(prev);
    delete ctx;
}

// Grow-only device scratch owned by the context: *ptr_out stays valid until a later call asks for more bytes (the old
// buffer is then freed after a device synchronisation) or the context is destroyed.  256-byte aligned.
int bark_ctx_workspace(bark_ctx *ctx, size_t bytes, void **ptr_out) {
    error_buffer()[0] = 0;
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!ptr_out) return fail(BARK_ERR_ARG, "bark_ctx_workspace: null output");
    if (bytes > ctx->ws_bytes) {
        if (ctx->ws) {
            BARK_HIP_CHECK(hipDeviceSynchronize());
            BARK_HIP_CHECK(hipFree(ctx->ws));
            ctx->ws = nullptr;
            ctx->ws_bytes = 0;
        }
        hipError_t e = hipMalloc(&ctx->ws, bytes);
        if (e != hipSuccess) {
            ctx->ws = nullptr;
            return fail(BARK_ERR_WORKSPACE, "bark_ctx_workspace: cannot allocate %zu bytes: %s", bytes, hipGetErrorString(e));
        }
        ctx->ws_bytes = bytes;
    }
    *ptr_out = ctx->ws;
    return BARK_OK;
}

size_t bark_ctx_workspace_bytes(const bark_ctx *ctx) { return ctx ? ctx->ws_bytes : 0; }

// Reads and clears the categorical-fault flag (synchronises `stream`): *cat_fault_out != 0 means a leaf walk enqueued
// with this context since the last call evaluated a categorical split on a NaN / inf / negative feature value —
// where the reference raises inside `1 << int(x)` (forest.py:38).
int bark_ctx_status(bark_ctx *ctx, void *stream_, int32_t *cat_fault_out) {
    error_buffer()[0] = 0;
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!cat_fault_out) return fail(BARK_ERR_ARG, "bark_ctx_status: null output");
    hipStream_t s = static_cast<hipStream_t>(stream_);
    BARK_HIP_CHECK(hipMemcpyAsync(ctx->fault_host, ctx->fault, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    BARK_HIP_CHECK(hipMemsetAsync(ctx->fault, 0, sizeof(int32_t), s));
    BARK_HIP_CHECK(hipStreamSynchronize(s));
    *cat_fault_out = ctx->fault_host[0];
    return BARK_OK;
}

}  // extern "C"
