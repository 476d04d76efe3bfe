// bark_ctx: the per-device context of the C ABI (include/bark_hip.h).
//
// The reference's callers are single-threaded Python; the library is re-entrant per context: helper
// streams, events, the scratch buffer and the categorical-fault flag belong to a bark_ctx, created and destroyed
// explicitly.  The only process-wide state left is the thread-local last-error buffer and the once-per-device
// dynamic-LDS attributes of the kernels (std::call_once).
#include <atomic>
#include <cstdlib>
#include <cstring>

#include "common.h"

namespace bark {

namespace {
// Launch-failure injection (include/bark_hip_testing.h) exists only in processes started with $BARK_TEST_HOOKS set: read once,
// when the library is loaded.  Without it launch_status() is hipGetLastError() plus one test of a constant.
const bool g_test_hooks = [] {
    const char *v = std::getenv("BARK_TEST_HOOKS");
    return v && v[0] && !(v[0] == '0' && v[1] == 0);
}();
std::atomic<long> g_fail_countdown{0};
}

hipError_t launch_status() {
    const hipError_t e = hipGetLastError();
    if (g_test_hooks && g_fail_countdown.load(std::memory_order_relaxed) > 0 && g_fail_countdown.fetch_sub(1) == 1)
        return hipErrorLaunchFailure;
    return e;
}

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
    for (hipEvent_t &e : ctx->rejoin)
        if (!e) BARK_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
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

// Test hook (include/bark_hip_testing.h): the k-th launch from now on whose status the library checks reports
// hipErrorLaunchFailure (k <= 0: off).  Process-wide; returns the previous countdown — or -1, and does nothing, in a process
// that was not started with $BARK_TEST_HOOKS set.
long bark_debug_fail_launch(long k) {
    if (!g_test_hooks) return -1;
    return g_fail_countdown.exchange(k > 0 ? k : 0);
}

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
    for (hipEvent_t e : ctx->rejoin)
        if (e) (void)hipEventDestroy(e);
    for (hipStream_t s : ctx->chain_streams) {
        (void)hipStreamSynchronize(s);
        (void)hipStreamDestroy(s);
    }
    for (hipEvent_t e : ctx->chain_done) (void)hipEventDestroy(e);
    if (ctx->chain_fork) (void)hipEventDestroy(ctx->chain_fork);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->fault) (void)hipFree(ctx->fault);
    if (ctx->fault_host) (void)hipHostFree(ctx->fault_host);
    if (ctx->stage_host) (void)hipHostFree(ctx->stage_host);
    if (ctx->stage_dev) (void)hipFree(ctx->stage_dev);
    if (ctx->stage_event) (void)hipEventDestroy(ctx->stage_event);
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

// ---------------------------------------------------------------------------------------------------------------
// Host-pointer entry points: the ABI without a tensor library on the caller's side.  The reference's sampler
// (bark_sampler.py:120 `_run_bark_sampler_multichain`, :216 `_step_bark_sampler`) is numba nopython code: it can call C
// functions through ctypes function pointers with integer / float arguments and numpy arrays' addresses, but no Python
// object — so no torch tensor can carry its device memory.  These functions give it device buffers, staged copies and the
// per-tree proposal (bark_sampler.py:233-257) straight from the two host trees; everything else it needs
// (bark_forest_pack*, bark_mll_batched_hip, bark_lowrank_swap_apply_hip, bark_lowrank_status_hip, bark_ctx_status) already
// takes plain pointers.  INTEGRATION.md §4 shows the nopython caller.
// ---------------------------------------------------------------------------------------------------------------
// The staging page (one pinned host page + its device twin per context) is shared by every staged entry point, whatever
// stream the caller passes: a staged copy records ctx->stage_event behind itself on its stream (stage_release), and the next
// staged call waits for THAT event before it touches the page (stage_acquire) — not for its own stream, which may be another
// one (upload on stream A, then upload or a host-pair proposal on stream B used to overwrite the page under A's pending copy).
static int stage_acquire(bark_ctx *ctx) {
    if (!ctx->stage_host) BARK_HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&ctx->stage_host), STAGE_BYTES, hipHostMallocDefault));
    if (!ctx->stage_dev) BARK_HIP_CHECK(hipMalloc(reinterpret_cast<void **>(&ctx->stage_dev), STAGE_BYTES));
    if (!ctx->stage_event) BARK_HIP_CHECK(hipEventCreateWithFlags(&ctx->stage_event, hipEventDisableTiming));
    if (ctx->stage_busy) {
        BARK_HIP_CHECK(hipEventSynchronize(ctx->stage_event));
        ctx->stage_busy = false;
    }
    return BARK_OK;
}
static int stage_release(bark_ctx *ctx, hipStream_t s) {  // the page is in use by everything enqueued on s so far
    BARK_HIP_CHECK(hipEventRecord(ctx->stage_event, s));
    ctx->stage_busy = true;
    return BARK_OK;
}

int bark_dev_alloc(bark_ctx *ctx, size_t bytes, void **ptr_out) {
    error_buffer()[0] = 0;
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!ptr_out || bytes == 0) return fail(BARK_ERR_ARG, "bark_dev_alloc: null output or zero bytes");
    *ptr_out = nullptr;
    const hipError_t e = hipMalloc(ptr_out, bytes);
    if (e != hipSuccess) {
        *ptr_out = nullptr;
        return fail(BARK_ERR_WORKSPACE, "bark_dev_alloc: cannot allocate %zu bytes: %s", bytes, hipGetErrorString(e));
    }
    return BARK_OK;
}

int bark_dev_free(bark_ctx *ctx, void *ptr) {
    error_buffer()[0] = 0;
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (ptr) BARK_HIP_CHECK(hipFree(ptr));  // synchronises the device, as hipFree does
    return BARK_OK;
}

// Host -> device.  Up to 64 KiB go through the context's pinned page (one memcpy + one asynchronous copy); larger blocks
// are copied from the caller's pageable memory.  Either way `src_host` may be reused when the call returns.
int bark_ctx_upload(bark_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes, void *stream_) {
    error_buffer()[0] = 0;
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!dst_dev || !src_host) return fail(BARK_ERR_ARG, "bark_ctx_upload: null pointer");
    if (bytes == 0) return BARK_OK;
    hipStream_t s = static_cast<hipStream_t>(stream_);
    if (bytes <= STAGE_BYTES) {
        if ((rc = stage_acquire(ctx))) return rc;  // an earlier staged copy — on whatever stream — has left the page
        std::memcpy(ctx->stage_host, src_host, bytes);
        BARK_HIP_CHECK(hipMemcpyAsync(dst_dev, ctx->stage_host, bytes, hipMemcpyHostToDevice, s));
        return stage_release(ctx, s);
    }
    BARK_HIP_CHECK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, s));
    BARK_HIP_CHECK(hipStreamSynchronize(s));
    return BARK_OK;
}

// Device -> host; synchronises `stream`: dst_host holds the data when the call returns.
int bark_ctx_download(bark_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes, void *stream_) {
    error_buffer()[0] = 0;
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!dst_host || !src_dev) return fail(BARK_ERR_ARG, "bark_ctx_download: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream_);
    if (bytes > 0 && bytes <= STAGE_BYTES) {
        if ((rc = stage_acquire(ctx))) return rc;  // (an earlier staged upload has left the page)
        BARK_HIP_CHECK(hipMemcpyAsync(ctx->stage_host, src_dev, bytes, hipMemcpyDeviceToHost, s));
        BARK_HIP_CHECK(hipStreamSynchronize(s));
        std::memcpy(dst_host, ctx->stage_host, bytes);
        return BARK_OK;
    }
    if (bytes > 0) BARK_HIP_CHECK(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, s));
    BARK_HIP_CHECK(hipStreamSynchronize(s));
    return BARK_OK;
}

int bark_stream_sync(bark_ctx *ctx, void *stream_) {
    error_buffer()[0] = 0;
    int rc = check_ctx(ctx);
    if (rc) return rc;
    BARK_HIP_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(stream_)));
    return BARK_OK;
}

// bark_sampler.py:233-257 for ONE tree, from the two trees as the sampler holds them: pair26 = [old tree, new tree], two
// runs of L packed 26-byte records in HOST memory (forest[tree_idx] and new_nodes).  Packs the pair (host), stages it
// through the pinned page, evaluates the swap (bark_tree_swap_eval_hip) and returns
//   scalars_host_out[0] = y'K^-1 y - y'K'^-1 y,  scalars_host_out[1] = log|K'| - log|K|,  *r_out = leaves of the pair
// (the rank bark_lowrank_swap_apply_hip(K_inv, N, *r_out, workspace, K_inv, stream) needs to commit an accepted proposal).
// Synchronises `stream`.  More than 64 leaves in the pair: BARK_ERR_ARG (the reference's own subtract-then-add chain through
// bark_lowrank_update_hip covers that case).
int bark_tree_swap_eval_host_pair(bark_ctx *ctx, const double *K_inv, int64_t N, const void *pair26, int64_t L,
                                  const int64_t *feat_types, int64_t d, const double *X, double s, const double *y,
                                  double *scalars_host_out, int64_t *r_out, void *workspace, size_t workspace_bytes,
                                  void *stream_) {
    error_buffer()[0] = 0;
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!pair26 || !feat_types || !scalars_host_out || !r_out || L < 1)
        return fail(BARK_ERR_ARG, "bark_tree_swap_eval_host_pair: bad argument");
    bark_pack_info info_old, info;
    if ((rc = bark_forest_pack_info(pair26, 1, 1, L, feat_types, d, &info_old))) return rc;  // the old tree alone: r_old
    if ((rc = bark_forest_pack_info(pair26, 1, 2, L, feat_types, d, &info))) return rc;
    constexpr size_t SCALARS_AT = STAGE_BYTES - 64;
    if ((size_t)info.packed_bytes > SCALARS_AT)
        return fail(BARK_ERR_ARG, "bark_tree_swap_eval_host_pair: packed pair of %lld bytes exceeds the staging page", (long long)info.packed_bytes);
    if ((rc = stage_acquire(ctx))) return rc;  // the page is free (an earlier staged copy, on whatever stream, has completed)
    hipStream_t st = static_cast<hipStream_t>(stream_);
    if ((rc = bark_forest_pack(pair26, feat_types, d, &info, ctx->stage_host))) return rc;
    BARK_HIP_CHECK(hipMemcpyAsync(ctx->stage_dev, ctx->stage_host, (size_t)info.packed_bytes, hipMemcpyHostToDevice, st));
    double *scalars_dev = reinterpret_cast<double *>(ctx->stage_dev + SCALARS_AT);
    rc = bark_tree_swap_eval_hip(ctx, K_inv, N, ctx->stage_dev, &info, X, d, info_old.max_bits, s, y, scalars_dev, workspace,
                                 workspace_bytes, stream_);
    if (rc) {  // kernels that read the device page may be enqueued already: the next staged call waits for them
        (void)stage_release(ctx, st);
        return rc;
    }
    BARK_HIP_CHECK(hipMemcpyAsync(ctx->stage_host + SCALARS_AT, scalars_dev, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
    BARK_HIP_CHECK(hipStreamSynchronize(st));
    std::memcpy(scalars_host_out, ctx->stage_host + SCALARS_AT, 2 * sizeof(double));
    *r_out = info.max_bits;
    return BARK_OK;
}

}  // extern "C"
