// Internal helpers shared by the libbarkhip.so translation units (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "../../include/bark_hip.h"
#include "../../include/bark_hip_testing.h"

namespace bark {

// thread-local last-error buffer behind bark_last_error()
char *error_buffer();
int fail(int code, const char *fmt, ...);

#define BARK_HIP_CHECK(expr)                                                                       \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return ::bark::fail(BARK_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                __FILE__, __LINE__);                                               \
    } while (0)

// Status of the launch just made: hipGetLastError(), or — test hook bark_debug_fail_launch(k), $BARK_TEST_HOOKS processes only — an injected failure of the
// k-th checked launch of the process (the kernel itself was enqueued; what is exercised is the error return path).
hipError_t launch_status();
#define BARK_LAUNCH_CHECK() BARK_HIP_CHECK(::bark::launch_status())

}  // namespace bark

// Per-device context (include/bark_hip.h: bark_ctx_create / bark_ctx_destroy).  Everything an entry point needs
// beyond its arguments lives here, so two host threads with a context (and a stream) each never share state.
struct bark_ctx {
    int device = 0;
    hipStream_t helper = nullptr;             // dense sweep: row launches beside the diag kernel
    hipStream_t helper2 = nullptr;            // dense sweep: look-ahead launches of the split-K bulk (even steps) / pipelined rows
    hipStream_t helper3 = nullptr;            // dense sweep: look-ahead bulk of the odd steps
    std::vector<hipEvent_t> events;           // fork / join events of the sweep (grown on demand, reused)
    hipEvent_t rejoin[3] = {nullptr, nullptr, nullptr};  // helper -> caller joins at the end of a chunk / on an error return
    std::vector<hipStream_t> chain_streams;   // multi-chain sampler step: one stream per chain (general shapes)
    std::vector<hipEvent_t> chain_done;
    hipEvent_t chain_fork = nullptr;
    void *ws = nullptr;                       // bark_ctx_workspace: grow-only device scratch
    size_t ws_bytes = 0;
    int32_t *fault = nullptr;                 // device: set by a leaf walk that met a NaN / inf / negative category
    int32_t *fault_host = nullptr;            // pinned mirror for bark_ctx_status
    char *stage_host = nullptr;               // pinned staging page of the host-pointer entry points (bark_ctx_upload, ..._host_pair)
    char *stage_dev = nullptr;                // its device twin
    hipEvent_t stage_event = nullptr;         // recorded behind the last staged asynchronous copy, on the stream it went to ...
    bool stage_busy = false;                  // ... and not yet waited for: the next staged call does, before it touches the page
};

namespace bark {

// ctx valid and made for the current device?  (entry points call this first)
int check_ctx(const bark_ctx *ctx);
int ctx_events(bark_ctx *ctx, size_t n);          // at least n events in ctx->events
int ctx_chain_streams(bark_ctx *ctx, size_t n);   // at least n chain streams + events
int set_lds_limits();                             // once per device: kernels with > 64 KiB of dynamic LDS (chol.hip)

constexpr size_t STAGE_BYTES = 64 * 1024;  // bark_ctx::stage_host / stage_dev
constexpr int NODE_BYTES = 26;          // forest.py:8-19, packed
constexpr uint32_t LEAF_FLAG = 0x80000000u;
constexpr uint32_t CAT_FLAG = 0x40000000u;
constexpr uint32_t FEAT_MASK = 0x3FFFFFFFu;

constexpr int TILE = 128;  // Cholesky block size == MFMA tile edge per workgroup
constexpr int MAX_LEAF_WORDS = 112;  // leaf-code dwords per point the Gram kernels can stage in LDS

inline int64_t round_up(int64_t x, int64_t q) { return (x + q - 1) / q * q; }

// Number of differing bytes of two dwords that each pack 4 dense leaf ids (one tree per byte): a tree pair
// agrees iff its byte of a ^ b is zero.  Ids < 128 keep bit 7 clear, so `x + 0x7f7f7f7f` cannot carry
// between bytes (SEVEN_BIT); the general form masks first.  forest.py:87 `np.equal(x1_leaves, x2_leaves)`.
template <bool SEVEN_BIT>
__device__ __forceinline__ uint32_t mismatched_bytes(uint32_t a, uint32_t b) {
    const uint32_t x = a ^ b;
    uint32_t z;
    if (SEVEN_BIT)
        z = (x + 0x7f7f7f7fu) & 0x80808080u;
    else
        z = (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;
    return __popc(z);
}

// leaf-code encodings (include/bark_hip.h): how the Gram kernels turn two code words into a count
enum LeafRep { REP_BYTES8 = 0, REP_BYTES7 = 1, REP_BITS = 2 };
inline LeafRep leaf_rep(const bark_pack_info *info) {
    if (bark_leaf_encoding(info) == BARK_LEAF_BITS) return REP_BITS;
    return info->max_leaves <= 128 ? REP_BYTES7 : REP_BYTES8;
}
// per code word: number of disagreeing trees (byte encodings) or of agreeing trees (one-hot bits)
template <int REP>
__device__ __forceinline__ uint32_t code_count(uint32_t a, uint32_t b) {
    if (REP == REP_BITS) return __popc(a & b);
    return mismatched_bytes<REP == REP_BYTES7>(a, b);
}
// trees that agree, from the accumulated per-word counts
template <int REP>
__device__ __forceinline__ int agree_count(uint32_t acc, int m) {
    return REP == REP_BITS ? (int)acc : m - (int)acc;
}

// Root-to-leaf walk of one point through one packed tree (wire format of pack.cpp) — forest.py:28-47 `_pass_one_through_tree`:
//   categorical:  (1 << int(x[f])) & int(threshold) != 0 -> left ;  otherwise:  x[f] <= float64(float32 threshold) -> left ;
//   NaN compares false -> right.  Bounded by the packer's max_depth.  Shared by traverse.hip and the one-launch kernel of N <= 128.
template <bool X_IN_LDS>
__device__ __forceinline__ uint4 walk_tree(const uint4 *__restrict__ tree, int max_depth, const double *xrow,
                                           int32_t *__restrict__ fault) {
    uint4 n = tree[0];
    for (int step = 0; step < max_depth && !(n.x & LEAF_FLAG); ++step) {
        const uint32_t f = n.x & FEAT_MASK;
        const double xv = xrow[f];
        bool left;
        if (n.x & CAT_FLAG) {
            const double xt = trunc(xv);  // int(): toward zero
            // `1 << int(x)` raises in the reference for NaN / inf / x <= -1 (forest.py:38): flag it, the host raises
            if (!(xt >= 0.0 && xt < INFINITY)) *fault = 1;
            left = (xt >= 0.0 && xt < 32.0) ? ((n.y >> (uint32_t)xt) & 1u) : false;
        } else {
            left = xv <= (double)__uint_as_float(n.y);
        }
        n = tree[left ? n.z : n.w];
    }
    return n;
}

}  // namespace bark
