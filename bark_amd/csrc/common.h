// Internal helpers shared by the libbarkhip.so translation units (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/bark_hip.h"

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

#define BARK_LAUNCH_CHECK() BARK_HIP_CHECK(hipGetLastError())

constexpr int NODE_BYTES = 26;          // forest.py:8-19, packed
constexpr uint32_t LEAF_FLAG = 0x80000000u;
constexpr uint32_t CAT_FLAG = 0x40000000u;
constexpr uint32_t FEAT_MASK = 0x3FFFFFFFu;

constexpr int TILE = 128;  // Cholesky block size == MFMA tile edge per workgroup

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

}  // namespace bark
