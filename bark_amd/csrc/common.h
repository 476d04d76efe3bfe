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

}  // namespace bark
