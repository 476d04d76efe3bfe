// Leaf traversal on gfx950 — src/bark/forest.py:28-67 (_pass_one_through_tree,
// pass_through_tree, pass_through_forest).
//
// One thread owns one data point and walks every tree of one forest; a workgroup is 256
// points of one forest sample.  The point's feature row sits in LDS (odd row stride, so the
// data-dependent `x[feature]` reads spread over banks); the handful of live nodes per tree
// come from the packed wire format (pack.cpp) through L1/L2 — every lane starts at the same
// root, so those loads are mostly broadcasts.  The walk is bounded by the packer's
// `max_depth`, so every wave terminates whatever the node bytes contain.
//
// Integer output, must be bit-identical to the reference:
//   categorical:  (1 << int(x[f])) & int(threshold)  != 0  -> left       (forest.py:37-39)
//   otherwise  :  x[f] <= float64(float32 threshold)       -> left       (forest.py:41)
//   NaN compares false -> right.
#include "common.h"

namespace bark {
namespace {

constexpr int WALK_THREADS = 256;
constexpr int IDX_CHUNK = 32;  // leaf_walk_kernel MODE 0: trees per staged chunk (one 128-byte segment per point)

// MODE 0: out[b][i][t] = original node index (uint32)   — the reference's (N, m) layout per forest
// MODE 1: out[b][w][i] = 4 dense leaf ids packed per dword — Gram kernel input, Npad points per plane
// MODE 2: out[b][w][i] = one-hot code: bit (leaf.z) set for the leaf reached in every tree; W = words.
//         Bit positions grow with the tree index, so a thread keeps one accumulator word and flushes
//         it whenever the next tree's bit lands in a later word.
template <int MODE, bool X_IN_LDS>
__global__ __launch_bounds__(WALK_THREADS) void leaf_walk_kernel(const uint4 *__restrict__ nodes, int stride, int m,
                                                                 int max_depth, const double *__restrict__ X, int N,
                                                                 int d, int npad, int words,
                                                                 uint32_t *__restrict__ out,
                                                                 int32_t *__restrict__ fault) {
    extern __shared__ __attribute__((aligned(16))) double xs[];
    const int tid = threadIdx.x;
    const int i = blockIdx.x * WALK_THREADS + tid;
    const int b = blockIdx.y;
    const int sd = d | 1;
    const double *xrow;
    if (X_IN_LDS) {
        // coalesced copy of up to 256 rows, then each thread reads its own (padded) row
        const int row0 = blockIdx.x * WALK_THREADS;
        const int rows = min(WALK_THREADS, N - row0);
        for (int e = tid; e < rows * d; e += WALK_THREADS) {
            const int r = e / d, c = e - r * d;
            xs[r * sd + c] = X[(size_t)row0 * d + e];
        }
        __syncthreads();
        xrow = xs + tid * sd;
    } else {
        xrow = X + (size_t)min(i, N - 1) * d;
    }
    const bool live = i < N;
    const uint4 *forest = nodes + (size_t)b * m * stride;

    if (MODE == 0) {
        // the reference's (N, m) layout: a thread's m results are one row, so threads would store m dwords apart.  The
        // results go through LDS in chunks of IDX_CHUNK trees instead and leave as 128-byte row segments, 32 lanes a row
        // (X_IN_LDS only: the staging tile sits behind the point rows; the fallback for very wide X stores directly)
        if (X_IN_LDS) {
            uint32_t *st = reinterpret_cast<uint32_t *>(xs + WALK_THREADS * sd);  // [256][IDX_CHUNK + 1]
            const int row0 = blockIdx.x * WALK_THREADS, rows = min(WALK_THREADS, N - row0);
            uint32_t *ob = out + ((size_t)b * N + row0) * m;
            for (int t0 = 0; t0 < m; t0 += IDX_CHUNK) {
                const int nt = min(IDX_CHUNK, m - t0);
                if (live)
                    for (int k = 0; k < nt; ++k)
                        st[tid * (IDX_CHUNK + 1) + k] = walk_tree<true>(forest + (size_t)(t0 + k) * stride, max_depth, xrow, fault).y;
                __syncthreads();
                for (int e = tid; e < rows * IDX_CHUNK; e += WALK_THREADS) {
                    const int r = e / IDX_CHUNK, k = e - r * IDX_CHUNK;
                    if (k < nt) ob[(size_t)r * m + t0 + k] = st[r * (IDX_CHUNK + 1) + k];
                }
                __syncthreads();
            }
            return;
        }
        if (!live) return;
        uint32_t *o = out + ((size_t)b * N + i) * m;
        for (int t = 0; t < m; ++t) o[t] = walk_tree<X_IN_LDS>(forest + (size_t)t * stride, max_depth, xrow, fault).y;
    } else if (MODE == 2) {
        if (i >= npad) return;
        uint32_t *o = out + (size_t)b * words * npad + i;
        int cur = 0;
        uint32_t accw = 0;
        if (live) {
            for (int t = 0; t < m; ++t) {
                const uint32_t bit = walk_tree<X_IN_LDS>(forest + (size_t)t * stride, max_depth, xrow, fault).z;
                const int w = (int)(bit >> 5);
                while (cur < w && cur < words) {  // monotone: earlier words are complete
                    o[(size_t)cur * npad] = accw;
                    accw = 0;
                    ++cur;
                }
                accw |= 1u << (bit & 31u);
            }
        }
        for (; cur < words; ++cur) {
            o[(size_t)cur * npad] = accw;
            accw = 0;
        }
    } else {
        const int W = (m + 3) >> 2;
        if (i >= npad) return;
        for (int w = 0; w < W; ++w) {
            uint32_t word = 0;
            if (live) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int t = w * 4 + q;
                    if (t < m) {
                        const uint4 leaf = walk_tree<X_IN_LDS>(forest + (size_t)t * stride, max_depth, xrow, fault);
                        word |= (leaf.x & 0xFFu) << (8 * q);
                    }
                }
            }
            out[((size_t)b * W + w) * npad + i] = word;
        }
    }
}

// Small grids (a lone matrix, c2, a few hundred small matrices: fewer than 8 workgroups per CU): the walk is a chain of m
// dependent tree walks per thread and too little runs beside it, so the trees of a point are shared out over WALK_GROUPS threads
// — 32 points per workgroup, 8 x the workgroups, an eighth of the chain each (N = 1024, one forest: 29 -> 8 us).  Same walks, same
// integer results.  MODE 2: the groups OR their bits into the point's words in LDS; MODE 1: a word (4 trees) belongs to
// one group.
constexpr int WALK_GROUPS = 8, WALK_POINTS = WALK_THREADS / WALK_GROUPS;
// Which grids: every one that would leave the one-thread-per-point kernel fewer than 8 workgroups per CU.  That kernel is a
// chain of m dependent walks (each two or three dependent 16-byte loads) per thread, and with one wave per SIMD nothing hides
// them: 256 forests x 256 points — exactly one workgroup per CU, so the old rule (fewer workgroups than CUs) passed it by —
// took 47 us of a 187 us evaluation (profiles/r05/small_n.txt).  Round 5, same box, one process per variant, device ms of the
// whole MLL call, threshold 256 | 1024 | 4096 | none: N = 64 x 256 forests 0.053 | 0.036 | 0.035 | 0.036, N = 128 x 256 0.077 | 0.059 |
// 0.059 | 0.059, N = 256 x 256 0.187 | 0.157 | 0.158 | 0.156, N = 512 x 256 0.554 | 0.522 | 0.515 | 0.530, N = 1024 x 64 1.050 | 1.003 |
// 1.014 | 1.005, N = 4096 x 16 7.006 | 6.938 | 6.933 | 6.966; N = 64 x 2048 (a grid of 2048) 0.155 | 0.155 | 0.165 | 0.166 — with 8 workgroups
// per CU the plain kernel hides its chains itself.  Identical integer results (the MLL digests of the A/B agree).
#ifndef BARK_WALK_NODES_LDS
#define BARK_WALK_NODES_LDS 1
#endif
#ifndef BARK_WALK_GROUPED_MAX_WGS
#define BARK_WALK_GROUPED_MAX_WGS 2048
#endif
constexpr int64_t WALK_GROUPED_MAX_WGS = BARK_WALK_GROUPED_MAX_WGS;  // grids of the one-thread-per-point kernel below this take the grouped one
// NLDS: the forest's packed nodes (m x stride x 16 bytes: ~9 KB for 50 prior trees) are copied into LDS first and walked
// there — a thread's six or seven trees are a chain of ~3 dependent node reads each, ~0.5 us apiece from L1 / L2, ~100 cycles
// from LDS (round 5: the walk was 10 of the 28 us of N = 64 x 256 forests).  Forests too bushy for the LDS left are walked in
// global memory as before.
template <int MODE, bool NLDS>
__global__ __launch_bounds__(WALK_THREADS) void leaf_walk_grouped_kernel(const uint4 *__restrict__ nodes, int stride, int m,
                                                                         int max_depth, const double *__restrict__ X, int N,
                                                                         int d, int npad, int words,
                                                                         uint32_t *__restrict__ out,
                                                                         int32_t *__restrict__ fault) {
    static_assert(MODE == 1 || MODE == 2, "leaf codes only");
    extern __shared__ __attribute__((aligned(16))) double xs[];
    const int tid = threadIdx.x, pl = tid % WALK_POINTS, g = tid / WALK_POINTS;
    const int row0 = blockIdx.x * WALK_POINTS, i = row0 + pl, b = blockIdx.y;
    const int sd = d | 1;
    uint32_t *acc = reinterpret_cast<uint32_t *>(xs + WALK_POINTS * sd);  // MODE 2: [WALK_POINTS][words]
    const int rows = min(WALK_POINTS, N - row0);
    for (int e = tid; e < rows * d; e += WALK_THREADS) {
        const int r = e / d, c = e - r * d;
        xs[r * sd + c] = X[(size_t)row0 * d + e];
    }
    if (MODE == 2)
        for (int e = tid; e < WALK_POINTS * words; e += WALK_THREADS) acc[e] = 0;
    const uint4 *forest = nodes + (size_t)b * m * stride;
    if (NLDS) {  // (16-byte aligned: the X rows and the accumulator words before it are padded to 16 bytes by the launcher)
        uint4 *ln = reinterpret_cast<uint4 *>(acc + (((MODE == 2 ? WALK_POINTS * words : 0) + 3) & ~3));
        for (int e = tid; e < m * stride; e += WALK_THREADS) ln[e] = forest[e];
        forest = ln;
    }
    __syncthreads();
    const double *xrow = xs + pl * sd;
    const bool live = i < N;
    if (MODE == 2) {
        if (live) {
            const int t0 = (int)(((long)m * g) / WALK_GROUPS), t1 = (int)(((long)m * (g + 1)) / WALK_GROUPS);
            for (int t = t0; t < t1; ++t) {
                const uint32_t bit = walk_tree<true>(forest + (size_t)t * stride, max_depth, xrow, fault).z;
                const int w = (int)(bit >> 5);
                if (w < words) atomicOr(&acc[pl * words + w], 1u << (bit & 31u));
            }
        }
        __syncthreads();
        uint32_t *o = out + (size_t)b * words * npad;
        for (int e = tid; e < WALK_POINTS * words; e += WALK_THREADS) {
            const int w = e / WALK_POINTS, q = e - w * WALK_POINTS;  // consecutive threads: consecutive points of a plane
            if (row0 + q < npad) o[(size_t)w * npad + row0 + q] = acc[q * words + w];
        }
    } else {
        const int W = (m + 3) >> 2;
        if (i >= npad) return;
        const int w0 = (int)(((long)W * g) / WALK_GROUPS), w1 = (int)(((long)W * (g + 1)) / WALK_GROUPS);
        for (int w = w0; w < w1; ++w) {
            uint32_t word = 0;
            if (live) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int t = w * 4 + q;
                    if (t < m) {
                        const uint4 leaf = walk_tree<true>(forest + (size_t)t * stride, max_depth, xrow, fault);
                        word |= (leaf.x & 0xFFu) << (8 * q);
                    }
                }
            }
            out[((size_t)b * W + w) * npad + i] = word;
        }
    }
}

template <int MODE>
int launch_walk(const void *packed, const bark_pack_info *info, const double *X, int64_t N, int64_t d, uint32_t *out,
                int32_t *fault, void *stream, int force_words = 0) {
    if (!packed || !info || !X || !out || !fault) return fail(BARK_ERR_ARG, "leaf walk: null argument");
    if (N < 1 || d < 1 || N > (1 << 30)) return fail(BARK_ERR_ARG, "leaf walk: bad N=%lld d=%lld", (long long)N, (long long)d);
    if (info->B > 65535) return fail(BARK_ERR_ARG, "leaf walk: at most 65535 forests per call (got %lld)", (long long)info->B);
    const int64_t npad = bark_leaf_npad(N);
    const int words = force_words ? force_words : (int)bark_leaf_words(info);
    if (MODE != 0 && !force_words && words > MAX_LEAF_WORDS)
        return fail(BARK_ERR_ARG, "forest needs %d leaf-code words per point (max %d): too many leaves in total", words,
                    MAX_LEAF_WORDS);
    const int64_t extent = MODE != 0 ? npad : N;
    dim3 grid((unsigned)((extent + WALK_THREADS - 1) / WALK_THREADS), (unsigned)info->B);
    const size_t lds = (size_t)WALK_THREADS * (d | 1) * sizeof(double) +
                       (MODE == 0 ? (size_t)WALK_THREADS * (IDX_CHUNK + 1) * sizeof(uint32_t) : 0);  // + MODE 0's staging tile
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint4 *nodes = static_cast<const uint4 *>(packed);
    if constexpr (MODE != 0) {
        // X rows (32 x (d | 1) doubles: a multiple of 16 bytes as 32 doubles are), the accumulator words rounded up to 16 bytes, the nodes
        const size_t gacc = MODE == 2 ? (((size_t)WALK_POINTS * words + 3) & ~(size_t)3) * sizeof(uint32_t) : 0;
        const size_t glds = (size_t)WALK_POINTS * (d | 1) * sizeof(double) + gacc;
        const size_t gnodes = (size_t)info->m * info->stride * sizeof(uint4);
        if ((int64_t)grid.x * grid.y < WALK_GROUPED_MAX_WGS && glds <= 64 * 1024) {  // too few workgroups to hide the chains: share a point's trees out
            const dim3 gg((unsigned)((npad + WALK_POINTS - 1) / WALK_POINTS), (unsigned)info->B);
            if (BARK_WALK_NODES_LDS && glds + gnodes <= 32 * 1024)
                hipLaunchKernelGGL((leaf_walk_grouped_kernel<MODE, true>), gg, dim3(WALK_THREADS), glds + gnodes, s, nodes, (int)info->stride,
                                   (int)info->m, (int)info->max_depth, X, (int)N, (int)d, (int)npad, words, out, fault);
            else
                hipLaunchKernelGGL((leaf_walk_grouped_kernel<MODE, false>), gg, dim3(WALK_THREADS), glds, s, nodes, (int)info->stride,
                                   (int)info->m, (int)info->max_depth, X, (int)N, (int)d, (int)npad, words, out, fault);
            BARK_LAUNCH_CHECK();
            return BARK_OK;
        }
    }
    if (lds <= 64 * 1024) {
        hipLaunchKernelGGL((leaf_walk_kernel<MODE, true>), grid, dim3(WALK_THREADS), lds, s, nodes, (int)info->stride,
                           (int)info->m, (int)info->max_depth, X, (int)N, (int)d, (int)npad, words, out, fault);
    } else {
        hipLaunchKernelGGL((leaf_walk_kernel<MODE, false>), grid, dim3(WALK_THREADS), 0, s, nodes, (int)info->stride,
                           (int)info->m, (int)info->max_depth, X, (int)N, (int)d, (int)npad, words, out, fault);
    }
    BARK_LAUNCH_CHECK();
    return BARK_OK;
}

}  // namespace

// one-hot leaf code with `words` = ceil(max_bits / 32) planes, whatever encoding the Gram kernels would pick
int walk_one_hot(const void *packed, const bark_pack_info *info, const double *X, int64_t N, int64_t d, int words,
                 uint32_t *out, int32_t *fault, hipStream_t stream) {
    return launch_walk<2>(packed, info, X, N, d, out, fault, stream, words);
}

// Gram-kernel leaf codes (encoding chosen from `info`), for the sweep entry points that already hold a context
int walk_codes(const void *packed, const bark_pack_info *info, const double *X, int64_t N, int64_t d, uint32_t *out,
               int32_t *fault, hipStream_t stream) {
    if (!info) return fail(BARK_ERR_ARG, "leaf codes: null info");
    return bark_leaf_encoding(info) == BARK_LEAF_BITS ? launch_walk<2>(packed, info, X, N, d, out, fault, stream)
                                                      : launch_walk<1>(packed, info, X, N, d, out, fault, stream);
}

}  // namespace bark

using namespace bark;

extern "C" {

int64_t bark_leaf_npad(int64_t N) { return round_up(N, TILE); }

int bark_leaf_indices_hip(bark_ctx *ctx, const void *packed, const bark_pack_info *info, const double *X, int64_t N,
                          int64_t d, uint32_t *out, void *stream) {
    error_buffer()[0] = 0;
    int rc = check_ctx(ctx);
    if (rc) return rc;
    return launch_walk<0>(packed, info, X, N, d, out, ctx->fault, stream);
}

// One-hot bits cost 2 VALU per 32 bits in the Gram kernels, packed bytes 4 (6 with ids >= 128) per 4 trees:
// take the bitset when it is the cheaper compare, and always when a tree has more leaves than a byte holds.
int bark_leaf_encoding(const bark_pack_info *info) {
    if (!info) return BARK_LEAF_BYTES;
    const int64_t wbits = (info->max_bits + 31) / 32, wbytes = (info->m + 3) / 4;
    return (info->max_leaves > 256 || wbits < 2 * wbytes) ? BARK_LEAF_BITS : BARK_LEAF_BYTES;
}

int64_t bark_leaf_words(const bark_pack_info *info) {
    if (!info) return 0;
    return bark_leaf_encoding(info) == BARK_LEAF_BITS ? (info->max_bits + 31) / 32 : (info->m + 3) / 4;
}

int bark_leaf_codes_hip(bark_ctx *ctx, const void *packed, const bark_pack_info *info, const double *X, int64_t N,
                        int64_t d, uint32_t *out, void *stream) {
    error_buffer()[0] = 0;
    int rc = check_ctx(ctx);
    if (rc) return rc;
    return walk_codes(packed, info, X, N, d, out, ctx->fault, static_cast<hipStream_t>(stream));
}

// out[i * ldo + c] = value if leaves[i * ldl] == ids[c] else 0 — the one-hot leaf vectors of forest.py:70-75
// (`np.equal(leaves[:, None], all_leaves[None, :])`), optionally scaled (bark_sampler.py:233-236 `* s_sqrtm`).
__global__ void onehot_match_kernel(const uint32_t *__restrict__ leaves, int64_t N, int64_t ldl,
                                    const uint32_t *__restrict__ ids, int r, double value, double *__restrict__ out,
                                    int64_t ldo) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * r) return;
    const int64_t i = e / r;
    const int c = (int)(e - i * r);
    out[i * ldo + c] = leaves[i * ldl] == ids[c] ? value : 0.0;
}

int bark_onehot_match_hip(const uint32_t *leaves, int64_t N, int64_t ldl, const uint32_t *ids, int64_t r, double value,
                          double *out, int64_t ldo, void *stream) {
    error_buffer()[0] = 0;
    if (!leaves || !ids || !out || N < 1 || r < 1 || ldl < 1 || ldo < r || N * r > ((int64_t)1 << 40))
        return fail(BARK_ERR_ARG, "bark_onehot_match_hip: bad argument");
    hipLaunchKernelGGL(onehot_match_kernel, dim3((unsigned)((N * r + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), leaves, N, ldl, ids, (int)r, value, out, ldo);
    BARK_LAUNCH_CHECK();
    return BARK_OK;
}

}  // extern "C"
