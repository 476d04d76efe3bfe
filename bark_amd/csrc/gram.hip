// Leaf-coincidence Gram matrix on gfx950 — src/bark/forest.py:78-98
// (forest_gram_matrix / batched_forest_gram_matrix).
//
//   K[i][j] = (1.0/m) * #{t : leaf_t(x1_i) == leaf_t(x2_j)}
//
// The reference materialises an N x M x m boolean tensor (forest.py:87) and sums it; here a
// 64x64 output tile is produced per workgroup from two strips of byte-packed leaf ids held in
// LDS.  Two encodings of a point's leaves (include/bark_hip.h, bark_leaf_encoding):
//   one-hot bits : tree t owns L_t bits, #agreeing trees = popcount(z_i & z_j): v_and + v_bcnt per 32 bits
//                  (a forest with ~3 leaves per tree needs 5 dwords for 50 trees);
//   packed bytes : 4 trees per dword, a pair agrees iff its byte of `a ^ b` is zero: xor + add + and + v_bcnt
//                  per 4 trees (dense ids < 128 keep bit 7 clear, so `x + 0x7f7f7f7f` is carry-free).
// With the byte code the kernel is VALU-bound (compute-only build 0.52 ms vs store-only 0.42 ms for
// 16 x 4096^2); with the bit code it is bound by the 8*N*M bytes of fp64 output it streams to HBM: each
// wave store instruction writes two full 512-byte row segments (32 lanes x 16 B).
//
// Bit-exactness: the reference computes `1 / m * count` => fl(fl(1/m) * count); optional
// `scale *` and `+ (1e-6 + noise)` on the diagonal follow in the reference's order
// (tree_gps.py:97-100).  The library is built with -ffp-contract=off so no FMA fuses them.
#include <type_traits>

#include "common.h"

namespace bark {
namespace {

// Output tile of a workgroup: GTR rows x GTC columns, 4096 entries (8 rows x 2 columns per thread).  The kernel is bound
// by its fp64 stores.  Measured at 16 x 4096^2 (tools/time_gram.py, same box; torch's fill kernel writes 6.3-6.6 TB/s):
// 64 x 64 tile 6.0 TB/s, 32 x 128 5.9 (6.35 with non-temporal stores), 16 x 256 5.6 (5.8 nt), 8 x 512 5.1 (5.7 nt) —
// longer row segments per workgroup do NOT help; streaming (nt) stores do once a wave writes full 1 KiB runs.
#ifndef BARK_GRAM_TC
#define BARK_GRAM_TC 128
#endif
constexpr int GRAM_TC_WIDE = BARK_GRAM_TC;  // build-time tuning constant: 64, 128, 256 or 512
constexpr int GRAM_TC_NARROW = 64;          // forests whose leaf codes are too wide for the wide tile's LDS strips
constexpr int GRAM_THREADS = 256;
static_assert(GRAM_TC_WIDE == 64 || GRAM_TC_WIDE == 128 || GRAM_TC_WIDE == 256 || GRAM_TC_WIDE == 512, "tile width");

struct GramArgs {
    const uint32_t *leaf1;  // (B, W, npad1)
    const uint32_t *leaf2;  // (B, W, npad2)
    int npad1, npad2, W, m;
    int N, M;        // real extents (rows from leaf1, cols from leaf2)
    int Nout, Mout;  // fill extents (>= N, M): beyond the real block write identity (sym) or 0
    const double *shift;
    const double *scale;
    const double *noise;
    double *out;
    long long ld, batch_stride;
    int pad_identity;  // 1: out[i][i] = 1 in the padding (symmetric fill for the Cholesky)
    int upper_only;    // 1: skip 64-tiles strictly below the 128-block diagonal
};

// VEC2: every row of the output starts on a 16-byte boundary (even ld and batch stride, aligned base): a thread's column pair is one
// 16-byte store.  Otherwise (an odd N x M output: every other row starts 8 bytes off) the rows that are off by 8 bytes are written
// as SHIFTED pairs — (own second value, the next lane's first) at column j0 + 1, which IS 16-byte aligned there; the first
// lane of a row run adds its first value, the last one its second, as 8-byte stores — instead of 8-byte stores throughout
// (3.4 TB/s at N = 4097 against 6.0 at N = 4096).
template <int REP, bool VEC2, int GTC>
__global__ __launch_bounds__(GRAM_THREADS) void gram_kernel(GramArgs p) {
    constexpr int GTR = 8 * (GRAM_THREADS / (GTC / 2));
    extern __shared__ __attribute__((aligned(16))) uint32_t strips[];  // rows[W][GTR] | cols[W][GTC]
    const int tid = threadIdx.x;
    const int b = blockIdx.z;
    const int row0 = blockIdx.y * GTR, col0 = blockIdx.x * GTC;
    if (p.upper_only && ((col0 + GTC - 1) >> 7) < (row0 >> 7)) return;  // wholly below the 128-block diagonal

    uint32_t *rows = strips;
    uint32_t *cols = strips + p.W * GTR;
    for (int e = tid; e < p.W * GTR; e += GRAM_THREADS) {
        const int w = e / GTR, r = e - w * GTR;
        const int gi = row0 + r;
        rows[e] = gi < p.npad1 ? p.leaf1[((size_t)b * p.W + w) * p.npad1 + gi] : 0u;
    }
    for (int e = tid; e < p.W * GTC; e += GRAM_THREADS) {
        const int w = e / GTC, r = e - w * GTC;
        const int gj = col0 + r;
        cols[e] = gj < p.npad2 ? p.leaf2[((size_t)b * p.W + w) * p.npad2 + gj] : 0u;
    }
    __syncthreads();

    // thread -> 8 rows x 2 columns: column pair cx of the tile, row group rg (8 consecutive rows).  A wave's 64 lanes
    // are consecutive column pairs (GTC >= 128: one full 1 KiB run of a row per store instruction; GTC = 64: two
    // 512-byte runs of rows 8 apart).
    const int cx = tid % (GTC / 2), rg = tid / (GTC / 2);
    const int rloc = rg * 8;
    uint32_t cnt[8][2] = {};  // disagreeing trees (byte codes) / agreeing trees (bit code)
    for (int w = 0; w < p.W; ++w) {
        const uint4 ra = *reinterpret_cast<const uint4 *>(rows + w * GTR + rloc);
        const uint4 rb = *reinterpret_cast<const uint4 *>(rows + w * GTR + rloc + 4);
        const uint2 cc = *reinterpret_cast<const uint2 *>(cols + w * GTC + 2 * cx);
        const uint32_t r[8] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
#pragma unroll
        for (int a = 0; a < 8; ++a) {
            cnt[a][0] += code_count<REP>(r[a], cc.x);
            cnt[a][1] += code_count<REP>(r[a], cc.y);
        }
    }

    const double inv_m = 1.0 / (double)p.m;  // forest.py:88 `1 / nodes.shape[0]`
    const bool has_scale = p.scale != nullptr, has_shift = p.shift != nullptr;
    const double sh = has_shift ? p.shift[b] : 0.0;
    const double sc = has_scale ? p.scale[b] : 1.0;
    const double jitter = p.noise ? (1e-6 + p.noise[b]) : 0.0;  // tree_gps.py:100
    double *outb = p.out + (size_t)b * p.batch_stride;
    const int j0 = col0 + 2 * cx;

#pragma unroll
    for (int a = 0; a < 8; ++a) {
        const int i = row0 + rloc + a;
        if (i >= p.Nout) continue;
        double v[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int j = j0 + q;
            double val;
            if (i < p.N && j < p.M) {
                val = inv_m * (double)agree_count<REP>(cnt[a][q], p.m);  // unused byte lanes are equal, unused bits 0
                if (has_shift) val = val - sh;  // forest.py:111
                if (has_scale) val = sc * val;
                if (p.noise && i == j) val = val + jitter;
            } else {
                val = (p.pad_identity && i == j) ? 1.0 : 0.0;
            }
            v[q] = val;
        }
        double *dst = outb + (size_t)i * p.ld + j0;
        typedef double gram_d2 __attribute__((ext_vector_type(2)));  // streaming stores: the matrix is far larger than L2
        if (VEC2) {
            if (j0 + 1 < p.Mout)
                __builtin_nontemporal_store((gram_d2){v[0], v[1]}, reinterpret_cast<gram_d2 *>(dst));
            else if (j0 < p.Mout)
                dst[0] = v[0];
        } else {
            const double nxt = __shfl_down(v[0], 1);  // the next column pair's first value (same row for every lane but the last of a run)
            const bool off8 = (reinterpret_cast<uintptr_t>(dst) & 8) != 0;
            if (!off8) {
                if (j0 + 1 < p.Mout)
                    __builtin_nontemporal_store((gram_d2){v[0], v[1]}, reinterpret_cast<gram_d2 *>(dst));
                else if (j0 < p.Mout)
                    dst[0] = v[0];
            } else {
                const bool first = cx == 0 || (tid & 63) == 0, has_next = cx + 1 < GTC / 2 && (tid & 63) != 63;  // of a row run, within the wave
                if (first && j0 < p.Mout) dst[0] = v[0];
                if (has_next && j0 + 2 < p.Mout)
                    __builtin_nontemporal_store((gram_d2){v[1], nxt}, reinterpret_cast<gram_d2 *>(dst + 1));
                else if (j0 + 1 < p.Mout)
                    dst[1] = v[1];
            }
        }
    }
}

}  // namespace

// shared with chol.hip (the MLL engine fills its workspace with this kernel)
int launch_gram(const uint32_t *leaf1, int npad1, const uint32_t *leaf2, int npad2, int64_t B, int64_t m, int N, int M,
                int Nout, int Mout, const double *shift, const double *scale, const double *noise, double *out, int64_t ld,
                int64_t batch_stride, bool pad_identity, bool upper_only, int rep, int words, hipStream_t stream) {
    GramArgs p;
    p.shift = shift;
    p.leaf1 = leaf1;
    p.leaf2 = leaf2;
    p.npad1 = npad1;
    p.npad2 = npad2;
    p.W = words;
    p.m = (int)m;
    p.N = N;
    p.M = M;
    p.Nout = Nout;
    p.Mout = Mout;
    p.scale = scale;
    p.noise = noise;
    p.out = out;
    p.ld = ld;
    p.batch_stride = batch_stride;
    p.pad_identity = pad_identity;
    p.upper_only = upper_only;
    if (B > 65535) return fail(BARK_ERR_ARG, "gram: at most 65535 forests per call");
    const bool vec2 = (ld % 2 == 0) && (batch_stride % 2 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    auto launch = [&](auto tc_tag) -> int {
        constexpr int TC = decltype(tc_tag)::value, TR = 8 * (GRAM_THREADS / (TC / 2));
        dim3 grid((unsigned)((Mout + TC - 1) / TC), (unsigned)((Nout + TR - 1) / TR), (unsigned)B);
        const size_t lds = (size_t)p.W * (TR + TC) * sizeof(uint32_t);
        if (lds > 64 * 1024) return fail(BARK_ERR_ARG, "gram: too many trees (m=%lld)", (long long)m);
#define BARK_GRAM_LAUNCH(R)                                                                                   \
    do {                                                                                                      \
        if (vec2)                                                                                             \
            hipLaunchKernelGGL((gram_kernel<R, true, TC>), grid, dim3(GRAM_THREADS), lds, stream, p);         \
        else                                                                                                  \
            hipLaunchKernelGGL((gram_kernel<R, false, TC>), grid, dim3(GRAM_THREADS), lds, stream, p);        \
    } while (0)
        if (rep == REP_BITS)
            BARK_GRAM_LAUNCH(REP_BITS);
        else if (rep == REP_BYTES7)
            BARK_GRAM_LAUNCH(REP_BYTES7);
        else
            BARK_GRAM_LAUNCH(REP_BYTES8);
#undef BARK_GRAM_LAUNCH
        return BARK_OK;
    };
    constexpr int WIDE_TR = 8 * (GRAM_THREADS / (GRAM_TC_WIDE / 2));
    const bool wide = (size_t)p.W * (WIDE_TR + GRAM_TC_WIDE) * sizeof(uint32_t) <= 64 * 1024;
    const int rc = wide ? launch(std::integral_constant<int, GRAM_TC_WIDE>{}) : launch(std::integral_constant<int, GRAM_TC_NARROW>{});
    if (rc) return rc;
    BARK_LAUNCH_CHECK();
    return BARK_OK;
}

}  // namespace bark

using namespace bark;

extern "C" int bark_gram_from_leaves_hip(const uint32_t *leaf1, int64_t N, const uint32_t *leaf2, int64_t M,
                                         const bark_pack_info *info, const double *shift, const double *scale,
                                         const double *noise, double *out, int64_t ld, int64_t batch_stride,
                                         void *stream) {
    error_buffer()[0] = 0;
    if (!leaf1 || !leaf2 || !out || !info) return fail(BARK_ERR_ARG, "gram: null argument");
    const int64_t B = info->B, m = info->m;
    if (N < 1 || M < 1 || B < 1 || m < 1 || ld < M || N > (1 << 30) || M > (1 << 30))
        return fail(BARK_ERR_ARG, "gram: bad shape N=%lld M=%lld B=%lld m=%lld ld=%lld", (long long)N, (long long)M,
                    (long long)B, (long long)m, (long long)ld);
    return launch_gram(leaf1, (int)bark_leaf_npad(N), leaf2, (int)bark_leaf_npad(M), B, m, (int)N, (int)M, (int)N,
                       (int)M, shift, scale, noise, out, ld, batch_stride, false, false, (int)leaf_rep(info),
                       (int)bark_leaf_words(info), static_cast<hipStream_t>(stream));
}
