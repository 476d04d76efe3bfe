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
#include "common.h"

namespace bark {
namespace {

constexpr int GT = 64;  // output tile edge
constexpr int GRAM_THREADS = 256;

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

template <int REP, bool VEC2>
__global__ __launch_bounds__(GRAM_THREADS) void gram_kernel(GramArgs p) {
    extern __shared__ __attribute__((aligned(16))) uint32_t strips[];  // rows[W][64] | cols[W][64]
    const int tid = threadIdx.x;
    const int b = blockIdx.z;
    const int row0 = blockIdx.y * GT, col0 = blockIdx.x * GT;
    if (p.upper_only && (col0 >> 7) < (row0 >> 7)) return;

    uint32_t *rows = strips;
    uint32_t *cols = strips + p.W * GT;
    for (int e = tid; e < p.W * GT; e += GRAM_THREADS) {
        const int w = e >> 6, r = e & 63;
        const int gi = row0 + r, gj = col0 + r;
        rows[e] = gi < p.npad1 ? p.leaf1[((size_t)b * p.W + w) * p.npad1 + gi] : 0u;
        cols[e] = gj < p.npad2 ? p.leaf2[((size_t)b * p.W + w) * p.npad2 + gj] : 0u;
    }
    __syncthreads();

    // thread -> 8 rows x 2 columns: wave w owns rows 16w..16w+15, half-wave h rows 8h..8h+7 of those, lane pair
    // column 2*cx.  One store instruction then writes two full 512-byte row segments (32 lanes x 16 B each).
    const int wave = tid >> 6, lane = tid & 63, half = lane >> 5, cx = lane & 31;
    const int rloc = wave * 16 + half * 8;
    uint32_t cnt[8][2] = {};  // disagreeing trees (byte codes) / agreeing trees (bit code)
    for (int w = 0; w < p.W; ++w) {
        const uint4 ra = *reinterpret_cast<const uint4 *>(rows + w * GT + rloc);
        const uint4 rb = *reinterpret_cast<const uint4 *>(rows + w * GT + rloc + 4);
        const uint2 cc = *reinterpret_cast<const uint2 *>(cols + w * GT + 2 * cx);
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
        if (VEC2 && j0 + 1 < p.Mout) {
            *reinterpret_cast<double2 *>(dst) = make_double2(v[0], v[1]);
        } else {
            if (j0 < p.Mout) dst[0] = v[0];
            if (j0 + 1 < p.Mout) dst[1] = v[1];
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
    dim3 grid((unsigned)((Mout + GT - 1) / GT), (unsigned)((Nout + GT - 1) / GT), (unsigned)B);
    const size_t lds = (size_t)2 * p.W * GT * sizeof(uint32_t);
    if (lds > 64 * 1024) return fail(BARK_ERR_ARG, "gram: too many trees (m=%lld)", (long long)m);
    const bool vec2 = (ld % 2 == 0) && (batch_stride % 2 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
#define BARK_GRAM_LAUNCH(R)                                                                                   \
    do {                                                                                                      \
        if (vec2)                                                                                             \
            hipLaunchKernelGGL((gram_kernel<R, true>), grid, dim3(GRAM_THREADS), lds, stream, p);             \
        else                                                                                                  \
            hipLaunchKernelGGL((gram_kernel<R, false>), grid, dim3(GRAM_THREADS), lds, stream, p);            \
    } while (0)
    if (rep == REP_BITS)
        BARK_GRAM_LAUNCH(REP_BITS);
    else if (rep == REP_BYTES7)
        BARK_GRAM_LAUNCH(REP_BYTES7);
    else
        BARK_GRAM_LAUNCH(REP_BYTES8);
#undef BARK_GRAM_LAUNCH
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
