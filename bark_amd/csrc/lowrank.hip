// Woodbury / matrix-determinant-lemma updates on gfx950 — src/bark/fitting/quick_inverse.py:13-33.
//
//   low_rank_inv_update(K_inv, U, subtract):  K_inv - K_inv U (mul I + U' K_inv U)^-1 U' K_inv      (:13-21)
//   low_rank_det_update(K_inv, U, logdet, s): logdet + log|det(I + mul U' K_inv U)|                 (:24-33)
//
// with mul = -1 for subtract.  In the sampler U is N x r (r = #leaves of one tree, single digits), so
// the work is two streaming passes over the N x N matrix and r x r algebra in between:
//   1. skinny_kernel   Y = K_inv U            one wave per row, 64-wide coalesced reads   (HBM read, 8 N^2 B)
//   2. small_kernel    G = U'Y, den = mul I + G, LU with partial pivoting -> den^-1, log|det|;  M = Y den^-1
//                      (one workgroup; r <= 64)
//   3. rank_update_kernel   out = K_inv - M Y'   64x64 tiles, M/Y strips in LDS          (HBM read+write, 16 N^2 B)
// K_inv is treated as a general (not necessarily symmetric) matrix exactly as the reference's formula does:
// the right factor is U' K_inv, computed as its own pass when `symmetric == 0`.
#include "common.h"

namespace bark {
namespace {

constexpr int LR_MAX = 64;  // max rank r
constexpr int LR_THREADS = 256;

// out[i][c] = sum_k K[i][k] * U[k][c]      (TRANS == 0)   one wave per output row i
// out[i][c] = sum_k K[k][i] * U[k][c]      (TRANS == 1)   (= (U' K)' : column i of K), lanes stride rows
template <int TRANS, int RT>  // RT: compile-time bound on r (registers)
__global__ __launch_bounds__(LR_THREADS) void skinny_kernel(const double *__restrict__ K, const double *__restrict__ U,
                                                            int N, int r, double *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double us[];  // [64][r] chunk of U
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * (LR_THREADS / 64) + wave;  // output row of this wave
    double s[RT];
#pragma unroll
    for (int c = 0; c < RT; ++c) s[c] = 0.0;
    for (int k0 = 0; k0 < N; k0 += 64) {
        __syncthreads();
        for (int e = threadIdx.x; e < 64 * r; e += LR_THREADS) {
            const int kk = e / r;
            us[e] = (k0 + kk < N) ? U[(size_t)(k0 + kk) * r + (e - kk * r)] : 0.0;
        }
        __syncthreads();
        const int k = k0 + lane;
        double kv = 0.0;
        if (i < N && k < N) kv = TRANS ? K[(size_t)k * N + i] : K[(size_t)i * N + k];
#pragma unroll
        for (int c = 0; c < RT; ++c)
            if (c < r) s[c] = fma(kv, us[lane * r + c], s[c]);
    }
#pragma unroll
    for (int c = 0; c < RT; ++c) {
        if (c < r) {
            double v = s[c];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0 && i < N) out[(size_t)i * r + c] = v;
        }
    }
}

// One workgroup: G = U'Y (r x r), den = C + G with C = diag(+-1) (the first r_neg columns carry -1: leaf
// vectors being removed), Gauss-Jordan with partial pivoting -> den^-1 and log|det den|; then
// M = Y den^-1 (N x r).  With `y`: also v = Y'y and dquad = v' den^-1 v (the change of y'K^-1 y).
// `logabsdet`, `M`, `y`/`dquad` may be null.
__global__ __launch_bounds__(LR_THREADS) void small_kernel(const double *__restrict__ U, const double *__restrict__ Y,
                                                           int N, int r, int r_neg, double *__restrict__ M,
                                                           double *__restrict__ logabsdet, int *__restrict__ singular,
                                                           const double *__restrict__ y, double *__restrict__ dquad) {
    __shared__ double aug[LR_MAX][2 * LR_MAX + 1];  // [den | I]
    __shared__ int piv_row;
    const int tid = threadIdx.x;
    // G[a][b] = sum_k U[k][a] Y[k][b]: one wave per (a,b) pair, lanes stride k, shuffle reduce
    for (int e = tid >> 6; e < r * r; e += LR_THREADS / 64) {
        const int a = e / r, bcol = e - a * r;
        double g = 0.0;
        for (int k = tid & 63; k < N; k += 64) g = fma(U[(size_t)k * r + a], Y[(size_t)k * r + bcol], g);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) g += __shfl_xor(g, off);
        if ((tid & 63) == 0) {
            aug[a][bcol] = g + (a == bcol ? (a < r_neg ? -1.0 : 1.0) : 0.0);
            aug[a][r + bcol] = (a == bcol) ? 1.0 : 0.0;
        }
    }
    __syncthreads();
    double logsum = 0.0;
    for (int col = 0; col < r; ++col) {
        if (tid == 0) {  // partial pivoting (np.linalg.solve / slogdet use LU with row pivoting)
            int best = col;
            double bv = fabs(aug[col][col]);
            for (int a = col + 1; a < r; ++a)
                if (fabs(aug[a][col]) > bv) {
                    bv = fabs(aug[a][col]);
                    best = a;
                }
            piv_row = best;
            if (bv == 0.0 && singular) *singular = col + 1;
        }
        __syncthreads();
        const int pr = piv_row;
        if (pr != col)
            for (int e = tid; e < 2 * r; e += LR_THREADS) {
                const double t = aug[col][e];
                aug[col][e] = aug[pr][e];
                aug[pr][e] = t;
            }
        __syncthreads();
        const double d = aug[col][col];
        logsum += log(fabs(d));
        __syncthreads();
        for (int e = tid; e < 2 * r; e += LR_THREADS) aug[col][e] = aug[col][e] / d;
        __syncthreads();
        for (int e = tid; e < r * 2 * r; e += LR_THREADS) {
            const int a = e / (2 * r), cc = e - a * 2 * r;
            if (a != col && cc != col) aug[a][cc] = fma(-aug[a][col], aug[col][cc], aug[a][cc]);
        }
        __syncthreads();
        for (int a = tid; a < r; a += LR_THREADS)
            if (a != col) aug[a][col] = 0.0;
        __syncthreads();
    }
    if (tid == 0 && logabsdet) *logabsdet = logsum;
    if (y && dquad) {  // v = Y'y (wave per column), dquad = v' den^-1 v
        __shared__ double vsh[LR_MAX];
        for (int a = tid >> 6; a < r; a += LR_THREADS / 64) {
            double s = 0.0;
            for (int k = tid & 63; k < N; k += 64) s = fma(Y[(size_t)k * r + a], y[k], s);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
            if ((tid & 63) == 0) vsh[a] = s;
        }
        __syncthreads();
        if (tid == 0) {
            double q = 0.0;
            for (int a = 0; a < r; ++a)
                for (int b2 = 0; b2 < r; ++b2) q = fma(vsh[a] * aug[a][r + b2], vsh[b2], q);
            *dquad = q;
        }
    }
    if (M)
        for (int e = tid; e < N * r; e += LR_THREADS) {
            const int i = e / r, c = e - i * r;
            double s = 0.0;
            for (int a = 0; a < r; ++a) s = fma(Y[(size_t)i * r + a], aug[a][r + c], s);
            M[e] = s;
        }
}

// out[i][j] = K[i][j] - sum_a M[i][a] * R[j][a]     (R = Y for symmetric K_inv, else (U'K_inv)')
__global__ __launch_bounds__(LR_THREADS) void rank_update_kernel(const double *__restrict__ K,
                                                                 const double *__restrict__ M,
                                                                 const double *__restrict__ R, int N, int r,
                                                                 double *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double strips[];  // Ms[64][r] | Rs[64][r]
    double *Ms = strips, *Rs = strips + 64 * r;
    const int row0 = blockIdx.y * 64, col0 = blockIdx.x * 64;
    for (int e = threadIdx.x; e < 64 * r; e += LR_THREADS) {
        const int q = e / r, a = e - q * r;
        Ms[e] = (row0 + q < N) ? M[(size_t)(row0 + q) * r + a] : 0.0;
        Rs[e] = (col0 + q < N) ? R[(size_t)(col0 + q) * r + a] : 0.0;
    }
    __syncthreads();
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // lanes run along a row: coalesced 512 B
    const int j = col0 + tx;
    for (int q = ty; q < 64; q += LR_THREADS / 64) {
        const int i = row0 + q;
        if (i < N && j < N) {
            double s = 0.0;
            for (int a = 0; a < r; ++a) s = fma(Ms[q * r + a], Rs[tx * r + a], s);
            out[(size_t)i * N + j] = K[(size_t)i * N + j] - s;
        }
    }
}

template <int TRANS>
void launch_skinny(dim3 grid, size_t lds, hipStream_t stream, const double *K, const double *U, int N, int r,
                   double *out) {
    if (r <= 8)
        hipLaunchKernelGGL((skinny_kernel<TRANS, 8>), grid, dim3(LR_THREADS), lds, stream, K, U, N, r, out);
    else if (r <= 16)
        hipLaunchKernelGGL((skinny_kernel<TRANS, 16>), grid, dim3(LR_THREADS), lds, stream, K, U, N, r, out);
    else if (r <= 32)
        hipLaunchKernelGGL((skinny_kernel<TRANS, 32>), grid, dim3(LR_THREADS), lds, stream, K, U, N, r, out);
    else
        hipLaunchKernelGGL((skinny_kernel<TRANS, 64>), grid, dim3(LR_THREADS), lds, stream, K, U, N, r, out);
}

}  // namespace
}  // namespace bark

using namespace bark;

extern "C" {

size_t bark_lowrank_workspace_bytes(int64_t N, int64_t r) {
    if (N < 1 || r < 1 || r > LR_MAX) return 0;
    return (size_t)(3 * N * r) * sizeof(double) + 64;  // Y, R, M + flag
}

int bark_lowrank_update_hip(const double *K_inv, int64_t N, const double *U, int64_t r, int subtract, int symmetric,
                            double *K_out, double *logabsdet_out, void *workspace, size_t workspace_bytes,
                            void *stream_) {
    error_buffer()[0] = 0;
    if (!K_inv || !U || !workspace || N < 1 || r < 1 || N > (1 << 24))
        return fail(BARK_ERR_ARG, "bark_lowrank_update_hip: bad argument");
    if (r > LR_MAX) return fail(BARK_ERR_ARG, "low-rank update supports rank <= %d (got %lld)", LR_MAX, (long long)r);
    if (workspace_bytes < bark_lowrank_workspace_bytes(N, r))
        return fail(BARK_ERR_WORKSPACE, "low-rank workspace too small");
    if (!K_out && !logabsdet_out) return fail(BARK_ERR_ARG, "nothing to compute");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    double *Y = static_cast<double *>(workspace);
    double *R = Y + N * r;
    double *M = R + N * r;
    int *flag = reinterpret_cast<int *>(M + N * r);
    const int r_neg = subtract ? (int)r : 0;
    const unsigned rows_per_block = LR_THREADS / 64;
    const dim3 g1((unsigned)((N + rows_per_block - 1) / rows_per_block));
    const size_t lds1 = (size_t)64 * r * sizeof(double);
    launch_skinny<0>(g1, lds1, stream, K_inv, U, (int)N, (int)r, Y);
    BARK_LAUNCH_CHECK();
    const double *Rp = Y;
    if (K_out && !symmetric) {  // right factor U' K_inv as its own pass
        launch_skinny<1>(g1, lds1, stream, K_inv, U, (int)N, (int)r, R);
        BARK_LAUNCH_CHECK();
        Rp = R;
    }
    BARK_HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(int), stream));
    hipLaunchKernelGGL(small_kernel, dim3(1), dim3(LR_THREADS), 0, stream, U, Y, (int)N, (int)r, r_neg,
                       K_out ? M : nullptr, logabsdet_out, flag, (const double *)nullptr, (double *)nullptr);
    BARK_LAUNCH_CHECK();
    if (K_out) {
        const dim3 g3((unsigned)((N + 63) / 64), (unsigned)((N + 63) / 64));
        hipLaunchKernelGGL(rank_update_kernel, g3, dim3(LR_THREADS), (size_t)2 * 64 * r * sizeof(double), stream, K_inv,
                           M, Rp, (int)N, (int)r, K_out);
        BARK_LAUNCH_CHECK();
    }
    return BARK_OK;
}

// ---- fused tree swap (the per-tree step of the sampler, bark_sampler.py:233-257) -----------------
// The reference evaluates a proposal by  subtract(U_old) -> add(U_new) -> mll  (2 Woodbury + 2 determinant
// updates + one quadratic form, ~9 passes over the N x N inverse) before it knows whether to accept.
// With U = [U_old U_new], C = diag(-I, +I), Y = K^-1 U, G = U'Y, v = Y'y:
//     log|K'| = log|K| + log|det(C + G)|,     y'K'^-1 y = y'K^-1 y - v'(C + G)^-1 v,
//     K'^-1   = K^-1 - Y (C + G)^-1 Y'        (only needed when the proposal is accepted).
// bark_lowrank_swap_eval_hip makes ONE pass over K^-1 and leaves Y, M = Y (C+G)^-1 in the workspace;
// bark_lowrank_swap_apply_hip performs the rank-(r_old + r_new) update from them (read + write pass).
// K_inv must be symmetric (it is an SPD inverse).  scalars_out (device): {dquad, log|det(C+G)|}.
int bark_lowrank_swap_eval_hip(const double *K_inv, int64_t N, const double *U, int64_t r_old, int64_t r_new,
                               const double *y, double *scalars_out, void *workspace, size_t workspace_bytes,
                               void *stream_) {
    error_buffer()[0] = 0;
    const int64_t r = r_old + r_new;
    if (!K_inv || !U || !y || !scalars_out || !workspace || N < 1 || r_old < 0 || r_new < 0 || r < 1 || N > (1 << 24))
        return fail(BARK_ERR_ARG, "bark_lowrank_swap_eval_hip: bad argument");
    if (r > LR_MAX) return fail(BARK_ERR_ARG, "tree swap supports r_old + r_new <= %d (got %lld)", LR_MAX, (long long)r);
    if (workspace_bytes < bark_lowrank_workspace_bytes(N, r)) return fail(BARK_ERR_WORKSPACE, "low-rank workspace too small");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    double *Y = static_cast<double *>(workspace);
    double *M = Y + 2 * N * r;
    int *flag = reinterpret_cast<int *>(M + N * r);
    const unsigned rows_per_block = LR_THREADS / 64;
    const dim3 g1((unsigned)((N + rows_per_block - 1) / rows_per_block));
    launch_skinny<0>(g1, (size_t)64 * r * sizeof(double), stream, K_inv, U, (int)N, (int)r, Y);
    BARK_LAUNCH_CHECK();
    BARK_HIP_CHECK(hipMemsetAsync(flag, 0, sizeof(int), stream));
    hipLaunchKernelGGL(small_kernel, dim3(1), dim3(LR_THREADS), 0, stream, U, Y, (int)N, (int)r, (int)r_old, M,
                       scalars_out + 1, flag, y, scalars_out);
    BARK_LAUNCH_CHECK();
    return BARK_OK;
}

int bark_lowrank_swap_apply_hip(const double *K_inv, int64_t N, int64_t r, const void *workspace, double *K_out,
                                void *stream_) {
    error_buffer()[0] = 0;
    if (!K_inv || !K_out || !workspace || N < 1 || r < 1 || r > LR_MAX)
        return fail(BARK_ERR_ARG, "bark_lowrank_swap_apply_hip: bad argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const double *Y = static_cast<const double *>(workspace);
    const double *M = Y + 2 * N * r;
    const dim3 g3((unsigned)((N + 63) / 64), (unsigned)((N + 63) / 64));
    hipLaunchKernelGGL(rank_update_kernel, g3, dim3(LR_THREADS), (size_t)2 * 64 * r * sizeof(double), stream, K_inv, M,
                       Y, (int)N, (int)r, K_out);
    BARK_LAUNCH_CHECK();
    return BARK_OK;
}

}  // extern "C"
