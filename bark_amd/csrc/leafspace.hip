// Leaf-space evaluation of the GP marginal log-likelihood (opt-in alternative to the dense N x N path).
//
// The forest kernel is K = (1/m) Z Z' with Z the N x R one-hot leaf-indicator matrix (R = sum over trees of
// their reachable leaves; the one-hot leaf code of traverse.hip is exactly the rows of Z).  For
// K_s = [scale] K + sigma2 I,  sigma2 = 1e-6 + noise,  c = scale / (m sigma2):
//     log|K_s|      = N log sigma2 + log|I_R + c Z'Z|                          (matrix determinant lemma)
//     y' K_s^-1 y   = ( y'y - c v' (I_R + c Z'Z)^-1 v ) / sigma2 ,  v = Z'y     (Woodbury)
// so the O(N^3) factorisation of the reference (mcmc_record_mll.py:69-70, bark_sampler.py:157-159) becomes an
// R x R one (R ~ 150 for 50 prior trees).  Z'Z[a][b] is a co-occurrence count = popcount over points of
// (bit-plane a AND bit-plane b); the R x R system is then handed to the same blocked Cholesky sweep (chol.hip).
// This path does NOT perform the Gram + N x N Cholesky work the benchmark metric counts and is never used
// by bench.py's timed region; it exists because it is the cheaper exact algorithm for the same quantity.
#include "common.h"

namespace bark {
namespace {

// codes (B, W, npad) uint32 [point fastest]  ->  planes (B, 32 W, Q) uint64, Q = npad / 64 [point-chunk fastest]:
// bit l of planes[b][a][q] says whether point 64 q + l reaches leaf a.  One wave per (q, w, b): 32 ballots.
__global__ __launch_bounds__(64) void bitplane_kernel(const uint32_t *__restrict__ codes, int W, int npad,
                                                      unsigned long long *__restrict__ planes) {
    const int q = blockIdx.x, w = blockIdx.y, b = blockIdx.z, lane = threadIdx.x;
    const int Q = npad >> 6;
    const uint32_t word = codes[((size_t)b * W + w) * npad + 64 * q + lane];
    unsigned long long mine = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const unsigned long long mask = __ballot((word >> k) & 1u);
        if (lane == k) mine = mask;
    }
    if (lane < 32) planes[((size_t)b * 32 * W + 32 * w + lane) * Q + q] = mine;
}

// M[a][c] = (a == c) + coef_b * sum_q popcount(planes[a][q] & planes[c][q])  for a, c < R; identity padding up to
// Rpad.  64 x 64 outputs per workgroup, plane rows staged through LDS in chunks of 32 words.
constexpr int CT = 64, CQ = 32;
__global__ __launch_bounds__(256) void cooc_kernel(const unsigned long long *__restrict__ planes, int rows_alloc, int Q,
                                                   int R, int Rpad, const double *__restrict__ noise,
                                                   const double *__restrict__ scale, int m, double *__restrict__ A,
                                                   long ld, long bstride) {
    __shared__ unsigned long long pa[CT][CQ + 1], pc[CT][CQ + 1];
    const int b = blockIdx.z, a0 = blockIdx.y * CT, c0 = blockIdx.x * CT, tid = threadIdx.x;
    const int ty = tid >> 4, tx = tid & 15;
    const unsigned long long *P = planes + (size_t)b * rows_alloc * Q;
    unsigned int cnt[4][4] = {};
    for (int q0 = 0; q0 < Q; q0 += CQ) {
        __syncthreads();
        for (int e = tid; e < CT * CQ; e += 256) {
            const int r = e / CQ, qq = e - r * CQ;
            const bool ok = q0 + qq < Q;
            pa[r][qq] = (ok && a0 + r < rows_alloc) ? P[(size_t)(a0 + r) * Q + q0 + qq] : 0ull;
            pc[r][qq] = (ok && c0 + r < rows_alloc) ? P[(size_t)(c0 + r) * Q + q0 + qq] : 0ull;
        }
        __syncthreads();
        for (int qq = 0; qq < CQ; ++qq) {
            unsigned long long ra[4], rc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i] = pa[ty * 4 + i][qq];
                rc[i] = pc[tx * 4 + i][qq];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) cnt[i][k] += __popcll(ra[i] & rc[k]);
        }
    }
    const double sigma2 = 1e-6 + noise[b];
    const double coef = (scale ? scale[b] : 1.0) / ((double)m * sigma2);
    double *Ab = A + (size_t)b * bstride;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int a = a0 + ty * 4 + i, c = c0 + tx * 4 + k;
            if (a >= Rpad || c >= Rpad) continue;
            double v = (a == c) ? 1.0 : 0.0;
            if (a < R && c < R) v += coef * (double)cnt[i][k];
            Ab[(size_t)a * ld + c] = v;
        }
}

// v[a] = sum_i [point i reaches leaf a] * y_i  -> right-hand side of the R x R system (zero padded to Rpad);
// also zeroes the accumulators / info of the sweep.  One wave per leaf.
__global__ __launch_bounds__(256) void leaf_sums_kernel(const unsigned long long *__restrict__ planes, int rows_alloc,
                                                        int Q, int R, int Rpad, const double *__restrict__ y, int N,
                                                        double *__restrict__ yz, double *__restrict__ accum,
                                                        int32_t *__restrict__ info) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int a = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        accum[(size_t)b * 2] = 0.0;
        accum[(size_t)b * 2 + 1] = 0.0;
        info[b] = 0;
    }
    if (a >= Rpad) return;
    double s = 0.0;
    if (a < R) {
        const unsigned long long *row = planes + ((size_t)b * rows_alloc + a) * Q;
        for (int q = 0; q < Q; ++q) {
            const unsigned long long mask = row[q];
            const int i = 64 * q + lane;
            if (((mask >> lane) & 1ull) && i < N) s += y[i];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) yz[(size_t)b * Rpad + a] = s;
}

// yy = y'y (one workgroup)
__global__ __launch_bounds__(256) void sumsq_kernel(const double *__restrict__ y, int N, double *__restrict__ out) {
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < N; i += 256) s = fma(y[i], y[i], s);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *out = red[0] + red[1] + red[2] + red[3];
}

// mll_b = -0.5 [ (yy - c quad_b) / sigma2 + N log sigma2 + logdetM_b (+ N log 2 pi) ]
__global__ void finish_leafspace_kernel(const double *__restrict__ accum, const double *__restrict__ yy,
                                        const double *__restrict__ noise, const double *__restrict__ scale, int m,
                                        int Bc, int N, int include_2pi, double *__restrict__ mll) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= Bc) return;
    const double sigma2 = 1e-6 + noise[b];
    const double coef = (scale ? scale[b] : 1.0) / ((double)m * sigma2);
    double v = (*yy - coef * accum[(size_t)b * 2]) / sigma2 + (double)N * log(sigma2) + accum[(size_t)b * 2 + 1];
    if (include_2pi) v = v + (double)N * log(2.0 * M_PI);
    mll[b] = -0.5 * v;
}

// Posterior at the candidates from the leaf-space quantities (derivation in include/bark_hip.h):
//     mu_c  = coef * sum_{a in L(c)} w[a]                    w = M^-1 v
//     var_c = (scale / m) * sum_{a, b in L(c)} Minv[a][b]     L(c) = the m leaves candidate c reaches
// One thread per (candidate, forest); its leaf positions are decoded from the candidate's one-hot code.
constexpr int LP_MAX_TREES = 64;
__global__ __launch_bounds__(128) void leaf_predict_kernel(const uint32_t *__restrict__ ccodes, int W, int cpad, int C,
                                                           const double *__restrict__ w, const double *__restrict__ Minv,
                                                           int R, const double *__restrict__ noise,
                                                           const double *__restrict__ scale, int m,
                                                           double *__restrict__ mu, double *__restrict__ var) {
    const int b = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    unsigned short leaf[LP_MAX_TREES];
    int n = 0;
    for (int wd = 0; wd < W; ++wd) {
        uint32_t bits = ccodes[((size_t)b * W + wd) * cpad + c];
        while (bits && n < LP_MAX_TREES) {
            const int k = __ffs(bits) - 1;
            bits &= bits - 1;
            leaf[n++] = (unsigned short)(32 * wd + k);
        }
    }
    const double *wb = w + (size_t)b * R;
    const double *Mb = Minv + (size_t)b * R * R;
    double s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < n; ++i) {
        s1 += wb[leaf[i]];
        const double *row = Mb + (size_t)leaf[i] * R;
        double r = 0.0;
        for (int k = 0; k < n; ++k) r += row[leaf[k]];
        s2 += r;
    }
    const double sigma2 = 1e-6 + noise[b];
    const double sc = scale[b];
    mu[(size_t)b * C + c] = sc / ((double)m * sigma2) * s1;
    var[(size_t)b * C + c] = sc / (double)m * s2;
}


// Explicit inverse in leaf space:  K_s^-1 = (I - c Z M^-1 Z') / sigma2,   K_s^-1 y = (y - c Z w) / sigma2.
// Step 1 (one wave per point i):  Wm[i][q] = sum_{a in L(i)} Minv[a][q]  (row i of Z M^-1) and zy_i = sum w_a.
// The leaf list is decoded from the one-hot code with wave-uniform control flow (no per-lane list).
__global__ __launch_bounds__(256) void leaf_rowsum_kernel(const uint32_t *__restrict__ codes, int W, int npad, int N,
                                                          const double *__restrict__ Minv, const double *__restrict__ w,
                                                          int R, const double *__restrict__ y,
                                                          const double *__restrict__ noise, const double *__restrict__ scale,
                                                          int m, double *__restrict__ Wm, double *__restrict__ kinv_y) {
    const int lane = threadIdx.x & 63, b = blockIdx.y;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= N) return;  // no barriers below
    const double *Mb = Minv + (size_t)b * R * R;
    const double *wb = w + (size_t)b * R;
    double *dst = Wm + ((size_t)b * N + i) * R;
    for (int qb = 0; qb < R; qb += 256) {
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        double sw = 0.0;
        for (int wd = 0; wd < W; ++wd) {
            uint32_t bits = __builtin_amdgcn_readfirstlane(codes[((size_t)b * W + wd) * npad + i]);
            while (bits) {
                const int a = 32 * wd + __builtin_ctz(bits);
                bits &= bits - 1;
                const double *row = Mb + (size_t)a * R + qb + lane;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (qb + 64 * u + lane < R) acc[u] += row[64 * u];
                sw += wb[a];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (qb + 64 * u + lane < R) dst[qb + 64 * u + lane] = acc[u];
        if (qb == 0 && lane == 0 && kinv_y) {
            const double sigma2 = 1e-6 + noise[b];
            const double coef = (scale ? scale[b] : 1.0) / ((double)m * sigma2);
            kinv_y[(size_t)b * N + i] = (y[i] - coef * sw) / sigma2;
        }
    }
}

// Step 2 (64 x 64 tile of the output):  out[i][j] = ([i == j] - c sum_{q in L(j)} Wm[i][q]) / sigma2.
// The leaf lists of the tile's 64 columns sit in LDS ([tree][column], conflict-free); a wave owns a row at a time,
// so its gathers stay inside one row of Wm (R doubles, L1-resident) and the store is one 512-byte segment.
__global__ __launch_bounds__(256) void leaf_inverse_kernel(const uint32_t *__restrict__ codes, int W, int npad, int N,
                                                           const double *__restrict__ Wm, int R,
                                                           const double *__restrict__ noise, const double *__restrict__ scale,
                                                           int m, double *__restrict__ out) {
    extern __shared__ unsigned short idx[];  // [m][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.z;
    const int col0 = blockIdx.x * 64, row0 = blockIdx.y * 64;
    const int j = col0 + lane;
    if (threadIdx.x < 64) {
        int n = 0;
        if (j < N)
            for (int wd = 0; wd < W && n < m; ++wd) {
                uint32_t bits = codes[((size_t)b * W + wd) * npad + j];
                while (bits && n < m) {
                    idx[n * 64 + lane] = (unsigned short)(32 * wd + __builtin_ctz(bits));
                    bits &= bits - 1;
                    ++n;
                }
            }
        for (; n < m; ++n) idx[n * 64 + lane] = 0;
    }
    __syncthreads();
    const double sigma2 = 1e-6 + noise[b];
    const double coef = (scale ? scale[b] : 1.0) / ((double)m * sigma2);
    const double inv_s2 = 1.0 / sigma2;
    for (int q = wave; q < 64; q += 4) {
        const int i = row0 + q;
        if (i >= N || j >= N) continue;
        const double *wrow = Wm + ((size_t)b * N + i) * R;
        double acc = 0.0;
#pragma unroll 4
        for (int t = 0; t < m; ++t) acc += wrow[idx[t * 64 + lane]];
        out[((size_t)b * N + i) * N + j] = ((i == j ? 1.0 : 0.0) - coef * acc) * inv_s2;
    }
}

}  // namespace

int leafspace_predict(const uint32_t *ccodes, int W, int cpad, int C, const double *w, const double *Minv, int R,
                      const double *noise, const double *scale, int m, int bc, double *mu, double *var, hipStream_t s) {
    if (m > LP_MAX_TREES) return fail(BARK_ERR_ARG, "leaf-space posterior supports at most %d trees", LP_MAX_TREES);
    hipLaunchKernelGGL(leaf_predict_kernel, dim3((unsigned)((C + 127) / 128), (unsigned)bc), dim3(128), 0, s, ccodes, W, cpad, C, w,
                       Minv, R, noise, scale, m, mu, var);
    BARK_LAUNCH_CHECK();
    return BARK_OK;
}

int leafspace_inverse(const uint32_t *codes, int W, int npad, int N, const double *Minv, const double *w, int R,
                      const double *y, const double *noise, const double *scale, int m, int bc, double *Wm, double *kinv,
                      double *kinv_y, hipStream_t s) {
    if (R > 65535 || m > 16384) return fail(BARK_ERR_ARG, "leaf-space inverse: forest too large (R = %d, m = %d)", R, m);
    hipLaunchKernelGGL(leaf_rowsum_kernel, dim3((unsigned)((N + 3) / 4), (unsigned)bc), dim3(256), 0, s, codes, W, npad, N, Minv,
                       w, R, y, noise, scale, m, Wm, kinv_y);
    BARK_LAUNCH_CHECK();
    const unsigned tiles = (unsigned)((N + 63) / 64);
    hipLaunchKernelGGL(leaf_inverse_kernel, dim3(tiles, tiles, (unsigned)bc), dim3(256), (size_t)m * 64 * sizeof(unsigned short),
                       s, codes, W, npad, N, Wm, R, noise, scale, m, kinv);
    BARK_LAUNCH_CHECK();
    return BARK_OK;
}

// launchers used by the entry point in chol.hip -----------------------------------------------------------
int leafspace_prepare(const uint32_t *codes, int W, int npad, unsigned long long *planes, int R, int Rpad,
                      const double *noise, const double *scale, int m, int bc, double *A, long ld, long bstride,
                      const double *y, int N, double *yz, double *accum, int32_t *info, hipStream_t s) {
    const int Q = npad / 64, rows_alloc = 32 * W;
    hipLaunchKernelGGL(bitplane_kernel, dim3((unsigned)Q, (unsigned)W, (unsigned)bc), dim3(64), 0, s, codes, W, npad, planes);
    BARK_LAUNCH_CHECK();
    const unsigned gt = (unsigned)((Rpad + CT - 1) / CT);
    hipLaunchKernelGGL(cooc_kernel, dim3(gt, gt, (unsigned)bc), dim3(256), 0, s, planes, rows_alloc, Q, R, Rpad, noise, scale, m,
                       A, ld, bstride);
    BARK_LAUNCH_CHECK();
    hipLaunchKernelGGL(leaf_sums_kernel, dim3((unsigned)((Rpad + 3) / 4), (unsigned)bc), dim3(256), 0, s, planes, rows_alloc, Q,
                       R, Rpad, y, N, yz, accum, info);
    BARK_LAUNCH_CHECK();
    return BARK_OK;
}

int leafspace_sumsq(const double *y, int N, double *out, hipStream_t s) {
    hipLaunchKernelGGL(sumsq_kernel, dim3(1), dim3(256), 0, s, y, N, out);
    BARK_LAUNCH_CHECK();
    return BARK_OK;
}

int leafspace_finish(const double *accum, const double *yy, const double *noise, const double *scale, int m, int bc, int N,
                     int include_2pi, double *mll, hipStream_t s) {
    hipLaunchKernelGGL(finish_leafspace_kernel, dim3((unsigned)((bc + 255) / 256)), dim3(256), 0, s, accum, yy, noise, scale, m,
                       bc, N, include_2pi, mll);
    BARK_LAUNCH_CHECK();
    return BARK_OK;
}

}  // namespace bark
