// chol_solve.h — part of the dense sweep's one translation unit (chol.hip includes it; kernels and helpers live in the
// anonymous namespace of that unit).  The solve kernels (U = W'T by the explicit block inverse, four widths), the right-hand-side / candidate kernels, and the
// finishing reductions (MLL, posterior mean / variance, V'V).
#pragma once
#include "chol_rows.h"

namespace bark {
namespace {

// ---------------------------------------------------------------------------------------------
// solve_kernel: U[j,i] = W_j' T[j,i] for every tile right of the diagonal; y_i -= U[j,i]' z_j.
// DEF == 1 (pipelined schedule): the stored tile lacks the last block row of its sum, T = T' - U[j-1,j]'U[j-1,i], and
//   U[j,i] = W_j' T' - (U[j-1,j] W_j)' U[j-1,i] = [-G_j ; W_j]' [U[j-1,i] ; T'[j,i]]
// is ONE K = 256 product: the right operand is the contiguous 256-row panel of column block i starting at block row
// j-1, the left one the stacked (256 x 128) block Mats::W that diag_kernel fills.
// ---------------------------------------------------------------------------------------------
template <int DEF>
__global__ __launch_bounds__(THREADS, 2) void solve_kernel(Mats p, int j, int n_right) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x;
    int b, t;
    if (!xcd_map(blockIdx.x, n_right, p.Bc, b, t)) return;
    const Lane q = lane_of(tid);
    const int cb = j + 1 + t;
    double *Ab = p.A + (size_t)b * p.bstride;
    double *tile = Ab + (size_t)j * NB * p.ld + (size_t)cb * NB;
    const double *Pb = p.W + (size_t)b * W_STRIDE + (size_t)(1 - DEF) * NB * NB;

    f64x4 acc[4][4];
    zero_acc(acc);
    // D[r][c] = sum_{k<=r} W[k][r] T[k][c]; this wave's 4 row tiles (16 rows each)
    // (readfirstlane: the skip branches around MFMAs must be scalar branches, MFMA ignores EXEC)
    const int wr_u = __builtin_amdgcn_readfirstlane(q.wr);
    const int rt[4] = {wr_u ? 1 : 0, wr_u ? 2 : 3, wr_u ? 5 : 4, wr_u ? 6 : 7};
    gemm_upper_tri<DEF>(acc, rt, Pb, tile - (size_t)DEF * NB * p.ld, p.ld, lds, tid, q);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            double *row = tile + (size_t)(rt[mt] * 16 + q.lk + 4 * v) * p.ld;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) row[acc_col(q, nt)] = acc[mt][nt][v];
        }
    if (cb >= p.nrb) return;  // candidate columns: no right-hand-side update

    // y_i[c] -= sum_r U[j,i][r][c] * z_j[r]   (summation order: y_partial / y_commit)
    double *part = lds;  // [8][128]; the GEMM ended with a barrier
    const double *zb = p.yz + (size_t)b * p.nrb * NB + (size_t)j * NB;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) y_partial<4>(acc[mt], rt[mt], zb, part, q.wc * 64, q);
    __syncthreads();
    if (tid < NB) y_commit(part, p.yz + (size_t)b * p.nrb * NB + (size_t)cb * NB, tid);
}

// ---------------------------------------------------------------------------------------------
// solve_narrow_kernel: the same U[j,i] = W_j' T[j,i] with a tile's 128 columns shared out over 4 / NT workgroups (32 NT
// columns each, 16 NT per wave) — for the critical path of lone matrices, where solve_kernel is one workgroup per tile on
// an otherwise idle chip and its 2.4 MFLOP of MFMA work on ONE CU (7.7 us of an 18 us launch) is what takes the time.
// Each workgroup still stages the whole k-rows of T (L2-resident: the reduce kernel just wrote them).  Per column the
// arithmetic and its order are those of solve_kernel: identical results.
// ---------------------------------------------------------------------------------------------
template <int NT, int DEF = 0>  // DEF == 1: the K = 256 product of solve_kernel<1> (dense block first, then W_j)
__global__ __launch_bounds__(THREADS, 2) void solve_narrow_kernel(Mats p, int j, int n_right) {
    constexpr int PARTS = 4 / NT, WCOLS = 32 * NT, nd = DEF * (NB / BK), nk = nd + NB / BK;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x;
    int b, tp;
    if (!xcd_map(blockIdx.x, n_right * PARTS, p.Bc, b, tp)) return;
    const Lane q = lane_of(tid);
    const int t = tp / PARTS, c_off = (tp - t * PARTS) * WCOLS + q.wc * 16 * NT;  // this wave's first column in the tile
    const int cb = j + 1 + t;
    double *tile = p.A + (size_t)b * p.bstride + (size_t)j * NB * p.ld + (size_t)cb * NB;
    const double *Wb = p.W + (size_t)b * W_STRIDE + (size_t)(1 - DEF) * NB * NB;  // [-G_j ;] W_j
    const double *Tp = tile - (size_t)DEF * NB * p.ld;                              // [U[j-1,i] ;] T[j,i]
    const int wr_u = __builtin_amdgcn_readfirstlane(q.wr);
    const int rt[4] = {wr_u ? 1 : 0, wr_u ? 2 : 3, wr_u ? 5 : 4, wr_u ? 6 : 7};
    f64x4 acc[4][NT];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f64x4){0.0, 0.0, 0.0, 0.0};
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    stage_dma(Wb, NB, Tp, p.ld, 0, lds, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) stage_dma(Wb, NB, Tp, p.ld, kt + 1, lds + ((kt + 1) & 1) * STAGE, wave, lane);
        const double *As = lds + (kt & 1) * STAGE;
        const double *Bs = As + BK * LDS_LD;
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            double bf[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bf[nt] = Bs[(kk * 4 + q.lk) * LDS_LD + c_off + nt * 16 + q.lr];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                if (kt < nd || kt - nd <= rt[mt]) {  // wave-uniform: dense block, then the triangular skip
                    const double a = As[(kk * 4 + q.lk) * LDS_LD + rt[mt] * 16 + q.lr];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bf[nt], acc[mt][nt], 0, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            double *row = tile + (size_t)(rt[mt] * 16 + q.lk + 4 * v) * p.ld + c_off + q.lr;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) row[nt * 16] = acc[mt][nt][v];
        }
    if (cb >= p.nrb) return;  // candidate columns: no right-hand-side update
    double *part = lds;  // [8][128]; only this workgroup's WCOLS columns are written and read
    const double *zb = p.yz + (size_t)b * p.nrb * NB + (size_t)j * NB;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) y_partial<NT>(acc[mt], rt[mt], zb, part, c_off, q);
    __syncthreads();
    const int c0 = (tp - t * PARTS) * WCOLS;
    if (tid < WCOLS) y_commit(part, p.yz + (size_t)b * p.nrb * NB + (size_t)cb * NB, c0 + tid);
}

// ---------------------------------------------------------------------------------------------
// solve_direct_kernel: the same U[j,i] = W_j' T[j,i] for the critical path of lone matrices, where the launch is bound by
// LATENCY, not work: solve_narrow_kernel stages eight (DEF: sixteen) k-tiles through LDS one DMA round trip after the other
// (10.4 us per launch whether it solves one tile or thirty-one — a sixth of a lone matrix's block step).  Here a tile's 128
// columns go to 8 workgroups of 16, wave w owns the row tiles w and 7 - w (9 of the 36 non-zero (k-tile, row-tile) products
// each), and both MFMA operands come straight from L2 — diag_kernel and the reduce kernel have just written them — with
// every load of the wave independent of the others (straight-line code per wave: template on the wave index).  Per element the
// MFMA sequence (k ascending, four k per MFMA, zero k-tiles of W_j skipped) and the right-hand-side update are those of
// solve_kernel: identical results.
// ---------------------------------------------------------------------------------------------
template <int DEF, int WV>
__device__ __forceinline__ void solve_direct_wave(const double *__restrict__ Wl, const double *__restrict__ Tl, long ld, f64x4 &accA,
                                                  f64x4 &accB) {
    constexpr int rtA = WV, rtB = 7 - WV;  // rtA < rtB
    if (DEF) {  // dense block [-G_j]' U[j-1,i] first
#pragma unroll
        for (int ks = 0; ks < NB / 4; ++ks) {
            const double bv = Tl[(size_t)(ks * 4) * ld];
            accA = __builtin_amdgcn_mfma_f64_16x16x4f64(Wl[(ks * 4) * NB + rtA * 16], bv, accA, 0, 0, 0);
            accB = __builtin_amdgcn_mfma_f64_16x16x4f64(Wl[(ks * 4) * NB + rtB * 16], bv, accB, 0, 0, 0);
        }
        Wl += (size_t)NB * NB;
        Tl += (size_t)NB * ld;
    }
#pragma unroll
    for (int ks = 0; ks < 4 * (rtB + 1); ++ks) {
        const double bv = Tl[(size_t)(ks * 4) * ld];
        if (ks < 4 * (rtA + 1)) accA = __builtin_amdgcn_mfma_f64_16x16x4f64(Wl[(ks * 4) * NB + rtA * 16], bv, accA, 0, 0, 0);
        accB = __builtin_amdgcn_mfma_f64_16x16x4f64(Wl[(ks * 4) * NB + rtB * 16], bv, accB, 0, 0, 0);
    }
}

template <int DEF>
__global__ __launch_bounds__(THREADS) void solve_direct_kernel(Mats p, int j, int n_right) {
    __shared__ double part[NSB_ROWS * NB];
    const int tid = threadIdx.x;
    int b, ts;
    if (!xcd_map(blockIdx.x, n_right * 8, p.Bc, b, ts)) return;
    const Lane q = lane_of(tid);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t = ts >> 3, c0 = (ts & 7) * 16, cb = j + 1 + t;
    double *tile = p.A + (size_t)b * p.bstride + (size_t)j * NB * p.ld + (size_t)cb * NB;
    const double *Wl = p.W + (size_t)b * W_STRIDE + (size_t)(1 - DEF) * NB * NB + (size_t)q.lk * NB + q.lr;  // [-G_j ;] W_j
    const double *Tl = tile - (size_t)DEF * NB * p.ld + (size_t)q.lk * p.ld + c0 + q.lr;                       // [U[j-1,i] ;] T[j,i]
    f64x4 acc[2][1] = {{{0.0, 0.0, 0.0, 0.0}}, {{0.0, 0.0, 0.0, 0.0}}};
    if (wave == 0)
        solve_direct_wave<DEF, 0>(Wl, Tl, p.ld, acc[0][0], acc[1][0]);
    else if (wave == 1)
        solve_direct_wave<DEF, 1>(Wl, Tl, p.ld, acc[0][0], acc[1][0]);
    else if (wave == 2)
        solve_direct_wave<DEF, 2>(Wl, Tl, p.ld, acc[0][0], acc[1][0]);
    else
        solve_direct_wave<DEF, 3>(Wl, Tl, p.ld, acc[0][0], acc[1][0]);
    const int rt[2] = {wave, 7 - wave};
    __syncthreads();  // in place: every wave has read its T rows of the strip before any row of it is overwritten
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int v = 0; v < 4; ++v) tile[(size_t)(rt[h] * 16 + q.lk + 4 * v) * p.ld + c0 + q.lr] = acc[h][0][v];
    if (cb >= p.nrb) return;  // candidate columns: no right-hand-side update
    const double *zb = p.yz + (size_t)b * p.nrb * NB + (size_t)j * NB;
#pragma unroll
    for (int h = 0; h < 2; ++h) y_partial<1>(acc[h], rt[h], zb, part, c0, q);
    __syncthreads();
    if (tid < 16) y_commit(part, p.yz + (size_t)b * p.nrb * NB + (size_t)cb * NB, c0 + tid);
}

// right-hand-side block := identity (N x N inside the padded candidate columns)
__global__ void identity_rhs_kernel(Mats p, int N, int cpad) {
    const int b = blockIdx.z, r = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < cpad) p.A[(size_t)b * p.bstride + (size_t)r * p.ld + (size_t)p.nrb * NB + c] = (r == c && r < N) ? 1.0 : 0.0;
}

// ---------------------------------------------------------------------------------------------
// vtv_kernel: out[ci][cj] = base + sign * sum_k V[k][ci] V[k][cj] over the candidate columns
// (V = U^-T K_Xx sits in the extra block columns after the sweep).  Same k-major MFMA product as the
// panel kernel.  `tri`: V = U^-T is lower triangular (identity right-hand side), so the sum starts at
// block row max(ti, tj).  Full covariance: base = scale_b, sign = -1.  Inverse: base = 0, sign = +1.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(THREADS, 2) void vtv_kernel(Mats p, int nct, int C, const double *base, double sign,
                                                          int tri, double *out) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x;
    int b, t;
    if (!xcd_map(blockIdx.x, nct * nct, p.Bc, b, t)) return;
    const Lane q = lane_of(tid);
    const int ti = t / nct, tj = t - ti * nct;
    const int kb = tri ? (ti > tj ? ti : tj) : 0;
    const double *Vb = p.A + (size_t)b * p.bstride + (size_t)kb * NB * p.ld + (size_t)p.nrb * NB;
    f64x4 acc[4][4];
    zero_acc(acc);
    gemm_kmajor_dma(acc, Vb + (size_t)ti * NB, p.ld, Vb + (size_t)tj * NB, p.ld, (p.nrb - kb) * NB, lds, tid, q);
    const double bs = base ? base[b] : 0.0;
    double *ob = out + (size_t)b * C * C;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int r = ti * NB + acc_row(q, mt, v);
            if (r >= C) continue;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int cc = tj * NB + acc_col(q, nt);
                if (cc < C) ob[(size_t)r * C + cc] = bs + sign * acc[mt][nt][v];
            }
        }
}

// yz[b][:] = y (zero padded); accum = 0; info = 0
__global__ void init_rhs_kernel(const double *__restrict__ y, int N, int npad, double *yz, double *accum,
                                int32_t *info, int32_t *sync) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < npad) yz[(size_t)b * npad + i] = i < N ? y[i] : 0.0;
    if (sync && b == 0 && blockIdx.x == 0 && threadIdx.x < 4) sync[threadIdx.x] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        accum[(size_t)b * 2] = 0.0;
        accum[(size_t)b * 2 + 1] = 0.0;
        info[b] = 0;
    }
}

// a leaf walk of this call met an invalid categorical value: every sample of the chunk reports it (info = -1)
__global__ void fault_info_kernel(const int32_t *fault, int32_t *info, int Bc) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < Bc && *fault) info[b] = -1;
}

// quick_inverse.py:38 / mcmc_record_mll.py:73
__global__ void finish_mll_kernel(const double *accum, int Bc, int N, int include_2pi, double *mll, const int32_t *fault,
                                  int32_t *info, const int32_t *sync) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= Bc) return;
    if (*fault) info[b] = -1;
    if (sync && sync[2] != 0) info[b] = -3;  // a device-side wait of this chunk timed out: its results are not valid
    double v = -accum[(size_t)b * 2] - accum[(size_t)b * 2 + 1];
    if (include_2pi) v = v - (double)N * log(2.0 * M_PI);
    mll[b] = 0.5 * v;
}

// mu[c] = sum_r V[r][c] z[r] ; var[c] = scale - sum_r V[r][c]^2      (V = U^-T K_Xx, candidate columns)
// mu[c] = sum_r V[r][c] z[r],  var[c] = scale - sum_r V[r][c]^2  (or the plain sum of squares without `scale`:
// identity right-hand side, diag(K_s^-1) = colsumsq(U^-T)) over the candidate block V of the factorised matrix.
// A workgroup owns 64 columns (one 512-byte row segment per wave load); its four waves take interleaved rows,
// eight in flight each, and are summed through LDS in wave order.  grid.z > 1 splits the rows into segments whose
// partial sums go to `part` ([segment][matrix][column][2]) for predict_finish_kernel — used when columns x
// matrices alone cannot fill the chip.
constexpr int PR_UNROLL = 8;
__global__ __launch_bounds__(256) void predict_reduce_kernel(Mats p, int N, int C, const double *scale, double *mu,
                                                             double *var, double *part) {
    __shared__ double red[2][4][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y, c = blockIdx.x * 64 + lane;
    const int nseg = gridDim.z, seg = blockIdx.z;
    const int rows_per = ((N + nseg - 1) / nseg + 3) & ~3;
    const int r_begin = seg * rows_per, r_end = min(N, r_begin + rows_per);
    const bool live = c < C;
    const double *V = p.A + (size_t)b * p.bstride + (size_t)p.nrb * NB + (live ? c : 0);
    const double *z = p.yz + (size_t)b * p.nrb * NB;
    double m = 0.0, s2 = 0.0;
    for (int r0 = r_begin + wave * PR_UNROLL; r0 < r_end; r0 += 4 * PR_UNROLL) {
        double v[PR_UNROLL];
#pragma unroll
        for (int u = 0; u < PR_UNROLL; ++u) v[u] = (live && r0 + u < r_end) ? V[(size_t)(r0 + u) * p.ld] : 0.0;
#pragma unroll
        for (int u = 0; u < PR_UNROLL; ++u) {
            m = fma(v[u], z[min(r0 + u, N - 1)], m);  // wave-uniform address
            s2 = fma(v[u], v[u], s2);
        }
    }
    red[0][wave][lane] = m;
    red[1][wave][lane] = s2;
    __syncthreads();
    if (threadIdx.x >= 64 || !live) return;
    m = ((red[0][0][lane] + red[0][1][lane]) + red[0][2][lane]) + red[0][3][lane];
    s2 = ((red[1][0][lane] + red[1][1][lane]) + red[1][2][lane]) + red[1][3][lane];
    if (nseg > 1) {
        double *dst = part + (((size_t)seg * gridDim.y + b) * C + c) * 2;
        dst[0] = m;
        dst[1] = s2;
        return;
    }
    mu[(size_t)b * C + c] = m;
    if (var) var[(size_t)b * C + c] = scale ? scale[b] - s2 : s2;
}

__global__ void predict_finish_kernel(const double *part, int nseg, int Bc, int C, const double *scale, double *mu,
                                      double *var) {
    const int b = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double m = 0.0, s2 = 0.0;
    for (int sg = 0; sg < nseg; ++sg) {  // fixed order
        const double *src = part + (((size_t)sg * Bc + b) * C + c) * 2;
        m += src[0];
        s2 += src[1];
    }
    mu[(size_t)b * C + c] = m;
    if (var) var[(size_t)b * C + c] = scale ? scale[b] - s2 : s2;
}

// `scratch` (the split-K slab area, idle after the sweep) may be null: then rows are never segmented
static int launch_predict_reduce(const Mats &p, int N, int C, int bc, const double *scale, double *mu, double *var,
                                 double *scratch, hipStream_t s) {
    const int col_groups = (C + 63) / 64;
    int nseg = 1;
    if (scratch && col_groups * bc < 512) {
        nseg = 1024 / (col_groups * bc);
        if (nseg > 16) nseg = 16;
        if (nseg > N / 256) nseg = N / 256;
        if (nseg < 1) nseg = 1;
    }
    hipLaunchKernelGGL(predict_reduce_kernel, dim3((unsigned)col_groups, (unsigned)bc, (unsigned)nseg), dim3(256), 0, s, p, N, C,
                       scale, mu, var, scratch);
    BARK_LAUNCH_CHECK();
    if (nseg > 1) {
        hipLaunchKernelGGL(predict_finish_kernel, dim3((unsigned)((C + 255) / 256), (unsigned)bc), dim3(256), 0, s, scratch, nseg,
                           bc, C, scale, mu, var);
        BARK_LAUNCH_CHECK();
    }
    return BARK_OK;
}

// y' K_inv y  (quick_inverse.py:38), one workgroup, grid-stride rows
__global__ __launch_bounds__(THREADS) void quadform_kernel(const double *__restrict__ Kinv,
                                                           const double *__restrict__ y, int N, double *out) {
    __shared__ double red[THREADS / 64];
    double total = 0.0;
    for (int r = blockIdx.x; r < N; r += gridDim.x) {
        double s = 0.0;
        for (int c = threadIdx.x; c < N; c += THREADS) s = fma(Kinv[(size_t)r * N + c], y[c], s);
        total = fma(s, y[r], total);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = total;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

}  // namespace
}  // namespace bark
