// chol_rows.h — part of the dense sweep's one translation unit (chol.hip includes it; kernels and helpers live in the
// anonymous namespace of that unit).  The row kernels: T = A - sum U'U per tile (row_kernel, with the Gram matrix generated in the epilogue and the diagonal tile
// as a SYRK), and its split-K form (panel_split_kernel + panel_reduce_kernel).
#pragma once
#include "chol_diag.h"

namespace bark {
namespace {

// ---------------------------------------------------------------------------------------------
// Tile helpers shared by the row kernels.
// ---------------------------------------------------------------------------------------------
// tile := A[rb,cb] - acc  (acc holds sum_{k<j} U[k,rb]'U[k,cb] in the MFMA D layout).
// GEN == 0: the A tile is read from HBM; GEN == 1 + LeafRep: it is generated from the leaf codes (bytes8 / bytes7 /
// bits): A[r][c] = [scale *] ((1/m) * #{t: leaf ids agree} [- shift])  (+ jitter on the global diagonal), identity in
// the padding — the Gram matrix is never written to or read from HBM.  Uses (and leaves dirty) the first
// 2 * nW * 128 dwords of LDS when GEN > 0; all 256 threads; the caller's GEMM ended with a barrier.
template <int GEN>
__device__ __forceinline__ void form_tile(const f64x4 (&acc)[4][4], const Mats &p, int b, int rb, int cb, double *tile,
                                          double *lds, int tid, const Lane &q) {
    if (GEN == 0) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                double *row = tile + (size_t)acc_row(q, mt, v) * p.ld;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int cc = acc_col(q, nt);
                    row[cc] = row[cc] - acc[mt][nt][v];
                }
            }
        return;
    }
    const int npad = p.nrb * NB;
    uint32_t *rows_l = reinterpret_cast<uint32_t *>(lds);  // [W][128] codes of this tile's rows
    uint32_t *cols_l = rows_l + p.nW * NB;                   // [W][128] codes of this tile's columns
    const uint32_t *lb = p.leafx + (size_t)b * p.nW * npad;
    for (int e = tid; e < p.nW * NB; e += THREADS) {
        const int w = e >> 7, r = e & (NB - 1);
        rows_l[e] = lb[(size_t)w * npad + rb * NB + r];
        cols_l[e] = lb[(size_t)w * npad + cb * NB + r];
    }
    __syncthreads();
    const double inv_m = 1.0 / (double)p.m;
    const bool has_scale = p.scale != nullptr, has_shift = p.shift != nullptr;
    const double sc = has_scale ? p.scale[b] : 1.0;
    const double sh = has_shift ? p.shift[b] : 0.0;
    const double jitter = 1e-6 + p.noise[b];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        uint32_t cnt[4][4] = {};
        for (int w = 0; w < p.nW; ++w) {
            uint32_t cw[4], rw[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) cw[nt] = cols_l[w * NB + acc_col(q, nt)];
#pragma unroll
            for (int v = 0; v < 4; ++v) rw[v] = rows_l[w * NB + acc_row(q, mt, v)];
#pragma unroll
            for (int v = 0; v < 4; ++v)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) cnt[v][nt] += code_count<(GEN > 0 ? GEN - 1 : 0)>(rw[v], cw[nt]);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int r = acc_row(q, mt, v), gi = rb * NB + r;
            double *row = tile + (size_t)r * p.ld;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int cc = acc_col(q, nt), gj = cb * NB + cc;
                double val;
                if (gi < p.N && gj < p.N) {
                    val = inv_m * (double)agree_count<(GEN > 0 ? GEN - 1 : 0)>(cnt[v][nt], p.m);
                    if (has_shift) val = val - sh;
                    if (has_scale) val = sc * val;
                    if (gi == gj) val = val + jitter;
                } else {
                    val = gi == gj ? 1.0 : 0.0;  // identity padding
                }
                row[cc] = val - acc[mt][nt][v];
            }
        }
    }
}

// y_i -= U[j,i]' z_j in a fixed summation order: per 16-row tile rt, p[rt][c] = sum of the lane's four rows (fma chain), then over the four lane groups
// (xor 16, xor 32); then y[c] -= ((p[0][c] + p[1][c]) + ...) + p[7][c].  `o` = the wave's U values of row tile rt,
// columns col0 + nt*16 + lr; z = z_j; part = LDS [8][128].
template <int NT>
__device__ __forceinline__ void y_partial(const f64x4 (&o)[NT], int rt, const double *__restrict__ z, double *part, int col0,
                                          const Lane &q) {
    double zr[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) zr[v] = z[rt * 16 + q.lk + 4 * v];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        double s = 0.0;
#pragma unroll
        for (int v = 0; v < 4; ++v) s = fma(o[nt][v], zr[v], s);
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (q.lk == 0) part[rt * NB + col0 + nt * 16 + q.lr] = s;
    }
}
__device__ __forceinline__ void y_commit(const double *part, double *yi, int tid) {  // after a barrier; tid < 128
    double s = part[tid];
#pragma unroll
    for (int rt = 1; rt < NSB_ROWS; ++rt) s += part[rt * NB + tid];
    yi[tid] = yi[tid] - s;
}

// ---------------------------------------------------------------------------------------------
// row_kernel: T[j,i] = A[j,i] - sum_{k<j} U[k,j]' U[k,i]  for every column block i > j, and the partial diagonal
// tile P[j+1,j+1] (same sum; diag_kernel(j+1) adds the k = j term).  1-D grid, (matrix, tile) from xcd_map.
// Fusing the triangular solve into this kernel's epilogue was built twice this round (T kept in registers; T read
// back through L2 by the same workgroup) and rejected on measurements: DESIGN.md, "Fused solve".
// ---------------------------------------------------------------------------------------------
// kdone: block rows of the K range this launch covers (j in the plain schedule; j - 1 in the pipelined one, where
// solve_kernel<1> / diag_kernel apply the rest).
// The partial DIAGONAL tile P[j+1,j+1] = A - sum_k U[k,j+1]'U[k,j+1] is symmetric and only its upper block triangle is ever
// read (diag_kernel / diag_pre_kernel take the 36 sub-blocks rb <= cb), so it is a SYRK, not a GEMM: 36 of the 64 16 x 16
// sub-block products, nine per wave (UpperBlocks, as diag_update), both MFMA operands from ONE LDS-DMA stage of the one panel
// (half the DMA of a square tile).  The diagonal tile is one of 32 - j tiles of a block row with the longest K each: 9.1 % of
// the row launches' tile x block-row products at N = 4096, of which this saves 7 / 16.  Per element the MFMA sequence is that
// of the square tile (k ascending, four k per MFMA): identical bits in the upper block triangle; the sub-blocks below it are
// not written.
template <int GEN, int W>
__device__ __forceinline__ void syrk_tile(const Mats &p, int b, int cbk, const double *__restrict__ panel, int K, double *tile,
                                          double *lds, int tid, int lane, int lr, int lk) {
    using T = UpperBlocks<W>;
    f64x4 acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = (f64x4){0.0, 0.0, 0.0, 0.0};
    // k-tiles of 32 rows (one operand: two of them fill the 72 KiB the square tile's two 16-row A + B stages take): 72 MFMAs per
    // wave between barriers.  With 16-row k-tiles (36 MFMAs, ~1 us) the next stage's DMA round trip, not the products, set the
    // pace: the tile took as long as a square one.
    constexpr int SK = 2 * BK;
    const int nk = K / SK;
    auto stage = [&](int kt, double *st) {  // wave W moves rows W, W+4, ..., W+28 of the k-tile
#pragma unroll
        for (int pp = 0; pp < SK / 4; ++pp) dma_row(panel + (size_t)(kt * SK + W + 4 * pp) * p.ld + lane * 2, st + (W + 4 * pp) * LDS_LD);
    };
    if (nk > 0) {
        stage(0, lds);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) stage(kt + 1, lds + ((kt + 1) & 1) * (SK * LDS_LD));
            const double *st = lds + (kt & 1) * (SK * LDS_LD);
#pragma unroll
            for (int kk = 0; kk < SK / 4; ++kk) {
                double fr[8];
#pragma unroll
                for (int blk = 0; blk < 8; ++blk)
                    if (blk >= W) fr[blk] = st[(kk * 4 + lk) * LDS_LD + blk * 16 + lr];
#pragma unroll
                for (int i = 0; i < 9; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[T::rb[i]], fr[T::cb[i]], acc[i], 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    if (GEN == 0) {
#pragma unroll
        for (int i = 0; i < 9; ++i)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                double *e = tile + (size_t)(T::rb[i] * 16 + lk + 4 * v) * p.ld + T::cb[i] * 16 + lr;
                *e = *e - acc[i][v];
            }
        return;
    }
    // A generated from the leaf codes, form_tile's arithmetic operation for operation (rows and columns are the same points)
    const int npad = p.nrb * NB;
    uint32_t *codes = reinterpret_cast<uint32_t *>(lds);  // [nW][128]
    const uint32_t *lb = p.leafx + (size_t)b * p.nW * npad;
    for (int e = tid; e < p.nW * NB; e += THREADS) codes[e] = lb[(size_t)(e >> 7) * npad + cbk * NB + (e & (NB - 1))];
    __syncthreads();
    const double inv_m = 1.0 / (double)p.m;
    const bool has_scale = p.scale != nullptr, has_shift = p.shift != nullptr;
    const double sc = has_scale ? p.scale[b] : 1.0, sh = has_shift ? p.shift[b] : 0.0, jitter = 1e-6 + p.noise[b];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        uint32_t cnt[4] = {0, 0, 0, 0};
        for (int w = 0; w < p.nW; ++w) {
            const uint32_t cw = codes[w * NB + T::cb[i] * 16 + lr];
#pragma unroll
            for (int v = 0; v < 4; ++v) cnt[v] += code_count<(GEN > 0 ? GEN - 1 : 0)>(codes[w * NB + T::rb[i] * 16 + lk + 4 * v], cw);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int r = T::rb[i] * 16 + lk + 4 * v, cc = T::cb[i] * 16 + lr, gi = cbk * NB + r, gj = cbk * NB + cc;
            double val;
            if (gi < p.N && gj < p.N) {
                val = inv_m * (double)agree_count<(GEN > 0 ? GEN - 1 : 0)>(cnt[v], p.m);
                if (has_shift) val = val - sh;
                if (has_scale) val = sc * val;
                if (gi == gj) val = val + jitter;
            } else {
                val = gi == gj ? 1.0 : 0.0;  // identity padding
            }
            tile[(size_t)r * p.ld + cc] = val - acc[i][v];
        }
    }
}

template <int GEN>  // 0: A tile read from HBM; 1 + LeafRep: A generated from the leaf codes (bytes8 / bytes7 / bits)
__global__ __launch_bounds__(THREADS, 2) void row_kernel(Mats p, int j, int kdone, int n_right, int n_tiles, int syrk) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x;
    int b, t;
    int rb, cb;
    if (syrk == 4) {
        // PAIR launch (plain schedule, lock-step chunks: Sweep::step_paired): block rows j and j+1 over the SAME K range (k < kdone),
        // n_tiles square tiles per matrix — (j, j+1), then (j, c), (j+1, c) next to each other for every column block c >= j+2, so
        // that the two tiles that stream the B panel of column block c run side by side on one XCD and the second finds it in L2
        // — and after all of them the SYRK workgroups of the partial diagonal tiles (j+1, j+1) and, if there is one, (j+2, j+2).
        const int first_diag = (int)xcd_grid(n_tiles, p.Bc), per = NXCD * ((p.Bc + NXCD - 1) / NXCD);
        if ((int)blockIdx.x >= first_diag) {
            const int local = (int)blockIdx.x - first_diag, which = local / per;
            b = local - which * per;
            if (b >= p.Bc) return;
            rb = cb = j + 1 + which;
        } else {
            if (!xcd_map(blockIdx.x, n_tiles, p.Bc, b, t)) return;
            if (t == 0) {
                rb = j;
                cb = j + 1;
            } else {
                rb = j + ((t - 1) & 1);
                cb = j + 2 + ((t - 1) >> 1);
            }
        }
    } else {
        if (syrk == 2) {  // the square tiles first (XCD-aware map over n_right tiles), then one SYRK workgroup per matrix: launch_rows
            const int first_diag = (int)xcd_grid(n_right, p.Bc);  // a multiple of 8: workgroup id % 8 == b % 8 in the tail as well
            if ((int)blockIdx.x >= first_diag) {
                b = (int)blockIdx.x - first_diag;
                t = n_right;
                if (b >= p.Bc) return;
            } else if (!xcd_map(blockIdx.x, n_right, p.Bc, b, t)) {
                return;
            }
        } else if (!xcd_map(blockIdx.x, n_tiles, p.Bc, b, t)) {
            return;
        }
        rb = t < n_right ? j : j + 1;
        cb = t < n_right ? j + 1 + t : j + 1;
    }
    const Lane q = lane_of(tid);
    double *Ab = p.A + (size_t)b * p.bstride;
    if (syrk && rb == cb) {  // workgroup-uniform: the partial diagonal tile
        const int wsel = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
        double *tile = Ab + (size_t)rb * NB * p.ld + (size_t)cb * NB;
        if (wsel == 0)
            syrk_tile<GEN, 0>(p, b, cb, Ab + (size_t)cb * NB, kdone * NB, tile, lds, tid, lane, q.lr, q.lk);
        else if (wsel == 1)
            syrk_tile<GEN, 1>(p, b, cb, Ab + (size_t)cb * NB, kdone * NB, tile, lds, tid, lane, q.lr, q.lk);
        else if (wsel == 2)
            syrk_tile<GEN, 2>(p, b, cb, Ab + (size_t)cb * NB, kdone * NB, tile, lds, tid, lane, q.lr, q.lk);
        else
            syrk_tile<GEN, 3>(p, b, cb, Ab + (size_t)cb * NB, kdone * NB, tile, lds, tid, lane, q.lr, q.lk);
        return;
    }
    f64x4 acc[4][4];
    zero_acc(acc);
    gemm_kmajor_dma(acc, Ab + (size_t)rb * NB, p.ld, Ab + (size_t)cb * NB, p.ld, kdone * NB, lds, tid, q);
    form_tile<GEN>(acc, p, b, rb, cb, Ab + (size_t)rb * NB * p.ld + (size_t)cb * NB, lds, tid, q);
}

// ---------------------------------------------------------------------------------------------
// Split-K variant of the panel update for under-filled steps (few matrices x few tiles, e.g. one
// N = 16384 matrix): the block rows [kb_lo, kb_hi) of every tile's K range are cut into S contiguous slabs, each
// accumulated by its own workgroup into slot s_off + s of the tile's S_tot scratch slabs; panel_reduce_kernel then
// forms T = A - sum over the S_tot slabs in a fixed order (deterministic, unlike atomics).  Tile index t' = tile * S + s.
// One launch over [0, j) is the plain split; the look-ahead schedule (Sweep::step) makes two: the bulk [0, j-1) one
// step early and the last block row [j-1, j) on the critical path.
// ---------------------------------------------------------------------------------------------
// (Letting the last-block-row launch of a look-ahead step also add the bulk slabs and store T — no reduce launch on the
// critical path — was measured much slower: one workgroup per tile streams its S slabs at a fraction of the rate the
// 16 reduce workgroups per tile reach; one N = 16384 matrix 26.9 -> 36.7 ms.)
__global__ __launch_bounds__(THREADS, 2) void panel_split_kernel(Mats p, int j, int n_right, int t_off, int n_tiles, int kb_lo,
                                                                  int kb_hi, int S, int s_off, int S_tot, double *slabs) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x;
    int b, ts;
    if (!xcd_map(blockIdx.x, n_tiles * S, p.Bc, b, ts)) return;  // n_tiles tiles starting at tile t_off of the block row
    const Lane q = lane_of(tid);
    const int tl = ts / S, s = ts - tl * S, t = t_off + tl;
    const int rb = t < n_right ? j : j + 1, cb = t < n_right ? j + 1 + t : j + 1;
    // slab s = k-tiles (16 rows) [k0, k1) of the range: cut at k-tile, not block-row, granularity, so the S workgroups of
    // a tile differ by one k-tile at most (with 7.45 block rows per slab the 8-row slabs set the pace: 7 % idle)
    const long nkt = (long)(kb_hi - kb_lo) * (NB / BK);
    const int k0 = kb_lo * (NB / BK) + (int)((nkt * s) / S), k1 = kb_lo * (NB / BK) + (int)((nkt * (s + 1)) / S);
    const double *Ab = p.A + (size_t)b * p.bstride + (size_t)k0 * BK * p.ld;
    f64x4 acc[4][4];
    zero_acc(acc);
    gemm_kmajor_dma(acc, Ab + (size_t)rb * NB, p.ld, Ab + (size_t)cb * NB, p.ld, (k1 - k0) * BK, lds, tid, q);
    double *slab = slabs + ((size_t)((size_t)b * n_tiles + tl) * S_tot + s_off + s) * NB * NB;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            double *row = slab + (size_t)acc_row(q, mt, v) * NB;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) row[acc_col(q, nt)] = acc[mt][nt][v];
        }
}

// (Round 5: 16 / 32 rows per workgroup — half / a quarter of the workgroups, two / four independent groups per thread — measured
// slower, 0.3-1.7 % / 1-10 %: N = 4096 x 16 6.92 | 6.98 | 7.01 ms, N = 16384 x 1 25.16 | 25.57 | 26.20, N = 4096 x 1 1.66 | 1.70 | 1.85.  Beside
// resident row workgroups the reduce launch lives in the third wave slot of a SIMD; many short workgroups fill it best.)
#ifndef BARK_RED_ROWS
#define BARK_RED_ROWS 8
#endif
constexpr int RED_ROWS = BARK_RED_ROWS;  // rows of a 128 x 128 tile per panel_reduce_kernel workgroup: 4 doubles per thread and group of 8 rows
static_assert(RED_ROWS % 8 == 0 && NB % RED_ROWS == 0, "whole groups of 8 rows");
// T = A - sum of the tile's S slabs (fixed order: bit-reproducible) for the n_tiles tiles from t_off on.
// GEN == 0: A is read from (and T written to) the materialised matrix; GEN == 1 + LeafRep: A is generated from the
// leaf codes exactly as form_tile does (MLL-only sweeps never materialise the Gram matrix).
template <int GEN>
__global__ __launch_bounds__(THREADS) void panel_reduce_kernel(Mats p, int j, int n_right, int t_off, int n_tiles, int S,
                                                               const double *slabs) {
    const int tl = blockIdx.x / (NB / RED_ROWS), part = blockIdx.x % (NB / RED_ROWS), b = blockIdx.y;
    const int t = t_off + tl;
    const int rb = t < n_right ? j : j + 1, cb = t < n_right ? j + 1 + t : j + 1;
    const double inv_m = 1.0 / (double)p.m;
    const bool has_scale = GEN != 0 && p.scale != nullptr, has_shift = GEN != 0 && p.shift != nullptr;
    const double sc = has_scale ? p.scale[b] : 1.0, sh = has_shift ? p.shift[b] : 0.0, jitter = GEN != 0 ? 1e-6 + p.noise[b] : 0.0;
#pragma unroll
    for (int q = 0; q < RED_ROWS / 8; ++q) {  // (independent groups: their loads overlap)
        const int e = (part * RED_ROWS + q * 8) * NB + 4 * threadIdx.x;  // four consecutive entries of one tile row
        const double *slab = slabs + (size_t)((size_t)b * n_tiles + tl) * S * NB * NB + e;
        f64x2 s0 = {0.0, 0.0}, s1 = {0.0, 0.0};
#pragma unroll 4
        for (int s = 0; s < S; ++s) {
            const f64x2 *src = reinterpret_cast<const f64x2 *>(slab + (size_t)s * NB * NB);
            s0 += src[0];
            s1 += src[1];
        }
        const int r = e >> 7, c0 = e & (NB - 1);
        f64x2 *dst = reinterpret_cast<f64x2 *>(p.A + (size_t)b * p.bstride + ((size_t)rb * NB + r) * p.ld + (size_t)cb * NB + c0);
        if (GEN == 0) {
            dst[0] -= s0;
            dst[1] -= s1;
            continue;
        }
        const int npad = p.nrb * NB, gi = rb * NB + r, gj0 = cb * NB + c0;
        const uint32_t *lb = p.leafx + (size_t)b * p.nW * npad;
        uint32_t cnt[4] = {0, 0, 0, 0};
        for (int w = 0; w < p.nW; ++w) {
            const uint32_t rw = lb[(size_t)w * npad + gi];
            const uint4 cw = *reinterpret_cast<const uint4 *>(lb + (size_t)w * npad + gj0);
            cnt[0] += code_count<(GEN > 0 ? GEN - 1 : 0)>(rw, cw.x);
            cnt[1] += code_count<(GEN > 0 ? GEN - 1 : 0)>(rw, cw.y);
            cnt[2] += code_count<(GEN > 0 ? GEN - 1 : 0)>(rw, cw.z);
            cnt[3] += code_count<(GEN > 0 ? GEN - 1 : 0)>(rw, cw.w);
        }
        double val[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gj = gj0 + i;
            double v;
            if (gi < p.N && gj < p.N) {
                v = inv_m * (double)agree_count<(GEN > 0 ? GEN - 1 : 0)>(cnt[i], p.m);
                if (has_shift) v = v - sh;
                if (has_scale) v = sc * v;
                if (gi == gj) v = v + jitter;
            } else {
                v = gi == gj ? 1.0 : 0.0;  // identity padding
            }
            val[i] = v;
        }
        dst[0] = (f64x2){val[0], val[1]} - s0;
        dst[1] = (f64x2){val[2], val[3]} - s1;
    }
}

}  // namespace
}  // namespace bark
