// chol_diag.h — part of the dense sweep's one translation unit (chol.hip includes it; kernels and helpers live in the
// anonymous namespace of that unit).  The diagonal-block kernels: 16 x 16 elimination through the matrix pipe (factor16), the blocked factor + inverse of a
// 128 x 128 tile (factor_tile / factor_tile8), diag_kernel, the one-launch kernels of N <= 256 (diag_kernel<true>, two_block_kernel),
// diag_pre_kernel, and the one-lane kernels of the device-side hand-over.
#pragma once
#include <type_traits>

#include "chol_tiles.h"

namespace bark {
namespace {

// ---------------------------------------------------------------------------------------------
// diag_kernel: factor + invert the j-th diagonal block (128x128) of every matrix of the chunk.
//
// Blocked in 16x16 sub-blocks held in one LDS image S[128][SD] (upper block triangle used):
//   for kb = 0..7:   (A) wave 0 eliminates the 16x16 diagonal sub-block in registers (one element
//                        column per lane, rows broadcast by ds_bpermute shuffles, no barriers) on
//                        the augmented [D | I], giving W_kk = U_kk^-1 directly and the pivots;
//                    (B) U[kb,cb] = W_kk' D[kb,cb]            (one 16x16x16 MFMA chain per block)
//                    (C) D[rb,cb] -= U[kb,rb]' U[kb,cb]       (one chain per trailing block)
//   then the block inverse X = U^-1 in place, column block by column block:
//                        X[rb,jb] = -(sum_{rb<=k<jb} X[rb,k] U[k,jb]) W_jj
// 3 barriers per kb + 2 per jb instead of ~5 per scalar column.
// ---------------------------------------------------------------------------------------------
constexpr int SB = 16;          // sub-block edge
constexpr int NSB = NB / SB;    // 8 sub-blocks per edge
constexpr int NBLK = NSB * (NSB + 1) / 2;  // 36 stored sub-blocks (upper block triangle)
constexpr int NSB_ROWS = NSB;             // 16-row tiles per block (y-update partials)

// S is stored as a packed upper block triangle: sub-block (rb, cb), rb <= cb, is a contiguous
// row-major 16x16 (2 KiB), so the whole 128x128 factor image takes 72 KiB instead of 136 KiB and the
// kernel can share a CU with a row workgroup (it runs beside row_kernel, which is on a helper stream).
// A k-major MFMA operand read (4 rows x 16 columns) is one contiguous 512-B span: conflict-free.
__device__ __forceinline__ int blk_off(int rb, int cb) { return (rb * NSB - (rb * (rb - 1)) / 2 + (cb - rb)) * SB * SB; }
__device__ __forceinline__ double &s_at(double *S, int r, int c) {  // element (r, c), r/16 <= c/16
    return S[blk_off(r >> 4, c >> 4) + (r & 15) * SB + (c & 15)];
}

// acc(16x16) += X' Y for two sub-blocks stored "k-major" (X[k][i], Y[k][j]) with row strides ldx, ldy
__device__ __forceinline__ void mfma_tn(f64x4 &acc, const double *X, int ldx, const double *Y, int ldy, int lr, int lk) {
#pragma unroll
    for (int kk = 0; kk < SB / 4; ++kk)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[(kk * 4 + lk) * ldx + lr], Y[(kk * 4 + lk) * ldy + lr], acc, 0, 0, 0);
}
// acc(16x16) += X Y with X stored row-major (X[i][k]) and Y k-major (Y[k][j])
__device__ __forceinline__ void mfma_nn(f64x4 &acc, const double *X, int ldx, const double *Y, int ldy, int lr, int lk) {
#pragma unroll
    for (int kk = 0; kk < SB / 4; ++kk)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[lr * ldx + kk * 4 + lk], Y[(kk * 4 + lk) * ldy + lr], acc, 0, 0, 0);
}

// (A): one wave eliminates the 16x16 diagonal sub-block `blk` (row stride SB) in registers and overwrites
// it with W = U_kk^-1 (upper triangular, row-major).  Lane (g = l>>4, c = l&15) owns rows g, g+4, g+8,
// g+12 of column c of [D | I].  Returns sum log(pivot) and the first bad pivot.
// value of `v` in lane `src` (a wave-uniform, here compile-time, lane index): two v_readlane_b32 instead of the
// LDS-crossbar ds_bpermute a general __shfl costs — this sits on the serial pivot chain of factor16
__device__ __forceinline__ double readlane_f64(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// 1 / d for a positive, normal d: v_rcp_f64 plus two Newton steps (full double accuracy, not correctly rounded;
// the IEEE division sequence is ~3x longer and also on the pivot chain)
__device__ __forceinline__ double recip_pos(double d) {
    double x = __builtin_amdgcn_rcp(d);
    x = fma(fma(-d, x, 1.0), x, x);
    x = fma(fma(-d, x, 1.0), x, x);
    return x;
}

// lane K of every 16-lane row broadcast to that row (DPP row_newbcast:K): VALU speed, no LDS round trip
template <int K>
__device__ __forceinline__ double row_bcast_f64(double v) {
    constexpr int ctrl = 0x150 + K;
    // every lane is written (row_mask = bank_mask = 0xF): mov_dpp leaves the old value undefined, update_dpp(0, ...) costs a
    // v_mov of the zero per half in front of every broadcast (20 of a block4's ~120 instructions)
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), ctrl, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), ctrl, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

// Four elimination steps of factor16 at once (pivots K0 = 4Q .. K0+3) on [D | I]; lane (g, c) holds rows g + 4v of column c in
// e[v] / f[v] — which is the D layout of the 16x16x4 MFMA, so the cross-lane work of the block goes through the matrix pipe.
// (The wave is bound by the ISSUE of its instructions — ~8 cycles per fp64 VALU operation, 4 per 32-bit one — not by the
// pivot-to-pivot latency: round 3's form, with lane swaps, a per-row multiplier recurrence and two FMA chains per row
// register, was ~930 instructions and 6.0 K cycles per 16 pivots.)
//   * the four pivot rows (register Q of the four lane groups) reach every lane group by ONE product with a 0/1 selector
//     (exact), instead of six lane-swap instructions and as many copies per double;
//   * the 4 x 4 diagonal block is eliminated inside those rows: the pivot and the multipliers of step k are row k's entries
//     at the block's columns (DPP row broadcasts); the four reciprocals stay on the dependent chain;
//   * the rank-4 update of ALL rows below — D -= M A, I -= M S with M[i][k] = A_k[i] / d_k, the pivot rows' own entries at
//     column i: the trailing block is symmetric and only its upper triangle is ever read — is one MFMA each.  M is zeroed
//     for the rows of this block and the finished ones.
// Inside a block the pivots are formed by the single-pivot order's operations; the rows below it get multipliers taken from the
// pivot ROW instead of the pivot column and the MFMA's own summation, so the factor agrees with round 3's to rounding (the parity
// tests' rtol 1e-9 against the reference's arithmetic is met with the same margin, ~1e-14 relative at N = 4096), not bit for bit.
// Out: piv[Q][k] the pivots (1.0 where one was not positive: flagged in badbits, the sweep continues finite), dsel[Q] the
// pivot of this lane's row g + 4Q.
template <int Q>
__device__ __forceinline__ void block4(f64x4 &e, f64x4 &f, double (&piv)[4][4], double (&dsel)[4], int &badbits, int c, int g,
                                       double gsel) {
    constexpr int K0 = 4 * Q;
    const f64x4 zero = {0.0, 0.0, 0.0, 0.0};
    const f64x4 Ar = __builtin_amdgcn_mfma_f64_16x16x4f64(gsel, e[Q], zero, 0, 0, 0);  // Ar[k] = D[K0 + k][c]
    double S0, S1, S2, S3;
    if constexpr (Q == 0) {  // still the identity
        S0 = c == 0 ? 1.0 : 0.0;
        S1 = c == 1 ? 1.0 : 0.0;
        S2 = c == 2 ? 1.0 : 0.0;
        S3 = c == 3 ? 1.0 : 0.0;
    } else {
        const f64x4 Sr = __builtin_amdgcn_mfma_f64_16x16x4f64(gsel, f[Q], zero, 0, 0, 0);  // I[K0 + k][c]
        S0 = Sr[0];
        S1 = Sr[1];
        S2 = Sr[2];
        S3 = Sr[3];
    }
    double A0 = Ar[0], A1 = Ar[1], A2 = Ar[2], A3 = Ar[3];
    auto pivot = [&](double d, int k) {  // d is wave-uniform
        const bool ok = d > 0.0;  // false for NaN too
        badbits |= ok ? 0 : 1 << (K0 + k);
        return ok ? d : 1.0;
    };
    const double p0 = pivot(row_bcast_f64<K0 + 0>(A0), 0);
    const double rd0 = recip_pos(p0);
    const double l10 = row_bcast_f64<K0 + 1>(A0) * rd0, l20 = row_bcast_f64<K0 + 2>(A0) * rd0, l30 = row_bcast_f64<K0 + 3>(A0) * rd0;
    A1 = fma(-l10, A0, A1);
    const double p1 = pivot(row_bcast_f64<K0 + 1>(A1), 1);
    const double rd1 = recip_pos(p1);
    const double l21 = row_bcast_f64<K0 + 2>(A1) * rd1, l31 = row_bcast_f64<K0 + 3>(A1) * rd1;
    A2 = fma(-l21, A1, fma(-l20, A0, A2));
    const double p2 = pivot(row_bcast_f64<K0 + 2>(A2), 2);
    const double rd2 = recip_pos(p2);
    const double l32 = row_bcast_f64<K0 + 3>(A2) * rd2;
    A3 = fma(-l32, A2, fma(-l31, A1, fma(-l30, A0, A3)));
    const double p3 = pivot(row_bcast_f64<K0 + 3>(A3), 3);
    piv[Q][0] = p0;
    piv[Q][1] = p1;
    piv[Q][2] = p2;
    piv[Q][3] = p3;
    S1 = fma(-l10, S0, S1);
    S2 = fma(-l21, S1, fma(-l20, S0, S2));
    S3 = fma(-l32, S2, fma(-l31, S1, fma(-l30, S0, S3)));
    // lane group k owns row K0 + k of the block
    const double X = g == 0 ? A0 : g == 1 ? A1 : g == 2 ? A2 : A3;
    const double Sx = g == 0 ? S0 : g == 1 ? S1 : g == 2 ? S2 : S3;
    dsel[Q] = g == 0 ? p0 : g == 1 ? p1 : g == 2 ? p2 : p3;
    if constexpr (Q + 1 < 4) {
        const double rd3 = recip_pos(p3);
        const double R = g == 0 ? rd0 : g == 1 ? rd1 : g == 2 ? rd2 : rd3;
        // A operand of lane (i = c, k = g): -M[i][k]
        const double mneg = c >= K0 + 4 ? -(X * R) : 0.0;
        e = __builtin_amdgcn_mfma_f64_16x16x4f64(mneg, X, e, 0, 0, 0);
        f = __builtin_amdgcn_mfma_f64_16x16x4f64(mneg, Sx, f, 0, 0, 0);
    }
    e[Q] = X;
    f[Q] = Sx;
    if constexpr (Q + 1 < 4) block4<Q + 1>(e, f, piv, dsel, badbits, c, g, gsel);
}

// One wave eliminates the 16x16 diagonal sub-block `blk` and overwrites it with W = U_kk^-1; `pacc` collects the pivots
// for log|D|: lane l keeps the product of the four pivots of block (l & 3) of the sub-block number ((l >> 2) & 7), and
// pivots_logsum turns the lot into the sum of logs ONCE per tile (the double-precision log is ~100 fp64 instructions,
// ~800 cycles: per sub-block it was an eighth of the chain).
// (e: the sub-block in the MFMA D layout — lane (g, c): rows g + 4v of column c; factor16 below loads it from `blk`)
__device__ __forceinline__ void factor16_reg(double *blk, f64x4 e, int lane, int base_index, double &pacc, int &bad) {
    const int c = lane & 15, g = lane >> 4;
    f64x4 f;
#pragma unroll
    for (int v = 0; v < 4; ++v) f[v] = (g + 4 * v == c) ? 1.0 : 0.0;
    double piv[4][4], dsel[4];
    int badbits = 0;
    block4<0>(e, f, piv, dsel, badbits, c, g, g == (c >> 2) ? 1.0 : 0.0);
    const int first = __builtin_ffs(badbits);  // 1 + index of the first pivot that was not positive; 0: none
    bad = (bad == 0 && first != 0) ? base_index + first : bad;
    {
        const int sel = lane & 3;
        const double p0 = (piv[0][0] * piv[0][1]) * (piv[0][2] * piv[0][3]), p1 = (piv[1][0] * piv[1][1]) * (piv[1][2] * piv[1][3]),
                     p2 = (piv[2][0] * piv[2][1]) * (piv[2][2] * piv[2][3]), p3 = (piv[3][0] * piv[3][1]) * (piv[3][2] * piv[3][3]);
        const double ps = sel == 0 ? p0 : sel == 1 ? p1 : sel == 2 ? p2 : p3;
        pacc = ((lane >> 2) & 7) == (base_index >> 4) ? ps : pacc;
    }
    // row r of the right half is (L~^-1)[r][:]; U^-T = diag(1/sqrt d) L~^-1, so W[c][r] = f * rsqrt(d_r), r = g + 4v.
    // 1/sqrt by v_rsq_f64 + two Newton steps (full double accuracy; IEEE sqrt + divide is ~4x longer)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const double dr = dsel[v];
        double rs = __builtin_amdgcn_rsq(dr);
        rs = rs * fma(-0.5 * dr * rs, rs, 1.5);
        rs = rs * fma(-0.5 * dr * rs, rs, 1.5);
        blk[c * SB + (g + 4 * v)] = f[v] * rs;
    }
}
__device__ __forceinline__ void factor16(double *blk, int lane, int base_index, double &pacc, int &bad) {
    const int c = lane & 15, g = lane >> 4;
    f64x4 e;
#pragma unroll
    for (int v = 0; v < 4; ++v) e[v] = blk[(g + 4 * v) * SB + c];
    factor16_reg(blk, e, lane, base_index, pacc, bad);
}

// sum over the tile's sub-blocks of 0.5 log(product of the 16 pivots), from factor16's per-lane products (pacc starts at 1.0;
// every lane of the wave): one log per lane, the four blocks of a sub-block meet by two quad-permute DPP adds, the eight
// sub-blocks are added in order.
__device__ __forceinline__ double pivots_logsum(double pacc) {
    double lg = log(pacc);
    lg += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(lg), 0xB1, 0xF, 0xF, false),   // quad_perm [1,0,3,2]
                           __builtin_amdgcn_update_dpp(0, __double2loint(lg), 0xB1, 0xF, 0xF, false));
    lg += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(lg), 0x4E, 0xF, 0xF, false),   // quad_perm [2,3,0,1]
                           __builtin_amdgcn_update_dpp(0, __double2loint(lg), 0x4E, 0xF, 0xF, false));
    double s = 0.0;
#pragma unroll
    for (int kb = 0; kb < NSB; ++kb) s += 0.5 * readlane_f64(lg, 4 * kb);
    return s;
}

// Rank-128 update of the diagonal tile, upper block triangle only: D = P - U[j-1,j]' U[j-1,j] for the 36 sub-blocks
// (rb <= cb), nine per wave (wave W takes block row W from the diagonal to the right edge plus the short rows at the
// bottom: 8+1, 7+2, 6+3, 5+4), written straight into the packed factor image S.  Both MFMA operands come from the SAME
// k-major panel, so one LDS-DMA stage of 16 x 128 doubles serves A and B fragments; the P values are requested before
// the product loop.  A full 128 x 128 product with 64 x 64 wave tiles would leave one wave computing only discarded
// sub-blocks (41 K cycles against 22 K here).  Per element the MFMA sequence is k-ascending, as everywhere.
template <int W>
struct UpperBlocks;
template <>
struct UpperBlocks<0> {
    static constexpr int rb[9] = {0, 0, 0, 0, 0, 0, 0, 0, 7}, cb[9] = {0, 1, 2, 3, 4, 5, 6, 7, 7};
};
template <>
struct UpperBlocks<1> {
    static constexpr int rb[9] = {1, 1, 1, 1, 1, 1, 1, 6, 6}, cb[9] = {1, 2, 3, 4, 5, 6, 7, 6, 7};
};
template <>
struct UpperBlocks<2> {
    static constexpr int rb[9] = {2, 2, 2, 2, 2, 2, 5, 5, 5}, cb[9] = {2, 3, 4, 5, 6, 7, 5, 6, 7};
};
template <>
struct UpperBlocks<3> {
    static constexpr int rb[9] = {3, 3, 3, 3, 3, 4, 4, 4, 4}, cb[9] = {3, 4, 5, 6, 7, 4, 5, 6, 7};
};

constexpr int UPD_STAGE = BK * LDS_LD;  // one operand: 16 rows x (128 + 16) doubles

// Matrices of ONE block row (N <= 128: where BARK itself lives — BO with tens of points, BASELINE configs[0] is N = 64):
// diag_kernel does the whole evaluation in one launch.  It generates its tile from the leaf codes (as form_tile does),
// takes y straight from the caller, and writes the MLL (finish_mll_kernel's arithmetic) — no Gram fill, no right-hand-side
// initialisation, no finishing launch (5 launches -> walk + this one).
struct OneBlock {
    const double *y;       // (N,) targets, or nullptr: the regular multi-block sweep
    double *mll;           // (Bc,) result
    const int32_t *fault;  // the context's categorical-fault flag (set by the leaf walk that precedes this launch)
    int include_2pi, rep;  // MLL convention; leaf-code encoding (LeafRep)
    // the leaf walk INSIDE the kernel (diag_kernel<true, 8>, round 5): nodes != nullptr — the chunk's packed forests, the points
    // and the walk's bounds; the codes then never touch global memory and the walk launch in front of the kernel is gone
    const uint4 *nodes;
    const double *X;
    int32_t *fault_w;
    int stride, m, max_depth, d;
};

// What every generated entry of matrix b needs besides the two points' codes — read ONCE per kernel phase (gen_ctx): as a
// per-entry read of p.scale[b] / p.shift[b] / p.noise[b] and a per-entry 1.0 / m the generation of a 128 x 128 tile's upper
// block triangle took 57 K cycles (24 us) of the one-launch kernels, most of it global-load latency (round 5,
// profiles/r05/small_n.txt).
struct GenCtx {
    double inv_m, sc, sh, jitter;
    int has_scale, has_shift, rep, nW, N, m;
};
__device__ __forceinline__ GenCtx gen_ctx(const Mats &p, int b, int rep) {
    GenCtx g;
    g.inv_m = 1.0 / (double)p.m;
    g.has_scale = p.scale != nullptr;
    g.has_shift = p.shift != nullptr;
    g.sc = g.has_scale ? p.scale[b] : 1.0;
    g.sh = g.has_shift ? p.shift[b] : 0.0;
    g.jitter = 1e-6 + p.noise[b];
    g.rep = rep;
    g.nW = p.nW;
    g.N = p.N;
    g.m = p.m;
    return g;
}
// A[gi[v]][gj], v = 0..3, of the matrix from the leaf codes staged in LDS (codes[w][cs]: cs points per code plane — 128 for one
// block row, 256 for two; planes are zero beyond the last point): form_tile's arithmetic, operation for operation.  The code
// words go round the OUTSIDE — one pass over the planes serves the four rows, five LDS reads per word with four independent
// counts — where an entry at a time was a chain of dependent LDS round trips per entry (36 of them per lane and tile).
template <int REP>
__device__ __forceinline__ void gen_counts4(const uint32_t *codes, int cs, int nW, const int (&gi)[4], int gj, uint32_t (&cnt)[4]) {
    for (int w = 0; w < nW; ++w) {
        const uint32_t *pl = codes + w * cs;
        const uint32_t cw = pl[gj];
#pragma unroll
        for (int v = 0; v < 4; ++v) cnt[v] += code_count<REP>(pl[gi[v]], cw);
    }
}
__device__ __forceinline__ void gen_rows4(const GenCtx &g, const uint32_t *codes, int cs, const int (&gi)[4], int gj, double (&out)[4]) {
    uint32_t cnt[4] = {0, 0, 0, 0};
    if (g.rep == REP_BITS)
        gen_counts4<REP_BITS>(codes, cs, g.nW, gi, gj, cnt);
    else if (g.rep == REP_BYTES7)
        gen_counts4<REP_BYTES7>(codes, cs, g.nW, gi, gj, cnt);
    else
        gen_counts4<REP_BYTES8>(codes, cs, g.nW, gi, gj, cnt);
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        // (selects, not branches: as if / else every value cost an exec-mask save and restore around a handful of operations)
        const int agree = g.rep == REP_BITS ? (int)cnt[v] : g.m - (int)cnt[v];
        double val = g.inv_m * (double)agree;
        const double shifted = val - g.sh;
        val = g.has_shift ? shifted : val;
        const double scaled = g.sc * val;
        val = g.has_scale ? scaled : val;
        const double jittered = val + g.jitter;
        const bool diag = gi[v] == gj;
        val = diag ? jittered : val;
        const bool live = gi[v] < g.N && gj < g.N;
        out[v] = live ? val : (diag ? 1.0 : 0.0);  // identity padding
    }
}

// codes != nullptr (one-block-row sweeps): the tile is generated from the leaf codes in LDS instead of read from `tile`.
// goff: index of the tile's first point (ONE: 0; the second block of TWO: 128), cs: points per code plane.
template <int W, bool ONE>
__device__ __forceinline__ void diag_update(const double *__restrict__ tile, long ld, const double *__restrict__ panel0,
                                            int nkb, double *lds, double *S, int lane, int lr, int lk, const Mats &p, int b,
                                            int rep, const uint32_t *codes, int cs = NB, int goff = 0) {
    using T = UpperBlocks<W>;
    double pre[9][4];
    GenCtx g = {};
    if (ONE) g = gen_ctx(p, b, rep);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        if (ONE) {
            const int r0 = goff + T::rb[i] * 16 + lk;
            const int gi[4] = {r0, r0 + 4, r0 + 8, r0 + 12};
            gen_rows4(g, codes, cs, gi, goff + T::cb[i] * 16 + lr, pre[i]);
        } else {
#pragma unroll
            for (int v = 0; v < 4; ++v) pre[i][v] = tile[(size_t)(T::rb[i] * 16 + lk + 4 * v) * ld + T::cb[i] * 16 + lr];
        }
    }
    f64x4 acc[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) acc[i] = (f64x4){0.0, 0.0, 0.0, 0.0};
    auto apply = [&](const double *panel) {  // acc += panel' panel (128 k-rows) on this wave's sub-blocks
        auto stage = [&](int kt, double *st) {  // wave W moves rows W, W+4, W+8, W+12 of the k-tile
#pragma unroll
            for (int pp = 0; pp < 4; ++pp)
                dma_row(panel + (size_t)(kt * BK + W + 4 * pp) * ld + lane * 2, st + (W + 4 * pp) * LDS_LD);
        };
        stage(0, lds);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int kt = 0; kt < NB / BK; ++kt) {
            if (kt + 1 < NB / BK) stage(kt + 1, lds + ((kt + 1) & 1) * UPD_STAGE);
            const double *st = lds + (kt & 1) * UPD_STAGE;
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                double fr[8];
#pragma unroll
                for (int blk = 0; blk < 8; ++blk)
                    if (blk >= W) fr[blk] = st[(kk * 4 + lk) * LDS_LD + blk * 16 + lr];
#pragma unroll
                for (int i = 0; i < 9; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[T::rb[i]], fr[T::cb[i]], acc[i], 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    };
    // workgroup-uniform branches: every wave takes the same barriers.  The plain schedule (nkb == 1) runs the second
    // call only, whose code is the critical path of small batches.
    if (nkb > 1) apply(panel0);
    if (nkb > 0) apply(panel0 + (size_t)(nkb - 1) * NB * ld);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        double *blk = S + blk_off(T::rb[i], T::cb[i]);
#pragma unroll
        for (int v = 0; v < 4; ++v) blk[(lk + 4 * v) * SB + lr] = pre[i][v] - acc[i][v];
    }
}

// nkb == 0 (diag_pre_kernel has applied the block rows above, or j == 0): the stored tile only moves into the packed image.
// Wave 0 stores the sub-block (0,0) first and eliminates it while its other loads — and the other waves' — are still in
// flight (the tile load and factor16(0) used to be 4 K + 6 K cycles one after the other, with three waves idle in the second).
template <int W>
__device__ __forceinline__ void diag_copy(const double *__restrict__ tile, long ld, double *S, int lane, int lr, int lk, double &pacc,
                                          int &bad) {
    using T = UpperBlocks<W>;
    double pre[9][4];
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int v = 0; v < 4; ++v) pre[i][v] = tile[(size_t)(T::rb[i] * 16 + lk + 4 * v) * ld + T::cb[i] * 16 + lr];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        double *blk = S + blk_off(T::rb[i], T::cb[i]);
#pragma unroll
        for (int v = 0; v < 4; ++v) blk[(lk + 4 * v) * SB + lr] = pre[i][v];
        if (W == 0 && i == 0) {  // UpperBlocks<0>: sub-block (0,0)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            factor16(blk, lane, 0, pacc, bad);
        }
    }
}

// The diagonal tile of the one-launch kernels of several block rows (diag_kernel<true, 8>, two_block_kernel<8>, multi_block_kernel), eight waves:
// D = A_jj (generated) - sum over ALL block rows a < nkb of panel_a' panel_a (diag_update's loop with every block row above instead of
// the first and the last); the product stages alias S.
// Eight waves: wave W < 4 takes the entries [0, 5) of UpperBlocks<W>, wave W + 4 the entries [5, 9) — the product is bound by the
// MFMA pipes of the CU (36 sub-blocks x 32 MFMAs per block row), and four waves left every pipe half idle.
template <int W, int LO, int HI>
__device__ __forceinline__ void mb_update(const double *__restrict__ col0, long ld, int nkb, double *lds, double *S, int wave_u, int lane,
                                          int lr, int lk, const Mats &p, int b, int rep, const uint32_t *codes, int cs, int goff) {
    using T = UpperBlocks<W>;
    double pre[9][4];
    const GenCtx g = gen_ctx(p, b, rep);
#pragma unroll
    for (int i = LO; i < HI; ++i) {
        const int r0 = goff + T::rb[i] * 16 + lk;
        const int gi[4] = {r0, r0 + 4, r0 + 8, r0 + 12};
        gen_rows4(g, codes, cs, gi, goff + T::cb[i] * 16 + lr, pre[i]);
    }
    f64x4 acc[9];
#pragma unroll
    for (int i = LO; i < HI; ++i) acc[i] = (f64x4){0.0, 0.0, 0.0, 0.0};
    const int nk = nkb * (NB / BK);  // k-tiles of the whole panel U[0 : 128 nkb, j-block]
    auto stage = [&](int kt, double *st) {  // wave w moves rows w and w + 8 of the k-tile
        dma_row(col0 + (size_t)(kt * BK + wave_u) * ld + lane * 2, st + wave_u * LDS_LD);
        dma_row(col0 + (size_t)(kt * BK + wave_u + 8) * ld + lane * 2, st + (wave_u + 8) * LDS_LD);
    };
    if (nk > 0) {
        stage(0, lds);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) stage(kt + 1, lds + ((kt + 1) & 1) * UPD_STAGE);
            const double *st = lds + (kt & 1) * UPD_STAGE;
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                double fr[8];
#pragma unroll
                for (int blk = 0; blk < 8; ++blk)
                    if (blk >= W) fr[blk] = st[(kk * 4 + lk) * LDS_LD + blk * 16 + lr];
#pragma unroll
                for (int i = LO; i < HI; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(fr[T::rb[i]], fr[T::cb[i]], acc[i], 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
#pragma unroll
    for (int i = LO; i < HI; ++i) {
        double *blk = S + blk_off(T::rb[i], T::cb[i]);
#pragma unroll
        for (int v = 0; v < 4; ++v) blk[(lk + 4 * v) * SB + lr] = pre[i][v] - acc[i][v];
    }
}

// G_j = U[j-1,j] W_j for diag_kernel's epilogue (pipelined schedule; W_j upper triangular: column block cbk sums the
// row blocks rbk <= cbk).  Wave w owns the 16-row blocks 2w, 2w+1 of G.  A fragments (U[j-1,j], 16 rows x 4 columns per
// MFMA) come straight from L2 — solve(j-1) wrote the tile just before — and the B fragments are the sub-blocks of W_j
// still in S.  Out: Gb[k][r] = -G.  
__device__ __forceinline__ void diag_g(const double *__restrict__ Up, long ld, const double *S, double *__restrict__ Gb, int wave_u,
                                    int lr, int lk) {
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        const int kr = wave_u * 2 + h;
        const double *urow = Up + (size_t)(kr * SB + lr) * ld + lk;
        f64x4 g[NSB];
#pragma unroll
        for (int cbk = 0; cbk < NSB; ++cbk) g[cbk] = (f64x4){0.0, 0.0, 0.0, 0.0};
        double a[4], an[4];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) a[q4] = urow[q4 * 4];
#pragma unroll
        for (int rbk = 0; rbk < NSB; ++rbk) {  // k' ascending for every element of G
            if (rbk + 1 < NSB) {
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) an[q4] = urow[(rbk + 1) * SB + q4 * 4];
            }
#pragma unroll
            for (int cbk = rbk; cbk < NSB; ++cbk) {
                const double *wb = S + blk_off(rbk, cbk);
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)
                    g[cbk] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q4], wb[(q4 * 4 + lk) * SB + lr], g[cbk], 0, 0, 0);
            }
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) a[q4] = an[q4];
            __builtin_amdgcn_sched_barrier(0);  // keep the LDS reads of later row blocks from being hoisted (spills)
        }
#pragma unroll
        for (int cbk = 0; cbk < NSB; ++cbk)
#pragma unroll
            for (int v = 0; v < 4; ++v) Gb[(size_t)(kr * SB + lk + 4 * v) * NB + cbk * SB + lr] = -g[cbk][v];
    }
}

// Column kb of the block inverse for the row rb of X a wave owns (rb < kb):  X[rb,kb] = -(sum_{rb<=k<kb} X[rb,k] U[k,kb]) W_kk,
// computed TRANSPOSED and kept in registers.  xt[d] is the MFMA accumulator (D layout) of X[rb,rb+d]' — which is exactly the B
// fragment of the product  t' = sum_k U[k,kb]' X[rb,k]'  (A fragments: the k-major U blocks, read from LDS without bank
// conflicts), and t' in its D layout is the B fragment of  X[rb,kb]' = -(W_kk' t').  So a row's blocks never come back from
// LDS (read row-major as A operands they are 8/16-way bank-conflicted: ~1 K cycles per block product, 9 K for the last column),
// and the per-wave transpose scratch with its two wave barriers per entry is gone.  Per element the same products are summed
// in the same order as X U and t W_kk (k ascending, four k per MFMA): identical bits.  Returns X[rb,kb]' (stored to the packed
// image one step later, transposed back).
// A 16 x 16 block held as an MFMA A fragment (lane (lr, lk), register v: M[lr][lk + 4v]) -> the same block in the D layout
// (M[lk + 4v][lr]), by four MFMAs with the identity as B — exact (x * 1 + zeros; only the sign of a zero can change).  The
// lane pattern the other way round is a 16-double-stride LDS access: 8/16-way bank conflicts that stall the LDS pipeline of
// the whole CU, i.e. the elimination chain of wave 0 too.
__device__ __forceinline__ f64x4 frag_transpose(const f64x4 &a, int lr, int lk) {
    f64x4 d = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < SB / 4; ++kk) d = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], (lk + 4 * kk == lr) ? 1.0 : 0.0, d, 0, 0, 0);
    return d;
}

// `tacc` carries the sum one step ahead: the terms k <= kb-2 of column kb only need blocks that were final a step earlier, so
// step kb-1 accumulates them (the other waves have slack there; in the last step, when wave 0 has nothing left to do, a row's
// whole sum used to be on the critical path: 5.5 K cycles) and step kb adds the term k = kb-1 and applies W_kk — the same
// terms in the same order.
template <int L>
__device__ __forceinline__ f64x4 x_entry(f64x4 (&xt)[L], f64x4 &tacc, int rb, int kb, int nsb, const double *S, const double *dblk,
                                         int lr, int lk) {
    auto term = [&](f64x4 &t, int k, int col, const f64x4 &xf) {  // t += U[k,col]' X[rb,k]'
        const double *ub = S + blk_off(k, col);
#pragma unroll
        for (int kk = 0; kk < SB / 4; ++kk) t = __builtin_amdgcn_mfma_f64_16x16x4f64(ub[(kk * 4 + lk) * SB + lr], xf[kk], t, 0, 0, 0);
    };
    if (kb == rb + 1) {  // X[rb,rb] = W_rb (factor16 left it in S, row-major): read k-major, turned in registers
        const double *w = S + blk_off(rb, rb);
        f64x4 wk;
#pragma unroll
        for (int v = 0; v < 4; ++v) wk[v] = w[(lk + 4 * v) * SB + lr];
        xt[0] = frag_transpose(wk, lr, lk);
        tacc = (f64x4){0.0, 0.0, 0.0, 0.0};
    }
    f64x4 t = tacc;  // terms k = rb .. kb-2
#pragma unroll
    for (int d = 0; d < L; ++d)
        if (rb + d == kb - 1) term(t, kb - 1, kb, xt[d]);  // wave-uniform (MFMA ignores EXEC)
    f64x4 x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < SB / 4; ++kk) x = __builtin_amdgcn_mfma_f64_16x16x4f64(dblk[(kk * 4 + lk) * SB + lr], t[kk], x, 0, 0, 0);
    x = -x;
#pragma unroll
    for (int d = 1; d < L; ++d)
        if (rb + d == kb) xt[d] = x;
    if (kb + 1 < nsb) {  // column kb+1: its terms k = rb .. kb-1
        f64x4 tn = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int d = 0; d < L; ++d)
            if (rb + d < kb) term(tn, rb + d, kb + 1, xt[d]);
        tacc = tn;
    }
    return x;
}

// Trailing update inside diag_kernel: D[rb,cb] -= U[kb,rb]' U[kb,cb] for Q blocks of a wave's list (entries i0, i0 + 3, ... of
// the row-major list of the trailing sub-blocks after (kb+1, kb+1); n = trailing block rows): Q independent MFMA chains
// interleaved, the destination blocks requested before the products.
template <int Q, int STRIDE = 3>  // STRIDE: waves that share the list (3 of 4, or 7 of 8: factor_tile8)
__device__ __forceinline__ void c_group(double *S, int kb, int n, int i0, int lr, int lk) {
    const double *a[Q], *b[Q];
    double *dd[Q];
#pragma unroll
    for (int qq = 0; qq < Q; ++qq) {
        int r = 0, rem = i0 + STRIDE * qq + 1;  // + 1: the list starts after (kb+1, kb+1)
        while (rem >= n - r) {
            rem -= n - r;
            ++r;
        }
        const int rb = kb + 1 + r, cb = rb + rem;
        a[qq] = S + blk_off(kb, rb);
        b[qq] = S + blk_off(kb, cb);
        dd[qq] = S + blk_off(rb, cb);
    }
    f64x4 u[Q], dv[Q];
#pragma unroll
    for (int qq = 0; qq < Q; ++qq) {
        u[qq] = (f64x4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int v = 0; v < 4; ++v) dv[qq][v] = dd[qq][(lk + 4 * v) * SB + lr];
    }
#pragma unroll
    for (int kk = 0; kk < SB / 4; ++kk) {
        const int o = (kk * 4 + lk) * SB + lr;
#pragma unroll
        for (int qq = 0; qq < Q; ++qq) u[qq] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[qq][o], b[qq][o], u[qq], 0, 0, 0);
    }
#pragma unroll
    for (int qq = 0; qq < Q; ++qq)
#pragma unroll
        for (int v = 0; v < 4; ++v) dd[qq][(lk + 4 * v) * SB + lr] = dv[qq][v] - u[qq][v];
}

// nkb: trailing block rows of U still to be applied to the stored diagonal tile, D = P - sum_{j-nkb <= k < j} U[k,j]'U[k,j]
// (1 in the plain schedule, 2 in the pipelined one, 0 for j == 0).
// want_g (pipelined schedule): also rows 0..127 of Mats::W := -U[j-1,j] W_j, the dense half of solve_kernel<1>'s left operand.
// publish (chain-bound chunks): the launch stores its block step in p.sync[3] when it starts — the helper streams' gate
// kernels wait for that instead of an event recorded between solve(j-1) and this kernel (diag_pre_kernel does it instead
// when it runs in front of this one).
// wait_slot >= 0: before it ends, the workgroup waits (bounded: ~2 s, then info = -3) until the progress counter
// p.sync[wait_slot] has reached wait_value — the row launch whose tiles the NEXT kernel of this stream (solve(j)) reads
// has retired.  This replaces an event wait between diag(j) and solve(j) on the caller's stream: an unresolved
// cross-stream event wait costs ~5-13 us there, kernels back to back 0.8 us (tools/gap_probe.hip), and the row launch
// is normally long done.  Only this kernel spins — at most 32 workgroups, on CUs the row kernels do not need.  What keeps the
// scheme live: workgroup 0 publishes the block step in sync[3] when it STARTS (or diag_pre_kernel does, in front of it), the
// helper streams' gates wait for nothing else, the host enqueues the row launches behind those gates promptly, and the helper
// streams run beside this one; where they cannot (serialised dispatch, one hardware queue) the wait is BOUNDED — 2 s, sticky
// for the rest of the chunk, info = -3 — it is not a deadlock-freedom argument by enqueue order: the gate of a helper stream
// is enqueued before the diag_kernel it waits for, and the rows a diag_kernel waits for are enqueued after it.
// ONE: the one-launch evaluation of matrices of one block row (OneBlock; j == 0, nkb == 0) — an instantiation of its
// own, so that the regular kernel carries none of its code (with a run-time switch diag_kernel ran 52 -> 60 us).
// TWO (MODE 2, round 5): matrices of TWO block rows (128 < N <= 256 — BO with a couple of hundred points), the whole
// evaluation in one launch like ONE: block 0 is generated and factored as in ONE; then U_01 = W_0' A_01 with A_01 generated
// on the fly as the MFMA B operand (W_0 is still in the factor image; the product goes to the matrix's tile (0,1) in the
// workspace, L2-resident scratch), y_1 -= U_01' z_0 from the accumulators; then block 1 = A_11 (generated) - U_01' U_01 by the
// rank-128 update of the regular kernel, factored, and the MLL written.  Seven launches (walk, Gram tile, right-hand side,
// diag, rows, solve, diag) -> walk + this one: N = 256 x 256 forests 0.157 -> see profiles/r05/small_n.txt.
__device__ __forceinline__ void two_block_offdiag(const Mats &p, int b, int rep, const uint32_t *codes, const double *S,
                                                  const double *z0, double *__restrict__ U01, double *ysub, int ct0, int nct, int lr,
                                                  int lk) {
    const GenCtx g = gen_ctx(p, b, rep);
#pragma unroll 1
    for (int ct = ct0; ct < ct0 + nct; ++ct) {  // this wave's 16-column tiles of U_01 (two with four waves, one with eight)
        f64x4 acc[NSB];
#pragma unroll
        for (int rt = 0; rt < NSB; ++rt) acc[rt] = (f64x4){0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
        for (int kt = 0; kt < NSB; ++kt) {  // k ascending for every element; W_0 is upper triangular: row tiles rt >= kt only
            double bv[4];  // B fragments of the k-tile: A_01[kt * 16 + kk * 4 + lk][ct * 16 + lr], kk = 0..3
            {
                const int r0 = kt * SB + lk;
                const int gi[4] = {r0, r0 + 4, r0 + 8, r0 + 12};
                gen_rows4(g, codes, 2 * NB, gi, NB + ct * SB + lr, bv);
            }
#pragma unroll
            for (int rt = 0; rt < NSB; ++rt) {
                if (rt >= kt) {  // wave-uniform (MFMA ignores EXEC: a scalar branch)
                    const double *wb = S + blk_off(kt, rt);
#pragma unroll
                    for (int kk = 0; kk < SB / 4; ++kk)
                        acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(wb[(kk * 4 + lk) * SB + lr], bv[kk], acc[rt], 0, 0, 0);
                }
            }
        }
        double sum = 0.0;
#pragma unroll
        for (int rt = 0; rt < NSB; ++rt)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int r = rt * SB + lk + 4 * v;
                U01[(size_t)r * p.ld + ct * SB + lr] = acc[rt][v];
                sum = fma(acc[rt][v], z0[r], sum);
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        if (lk == 0) ysub[ct * SB + lr] = sum;
    }
}

// The blocked Cholesky + inverse of the tile in the factor image S (diag_kernel's middle; also the two blocks of two_block_kernel).
// copy_only: diag_copy has eliminated sub-block (0,0) already.  nsb: live 16-wide sub-blocks (ONE: the rest is identity padding).
// Out: S = the inverse W (upper block triangle), logsum / bad in wave 0.
template <bool ONE>
__device__ __forceinline__ void factor_tile(double *S, int nsb, bool copy_only, int wave_u, int lane, int lr, int lk, double &logsum,
                                            double &pacc, int &bad) {
    // --- blocked Cholesky D = U'U and X = U^-1, software-pipelined over the four waves ------------------------------
    // The serial part is the eight 16x16 eliminations (factor16, one wave, ~3.8 K cycles each; 6.0 K before round 4's MFMA form).  Wave 0 runs that
    // chain: in step kb it updates only the NEXT diagonal sub-block with row kb and factors it, while waves 1-3 do the
    // rest of step kb's trailing update (C) and the column kb of the inverse — so neither waits for the other:
    //   top of step kb (all waves)  (B) U[kb,cb] = W_kk' D[kb,cb], cb > kb                       | barrier
    //   wave 0                      D[kb+1,kb+1] -= U[kb,kb+1]'U[kb,kb+1];  factor16 -> W_{kb+1}
    //   waves 1-3                   store column kb-1 of X (computed last step, held in registers);
    //                               (C) D[rb,cb] -= U[kb,rb]'U[kb,cb] for the other (rb, cb);
    //                               X[rb,kb] = -(sum_{rb<=k<kb} X[rb,k] U[k,kb]) W_kk  -> registers           | barrier
    // A wave owns whole ROWS of X and keeps their blocks in registers (x_entry); a column of X goes to the packed image one
    // step after it was computed, when nobody reads the U blocks it replaces any more.
    // Same MFMA chain per element as the unpipelined order: identical results.
    // rows of X owned by this wave (-1: none).  Full tiles: {0,5}, {1,4}, {2,3} for waves 1-3 (7 + 2, 6 + 3, 5 + 4 block products
    // in the last step, the longest) and row 6 — one product, in the last step, when wave 0 has no elimination left — for
    // wave 0.  One-block-row matrices (any number of live sub-blocks): {0,6}, {1,4}, {2,3,5}, none for wave 0.
    const int xrow[3] = {wave_u == 1 ? 0 : wave_u == 2 ? 1 : wave_u == 3 ? 2 : (ONE ? -1 : 6),
                         wave_u == 1 ? (ONE ? 6 : 5) : wave_u == 2 ? 4 : wave_u == 3 ? 3 : -1, (ONE && wave_u == 3) ? 5 : -1};
    f64x4 pend[3];  // column kb of X (transposed, see x_entry) for the owned rows, stored at the start of the next step
    if (!copy_only) {
        if (wave_u == 0) factor16(S + blk_off(0, 0), lane, 0, pacc, bad);
        __syncthreads();
    }
    auto phase_b = [&](int kb, double *dblk) {  // (B) U[kb,cb] = W_kk' D[kb,cb]
        for (int cb = kb + 1 + wave_u; cb < nsb; cb += 4) {
            double *blk = S + blk_off(kb, cb);
            f64x4 u = {0.0, 0.0, 0.0, 0.0};
            mfma_tn(u, dblk, SB, blk, SB, lr, lk);
#pragma unroll
            for (int v = 0; v < 4; ++v) blk[(lk + 4 * v) * SB + lr] = u[v];
        }
    };
    // Wave 0 and waves 1-3 run the loop as two separate code regions (same barriers, two per step, in both): the elimination's
    // registers and the X rows the other waves keep in registers then never coexist in one wave's allocation.
    if (wave_u == 0) {
        f64x4 xw[2], tw;  // row nsb - 2 of X from the diagonal on (full tiles only)
        for (int kb = 0; kb < nsb; ++kb) {
            double *dblk = S + blk_off(kb, kb);  // W_kk
            DIAG_STAMP(8 + kb);
            phase_b(kb, dblk);
            __syncthreads();
            DIAG_STAMP(16 + kb);
            if (kb + 1 < nsb) {
                const double *urow = S + blk_off(kb, kb + 1);
                double *dst = S + blk_off(kb + 1, kb + 1);
                f64x4 u = {0.0, 0.0, 0.0, 0.0};
                mfma_tn(u, urow, SB, urow, SB, lr, lk);
#pragma unroll
                for (int v = 0; v < 4; ++v) dst[(lk + 4 * v) * SB + lr] -= u[v];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                factor16(dst, lane, (kb + 1) * SB, pacc, bad);
                DIAG_STAMP(24 + kb);
            } else {  // last step: nothing left to eliminate — the one entry of X's row 6, and the logs of all the pivots (this
                      // wave used to wait ~3 K cycles for the others here)
                if (!ONE) pend[0] = x_entry(xw, tw, xrow[0], kb, nsb, S, dblk, lr, lk);
                logsum = pivots_logsum(pacc);
            }
            __syncthreads();
        }
    } else {
        f64x4 xt0[7], xt1[4], xt2[ONE ? 2 : 1];  // the owned rows' blocks from the diagonal on (rows {0,1,2} | {3,4,5,6} | {5})
        f64x4 ta0, ta1, ta2;                      // ... and their next column's sum so far
        for (int kb = 0; kb < nsb; ++kb) {
            double *dblk = S + blk_off(kb, kb);  // W_kk
            phase_b(kb, dblk);
            __syncthreads();
            if (kb >= 2) {  // column kb-1 of X, computed in the previous step (held transposed: lane (lr, lk) has X[lr][lk + 4v])
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    if (xrow[i] >= 0 && xrow[i] < kb - 1) {
                        double *dst = S + blk_off(xrow[i], kb - 1);
                        const f64x4 xs = frag_transpose(pend[i], lr, lk);
#pragma unroll
                        for (int v = 0; v < 4; ++v) dst[(lk + 4 * v) * SB + lr] = xs[v];
                    }
            }
            {  // (C), all but the next diagonal sub-block: block p of the row-major list goes to wave p % 3 + 1, which takes its
               // blocks three at a time (c_group: independent MFMA chains interleaved; one block at a time cost ~960 cycles each,
               // and the first sub-block steps waited for these waves), then the two or one left over
                const int n = nsb - 1 - kb, total = n * (n + 1) / 2 - 1;
                int i0 = wave_u - 1;
                for (; i0 + 6 < total; i0 += 9) c_group<3>(S, kb, n, i0, lr, lk);
                if (i0 + 3 < total)
                    c_group<2>(S, kb, n, i0, lr, lk);
                else if (i0 < total)
                    c_group<1>(S, kb, n, i0, lr, lk);
            }
            if (kb >= 1) {  // column kb of X for the owned rows above the diagonal
                if (xrow[0] >= 0 && xrow[0] < kb) pend[0] = x_entry(xt0, ta0, xrow[0], kb, nsb, S, dblk, lr, lk);
                if (xrow[1] >= 0 && xrow[1] < kb) pend[1] = x_entry(xt1, ta1, xrow[1], kb, nsb, S, dblk, lr, lk);
                if (ONE && xrow[2] >= 0 && xrow[2] < kb) pend[2] = x_entry(xt2, ta2, xrow[2], kb, nsb, S, dblk, lr, lk);
            }
            __syncthreads();
        }
    }
    {  // last column of X
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (xrow[i] >= 0 && xrow[i] < nsb - 1) {
                double *dst = S + blk_off(xrow[i], nsb - 1);
                const f64x4 xs = frag_transpose(pend[i], lr, lk);
#pragma unroll
                for (int v = 0; v < 4; ++v) dst[(lk + 4 * v) * SB + lr] = xs[v];
            }
    }
    __syncthreads();

}

// factor_tile with EIGHT waves (round 5; chain-bound launches: a lone / few matrices, the one-launch kernels of small N).  Since
// round 4's MFMA form of factor16 (6.0 K -> 4.1 K cycles per sub-block step) the three helper waves bound most steps of the
// four-wave form — wave 0 waited 0.4-2.0 K cycles per step, 4 K in the last, for their trailing updates and inverse columns
// (profiles/r05/diag_waves.txt).  Here SIX helpers do that work — waves 1-3 and 5-7, two on each of SIMDs 1-3 — and wave 4 only
// keeps the barriers' count: it shares SIMD 0 with wave 0, and with work of its own it stretched the elimination chain from 4.1 K
// to 4.7-5.8 K cycles per step (same file), which ate what the helpers had gained.  A helper owns one row of X (the sixth: rows 5
// and 6), the trailing blocks go round the six, phase (B) round the seven working waves.  Same blocks, same MFMA chain per
// element, same two barriers per step: identical bits.  Not for launches that share CUs with row workgroups.
template <bool ONE>
__device__ __forceinline__ void factor_tile8(double *S, int nsb, bool copy_only, int wave_u, int lane, int lr, int lk, double &logsum,
                                             double &pacc, int &bad) {
    const int bw = wave_u < 4 ? wave_u : wave_u - 1;  // phase (B): index among the seven working waves (wave 4: none)
    const int h = wave_u < 4 ? wave_u - 1 : wave_u - 2;  // helper index 0..5 (waves 1-3, 5-7)
    const int xr = h;                                    // the row of X a helper owns; helper 5 owns row 6 as well
    f64x4 pend, pend6;  // column kb of the owned row(s) (transposed, see x_entry), stored at the start of the next step
    if (!copy_only) {
        if (wave_u == 0) factor16(S + blk_off(0, 0), lane, 0, pacc, bad);
        __syncthreads();
    }
    auto phase_b = [&](int kb, double *dblk) {  // (B) U[kb,cb] = W_kk' D[kb,cb]: at most one block per working wave
        const int cb = kb + 1 + bw;
        if (cb < nsb) {
            double *blk = S + blk_off(kb, cb);
            f64x4 u = {0.0, 0.0, 0.0, 0.0};
            mfma_tn(u, dblk, SB, blk, SB, lr, lk);
#pragma unroll
            for (int v = 0; v < 4; ++v) blk[(lk + 4 * v) * SB + lr] = u[v];
        }
    };
    if (wave_u == 0) {
        // The chain stays in registers from one elimination to the next: U[kb,kb+1] = W_kk' D[kb,kb+1] comes out of the MFMA in the
        // D layout, which IS both operand layouts of U'U (A[i][k] and B[k][j] of lane (lr, lk) at k-step kk are u[kk]), and
        // D[kb+1,kb+1] - U'U in that layout is what factor16 eliminates.  The block goes to LDS only for the other waves
        // ((C) reads U[kb,kb+1]); wave 0 used to write it, wait for the barrier, read it back twice, read-modify-write the
        // diagonal block and read that back: ~1 K of the step's 5.5 K cycles.  Same MFMA chains, same subtraction: identical bits.
        for (int kb = 0; kb < nsb; ++kb) {
            double *dblk = S + blk_off(kb, kb);  // W_kk
            DIAG_STAMP(8 + kb);
            f64x4 ub = {0.0, 0.0, 0.0, 0.0}, dnext = {0.0, 0.0, 0.0, 0.0};
            double *dst = S + blk_off(kb + 1 < nsb ? kb + 1 : kb, kb + 1 < nsb ? kb + 1 : kb);
            if (kb + 1 < nsb) {
#pragma unroll
                for (int v = 0; v < 4; ++v) dnext[v] = dst[(lk + 4 * v) * SB + lr];  // final since the last barrier
                double *blk = S + blk_off(kb, kb + 1);  // phase (B), this wave's block: cb = kb + 1
                mfma_tn(ub, dblk, SB, blk, SB, lr, lk);
#pragma unroll
                for (int v = 0; v < 4; ++v) blk[(lk + 4 * v) * SB + lr] = ub[v];
            }
            __syncthreads();
            DIAG_STAMP(16 + kb);
            if (kb + 1 < nsb) {
                f64x4 u = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < SB / 4; ++kk) u = __builtin_amdgcn_mfma_f64_16x16x4f64(ub[kk], ub[kk], u, 0, 0, 0);
                factor16_reg(dst, dnext - u, lane, (kb + 1) * SB, pacc, bad);
                DIAG_STAMP(24 + kb);
            } else {
                logsum = pivots_logsum(pacc);
            }
            __syncthreads();
        }
    } else if (wave_u == 4) {  // SIMD 0 belongs to the elimination chain
        for (int kb = 0; kb < nsb; ++kb) {
            __syncthreads();
            __syncthreads();
        }
    } else {
        f64x4 xt[7], ta;       // the owned row's blocks from the diagonal on, and its next column's sum so far
        f64x4 xt6[1], ta6;     // helper 5: the same for row 6
        auto store_col = [&](int row, int col, const f64x4 &pe) {
            double *dst = S + blk_off(row, col);
            const f64x4 xs = frag_transpose(pe, lr, lk);
#pragma unroll
            for (int v = 0; v < 4; ++v) dst[(lk + 4 * v) * SB + lr] = xs[v];
        };
        for (int kb = 0; kb < nsb; ++kb) {
            double *dblk = S + blk_off(kb, kb);  // W_kk
            phase_b(kb, dblk);
            __syncthreads();
            if (kb >= 2 && xr < kb - 1) store_col(xr, kb - 1, pend);  // column kb-1 of X, computed in the previous step
            // (row 6 has its only entry in the last step: stored after the loop)
            {  // (C), all but the next diagonal sub-block: block p of the row-major list goes to helper p % 6
                const int n = nsb - 1 - kb, total = n * (n + 1) / 2 - 1;
                int i0 = h;
                for (; i0 + 12 < total; i0 += 18) c_group<3, 6>(S, kb, n, i0, lr, lk);
                if (i0 + 6 < total)
                    c_group<2, 6>(S, kb, n, i0, lr, lk);
                else if (i0 < total)
                    c_group<1, 6>(S, kb, n, i0, lr, lk);
            }
            if (kb >= 1 && xr < kb) pend = x_entry(xt, ta, xr, kb, nsb, S, dblk, lr, lk);
            if (h == 5 && kb == 7) pend6 = x_entry(xt6, ta6, 6, kb, nsb, S, dblk, lr, lk);  // (wave-uniform)
            __syncthreads();
        }
        if (xr < nsb - 1) store_col(xr, nsb - 1, pend);  // last column of X
        if (h == 5 && nsb == NSB) store_col(6, 7, pend6);
    }
    __syncthreads();
}

// NW: waves per workgroup — 4, or 8 for chain-bound launches (factor_tile8; the phases around the factorisation stay with the
// first four waves, the others only keep the barriers' count).
template <bool ONE, int NW = 4>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void diag_kernel(Mats p, int j, int nkb, int want_g, int wait_slot, int wait_value, OneBlock ob,
                                                          int publish) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, b = blockIdx.x;
    const Lane q = lane_of(tid);
    const int wave = tid >> 6, lane = tid & 63;
    double *Ab = p.A + (size_t)b * p.bstride;
    double *tile = Ab + (size_t)j * NB * p.ld + (size_t)j * NB;
    // operands of the kernel's last phase, requested now so their latency hides behind the factorisation
    // (solve_kernel(j-1) finished updating y_j before this launch)
    if (publish && b == 0 && tid == 0 && p.sync)  // this launch has started: everything before it on this stream is done
        __hip_atomic_store(p.sync + 3, j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    constexpr bool one = ONE;
    const double y_in = tid < NB ? (one ? (tid < p.N ? ob.y[tid] : 0.0) : p.yz[(size_t)b * p.nrb * NB + (size_t)j * NB + tid]) : 0.0;
    const double acc_quad = (tid == 0 && !one) ? p.accum[(size_t)b * 2 + 0] : 0.0;
    const double acc_logdet = (tid == 0 && !one) ? p.accum[(size_t)b * 2 + 1] : 0.0;
    const int info_in = (tid == 0 && !one) ? p.info[b] : 0;
    DIAG_STAMP(0);

    double *S = lds;                                          // packed upper block triangle, NBLK x [16][16]
    double *vec = lds + NBLK * SB * SB;                       // [2][128] y | upper-half partial sums
    double *red = vec + 2 * NB;                               // [8]
    uint32_t *codes = nullptr;
    if (one) {  // the matrix's leaf codes (nW x 128 dwords) behind everything else in LDS
        codes = reinterpret_cast<uint32_t *>(red + 8);
        if (NW == 8 && ob.nodes) {
            // ... produced HERE: the points' rows and the forest's packed nodes are staged where the factor image will be, thread
            // (point t & 127, group t >> 7) walks a quarter of the trees (leaf_walk_grouped_kernel's scheme with four groups),
            // one-hot bits are OR-ed into the code words in LDS, a byte-code word (four trees) belongs to one group.
            const int sd = ob.d | 1, nn = ob.m * ob.stride;
            double *xs = lds;                                                                   // [N][sd]
            uint4 *ln = reinterpret_cast<uint4 *>(lds + ((p.N * sd + 1) & ~1));                 // [m][stride]
            for (int e = tid; e < p.N * ob.d; e += NW * 64) {
                const int r = e / ob.d, c = e - r * ob.d;
                xs[r * sd + c] = ob.X[e];
            }
            const uint4 *forest = ob.nodes + (size_t)b * nn;
            for (int e = tid; e < nn; e += NW * 64) ln[e] = forest[e];
            for (int e = tid; e < p.nW * NB; e += NW * 64) codes[e] = 0;
            __syncthreads();
            const int pl = tid & (NB - 1), g = tid >> 7;  // four groups of 128 threads
            if (pl < p.N) {
                const double *xrow = xs + pl * sd;
                if (ob.rep == REP_BITS) {
                    const int t0 = (int)(((long)ob.m * g) / 4), t1 = (int)(((long)ob.m * (g + 1)) / 4);
                    for (int t = t0; t < t1; ++t) {
                        const uint32_t bit = walk_tree<true>(ln + (size_t)t * ob.stride, ob.max_depth, xrow, ob.fault_w).z;
                        const int w = (int)(bit >> 5);
                        if (w < p.nW) atomicOr(&codes[w * NB + pl], 1u << (bit & 31u));
                    }
                } else {
                    const int W = (ob.m + 3) >> 2;
                    const int w0 = (int)(((long)W * g) / 4), w1 = (int)(((long)W * (g + 1)) / 4);
                    for (int w = w0; w < w1; ++w) {
                        uint32_t word = 0;
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq) {
                            const int t = w * 4 + qq;
                            if (t < ob.m) word |= (walk_tree<true>(ln + (size_t)t * ob.stride, ob.max_depth, xrow, ob.fault_w).x & 0xFFu) << (8 * qq);
                        }
                        codes[w * NB + pl] = word;
                    }
                }
            }
        } else {
            const uint32_t *lb = p.leafx + (size_t)b * p.nW * NB;  // npad == 128
            for (int e = tid; e < p.nW * NB; e += NW * 64) codes[e] = lb[e];
        }
        __syncthreads();
    }
    double logsum = 0.0;  // sum of log(pivot) / 2 ... (wave 0; pivots_logsum in the last sub-block step)
    double pacc = 1.0;    // ... from factor16's pivot products
    int bad = 0;
    const bool copy_only = !ONE && nkb == 0;  // workgroup-uniform
    if (copy_only) {
        const int wsel = __builtin_amdgcn_readfirstlane(wave);
        if (wsel == 0)
            diag_copy<0>(tile, p.ld, S, lane, q.lr, q.lk, pacc, bad);
        else if (wsel == 1)
            diag_copy<1>(tile, p.ld, S, lane, q.lr, q.lk, pacc, bad);
        else if (wsel == 2)
            diag_copy<2>(tile, p.ld, S, lane, q.lr, q.lk, pacc, bad);
        else if (NW == 4 || wsel == 3)
            diag_copy<3>(tile, p.ld, S, lane, q.lr, q.lk, pacc, bad);
    } else {
        // D = P - sum_k U[k,j]'U[k,j] on the upper block triangle (the product stages alias S: the update's last barrier
        // precedes the writes of S)
        const double *prev = Ab + (size_t)(j - nkb) * NB * p.ld + (size_t)j * NB;  // U[j-nkb, j]
        const int wsel = __builtin_amdgcn_readfirstlane(wave);
        if (ONE && NW == 8) {  // the one-launch kernel of N <= 128: the tile is generated by all eight waves (nkb == 0: no product loop)
            switch (wsel) {
                case 0: mb_update<0, 0, 5>(prev, p.ld, 0, lds, S, wsel, lane, q.lr, q.lk, p, b, ob.rep, codes, NB, 0); break;
                case 1: mb_update<1, 0, 5>(prev, p.ld, 0, lds, S, wsel, lane, q.lr, q.lk, p, b, ob.rep, codes, NB, 0); break;
                case 2: mb_update<2, 0, 5>(prev, p.ld, 0, lds, S, wsel, lane, q.lr, q.lk, p, b, ob.rep, codes, NB, 0); break;
                case 3: mb_update<3, 0, 5>(prev, p.ld, 0, lds, S, wsel, lane, q.lr, q.lk, p, b, ob.rep, codes, NB, 0); break;
                case 4: mb_update<0, 5, 9>(prev, p.ld, 0, lds, S, wsel, lane, q.lr, q.lk, p, b, ob.rep, codes, NB, 0); break;
                case 5: mb_update<1, 5, 9>(prev, p.ld, 0, lds, S, wsel, lane, q.lr, q.lk, p, b, ob.rep, codes, NB, 0); break;
                case 6: mb_update<2, 5, 9>(prev, p.ld, 0, lds, S, wsel, lane, q.lr, q.lk, p, b, ob.rep, codes, NB, 0); break;
                default: mb_update<3, 5, 9>(prev, p.ld, 0, lds, S, wsel, lane, q.lr, q.lk, p, b, ob.rep, codes, NB, 0); break;
            }
        } else if (wsel == 0)
            diag_update<0, ONE>(tile, p.ld, prev, nkb, lds, S, lane, q.lr, q.lk, p, b, ob.rep, codes);
        else if (wsel == 1)
            diag_update<1, ONE>(tile, p.ld, prev, nkb, lds, S, lane, q.lr, q.lk, p, b, ob.rep, codes);
        else if (wsel == 2)
            diag_update<2, ONE>(tile, p.ld, prev, nkb, lds, S, lane, q.lr, q.lk, p, b, ob.rep, codes);
        else if (NW == 4 || wsel == 3)
            diag_update<3, ONE>(tile, p.ld, prev, nkb, lds, S, lane, q.lr, q.lk, p, b, ob.rep, codes);
        else  // waves 4-7: the barriers of the update's product loops (diag_update: 1 + 8 per block row applied, two at most)
            for (int a = (nkb > 1 ? 2 : nkb) * (1 + NB / BK); a > 0; --a) __syncthreads();
    }
    __syncthreads();

    const int lr = q.lr, lk = q.lk;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // one-block-row matrices of fewer than 113 points: the sub-blocks beyond the last live one are identity padding — their
    // factor, their inverse and their share of log|D| are what the tile generation left there, nothing to compute
    const int nsb = one ? (p.N + SB - 1) / SB : NSB;
    DIAG_STAMP(1);
    if (NW == 8)
        factor_tile8<ONE>(S, nsb, copy_only, wave_u, lane, lr, lk, logsum, pacc, bad);
    else
        factor_tile<ONE>(S, nsb, copy_only, wave_u, lane, lr, lk, logsum, pacc, bad);
    DIAG_STAMP(2);

    // --- W_j out, sub-block by sub-block (explicit zeros below the block diagonal: solve_kernel multiplies the full
    // tile; the diagonal sub-blocks are upper triangular with exact zeros already) --------------------------------
    double *Wb = w_block(p, b);
    auto write_w = [&](int t) {  // t: 0 .. 255
        // the 36 sub-blocks on or above the block diagonal only: every consumer skips the k-tiles below it (gemm_upper_tri,
        // solve_narrow_kernel, solve_direct_kernel), so what the buffer holds there never reaches an MFMA.  Two doubles per
        // thread, the two halves of the 256 threads on alternate sub-blocks.
        const int half = __builtin_amdgcn_readfirstlane(t >> 7), e = 2 * (t & 127), r = e >> 4, c = e & 15;
        int cnt = 0;
#pragma unroll
        for (int rbk = 0; rbk < NSB; ++rbk)
#pragma unroll
            for (int cbk = rbk; cbk < NSB; ++cbk, ++cnt)
                if ((cnt & 1) == half)
                    *reinterpret_cast<f64x2 *>(Wb + (size_t)(rbk * SB + r) * NB + cbk * SB + c) =
                        *reinterpret_cast<const f64x2 *>(S + blk_off(rbk, cbk) + e);
    };
    // (nobody reads W_0 of a one-block-row matrix.)  Eight waves: waves 4-7 write W_j at the very end, beside the first four
    // waves' z_j — they only pass the barriers of that phase first.
    if (!one && NW == 4) write_w(tid);

    DIAG_STAMP(3);
    if (want_g && (NW == 4 || wave_u < 4))  // workgroup-uniform (wave-uniform with eight waves: no barrier inside)
        diag_g(Ab + (size_t)(j - 1) * NB * p.ld + (size_t)j * NB, p.ld, S, p.W + (size_t)b * W_STRIDE, wave_u, lr, lk);

    DIAG_STAMP(4);
    // --- z_j = W_j' y_j ; quad += |z_j|^2 ; logdet += 2 sum log u_kk ----------------------------
    // thread (c, half) sums the sub-block rows 4*half .. 4*half+3 of column c (only sub-blocks on or above the block
    // diagonal exist: W_j is upper triangular).  y_j and the accumulators were loaded at kernel entry.
    double *yb = p.yz + (size_t)b * p.nrb * NB + (size_t)j * NB;
    if (tid < NB) vec[tid] = y_in;
    __syncthreads();
    {
        const int c = tid & (NB - 1), half = tid >> 7, cbk = c >> 4, cc = c & 15;  // (eight waves: half = 2, 3 do nothing)
        double part = 0.0;
#pragma unroll
        for (int i = 0; i < NSB / 2; ++i) {
            const int rbk = half * (NSB / 2) + i;
            if (rbk <= cbk && (NW == 4 || half < 2)) {
                const double *col = S + blk_off(rbk, cbk) + cc;
#pragma unroll
                for (int rr = 0; rr < SB; ++rr) part = fma(col[rr * SB], vec[rbk * SB + rr], part);
            }
        }
        if (half == 1) vec[NB + c] = part;  // vec has 2 * NB doubles
        __syncthreads();
        double zz = 0.0;
        if (!half) {
            const double z = part + vec[NB + c];
            if (!one) yb[c] = z;
            zz = z * z;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) zz += __shfl_xor(zz, off);
        if (lane == 0) red[wave] = zz;
    }
    __syncthreads();
    DIAG_STAMP(5);
    if (!one && NW == 8 && tid >= THREADS) write_w(tid - THREADS);
    if (tid == 0 && one) {  // finish_mll_kernel's arithmetic (quick_inverse.py:38 / mcmc_record_mll.py:73)
        double v = -(red[0] + red[1]) - 2.0 * logsum;
        if (ob.include_2pi) v = v - (double)p.N * log(2.0 * M_PI);
        ob.mll[b] = 0.5 * v;
        p.info[b] = *ob.fault ? -1 : (bad ? bad : 0);
    } else if (tid == 0) {  // wave 0 ran factor16: its logsum / bad are the matrix's
        p.accum[(size_t)b * 2 + 0] = acc_quad + (red[0] + red[1]);
        p.accum[(size_t)b * 2 + 1] = acc_logdet + 2.0 * logsum;
        int code = (bad && info_in == 0) ? j * NB + bad : 0;
        if (wait_slot >= 0) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
            const int32_t *flag = p.sync + wait_slot;
            bool ok = true;
            // a time-out is sticky (sync[2] != 0): once one wait of the chunk has run into its bound every later one gives up
            // at once — under serialised dispatch (rocprofv3 --pmc, AMD_SERIALIZE_KERNEL) every wait would time out, and 2 s
            // per block step is minutes at N = 16384; now the whole call costs one bound
            while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < wait_value) {
                if (__hip_atomic_load(p.sync + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                    ok = false;
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
                if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {  // 2 s: the row stream is not running beside us
                    ok = false;
                    break;
                }
            }
            if (!ok) {
                atomicAdd(p.sync + 2, 1);
                code = -3;
            }
        }
        if (code != 0 && (info_in == 0 || code == -3)) p.info[b] = code;
        if (ob.mll) {  // last block step of an MLL-only sweep: finish_mll_kernel's arithmetic here, one launch less
            if (*ob.fault) p.info[b] = -1;
            if (p.sync && __hip_atomic_load(p.sync + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) p.info[b] = -3;
            double v = -(acc_quad + (red[0] + red[1])) - (acc_logdet + 2.0 * logsum);
            if (ob.include_2pi) v = v - (double)p.N * log(2.0 * M_PI);
            ob.mll[b] = 0.5 * v;
        }
    }
}

// two_block_kernel (round 5): matrices of TWO block rows (128 < N <= 256 — BO with a couple of hundred points), the whole
// evaluation in one launch, as diag_kernel<true> does it for one block row: block 0 is generated and factored; then
// U_01 = W_0' A_01 with A_01 generated on the fly as the MFMA B operand (two_block_offdiag: W_0 is still in the factor image; the
// product goes to the matrix's tile (0,1) of the workspace — L2-resident scratch — and y_1 -= U_01' z_0 comes from the
// accumulators); then block 1 = A_11 (generated) - U_01' U_01 by the regular rank-128 update (diag_update), factored, and
// the MLL written.  Seven launches (leaf walk, Gram tile, right-hand side, diag, rows, solve, diag) become two.  A kernel of
// its own, not a third mode of diag_kernel: wrapping that kernel's body in a loop over the blocks took the regular
// instantiation from 223 VGPRs to 256 with scratch.
#ifdef BARK_TWO_STAMPS
__device__ unsigned long long g_two_stamps[16];
#endif
template <int NW>  // waves per workgroup: 4, or 8 (factor_tile8) while a chunk has a CU per matrix
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void two_block_kernel(Mats p, OneBlock ob) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, b = blockIdx.x;
    const Lane q = lane_of(tid);
    const int wave = tid >> 6, lane = tid & 63, lr = q.lr, lk = q.lk;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    double *Ab = p.A + (size_t)b * p.bstride;
    double *S = lds;                      // packed upper block triangle, NBLK x [16][16]
    double *vec = lds + NBLK * SB * SB;   // [2][128] y | upper-half partial sums
    double *red = vec + 2 * NB;           // [8]
    constexpr int CS = 2 * NB;            // points per code plane
    uint32_t *codes = reinterpret_cast<uint32_t *>(red + 8);          // [nW][256]
    double *zsave = reinterpret_cast<double *>(codes + p.nW * CS);    // z_0
    double *ysub = zsave + NB;                                        // U_01' z_0
    {
        const uint32_t *lb = p.leafx + (size_t)b * p.nW * CS;  // npad == 256
        for (int e = tid; e < p.nW * CS; e += NW * 64) codes[e] = lb[e];
    }
    double y_in = tid < NB ? ob.y[tid] : 0.0;  // N > 128
    __syncthreads();
    double quad_sum = 0.0, logsum_sum = 0.0;  // over the two blocks (thread 0: wave 0 ran factor16)
    int bad_all = 0;
#ifdef BARK_TWO_STAMPS
    int stamp_n = 0;
#define TWO_STAMP() do { if (tid == 0 && b == 0) g_two_stamps[stamp_n++] = __builtin_readcyclecounter(); } while (0)
#else
#define TWO_STAMP() do {} while (0)
#endif
    TWO_STAMP();
    auto block = [&](auto BLK) {
        constexpr int blk = decltype(BLK)::value;
        double logsum = 0.0, pacc = 1.0;
        int bad = 0;
        if (NW == 8) {  // the 36 sub-blocks over eight waves (mb_update); every wave runs the product loop's barriers
            switch (wave_u) {
                case 0: mb_update<0, 0, 5>(Ab + NB, p.ld, blk, lds, S, wave_u, lane, lr, lk, p, b, ob.rep, codes, CS, blk * NB); break;
                case 1: mb_update<1, 0, 5>(Ab + NB, p.ld, blk, lds, S, wave_u, lane, lr, lk, p, b, ob.rep, codes, CS, blk * NB); break;
                case 2: mb_update<2, 0, 5>(Ab + NB, p.ld, blk, lds, S, wave_u, lane, lr, lk, p, b, ob.rep, codes, CS, blk * NB); break;
                case 3: mb_update<3, 0, 5>(Ab + NB, p.ld, blk, lds, S, wave_u, lane, lr, lk, p, b, ob.rep, codes, CS, blk * NB); break;
                case 4: mb_update<0, 5, 9>(Ab + NB, p.ld, blk, lds, S, wave_u, lane, lr, lk, p, b, ob.rep, codes, CS, blk * NB); break;
                case 5: mb_update<1, 5, 9>(Ab + NB, p.ld, blk, lds, S, wave_u, lane, lr, lk, p, b, ob.rep, codes, CS, blk * NB); break;
                case 6: mb_update<2, 5, 9>(Ab + NB, p.ld, blk, lds, S, wave_u, lane, lr, lk, p, b, ob.rep, codes, CS, blk * NB); break;
                default: mb_update<3, 5, 9>(Ab + NB, p.ld, blk, lds, S, wave_u, lane, lr, lk, p, b, ob.rep, codes, CS, blk * NB); break;
            }
        } else if (wave_u == 0)
            diag_update<0, true>(nullptr, p.ld, Ab + NB, blk, lds, S, lane, lr, lk, p, b, ob.rep, codes, CS, blk * NB);
        else if (wave_u == 1)
            diag_update<1, true>(nullptr, p.ld, Ab + NB, blk, lds, S, lane, lr, lk, p, b, ob.rep, codes, CS, blk * NB);
        else if (wave_u == 2)
            diag_update<2, true>(nullptr, p.ld, Ab + NB, blk, lds, S, lane, lr, lk, p, b, ob.rep, codes, CS, blk * NB);
        else
            diag_update<3, true>(nullptr, p.ld, Ab + NB, blk, lds, S, lane, lr, lk, p, b, ob.rep, codes, CS, blk * NB);
        __syncthreads();
        TWO_STAMP();
        const int nsb = blk == 0 ? NSB : (p.N - NB + SB - 1) / SB;  // the second block's live sub-blocks; the rest is identity padding
        if (NW == 8)
            factor_tile8<true>(S, nsb, false, wave_u, lane, lr, lk, logsum, pacc, bad);
        else
            factor_tile<true>(S, nsb, false, wave_u, lane, lr, lk, logsum, pacc, bad);
        TWO_STAMP();
        // z = W' y ; |z|^2 (diag_kernel's last phase)
        if (tid < NB) vec[tid] = y_in;
        __syncthreads();
        {
            const int c = tid & (NB - 1), half = tid >> 7, cbk = c >> 4, cc = c & 15;  // (eight waves: half = 2, 3 do nothing)
            double part = 0.0;
#pragma unroll
            for (int i = 0; i < NSB / 2; ++i) {
                const int rbk = half * (NSB / 2) + i;
                if (rbk <= cbk && (NW == 4 || half < 2)) {
                    const double *col = S + blk_off(rbk, cbk) + cc;
#pragma unroll
                    for (int rr = 0; rr < SB; ++rr) part = fma(col[rr * SB], vec[rbk * SB + rr], part);
                }
            }
            if (half == 1) vec[NB + c] = part;
            __syncthreads();
            double zz = 0.0;
            if (!half) {
                const double z = part + vec[NB + c];
                zsave[c] = z;
                zz = z * z;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) zz += __shfl_xor(zz, off);
            if (lane == 0) red[wave] = zz;
        }
        __syncthreads();
        quad_sum += red[0] + red[1];
        logsum_sum += logsum;
        bad_all = (bad && !bad_all) ? blk * NB + bad : bad_all;
        TWO_STAMP();
        if (blk == 0) {
            two_block_offdiag(p, b, ob.rep, codes, S, zsave, Ab + NB, ysub, NW == 8 ? wave_u : 2 * wave_u, NW == 8 ? 1 : 2, lr, lk);
            __syncthreads();  // U_01 (global: this workgroup's own stores) and ysub are visible to every wave
            TWO_STAMP();
            y_in = tid < NB ? ((NB + tid < p.N ? ob.y[NB + tid] : 0.0) - ysub[tid]) : 0.0;
        }
    };
    block(std::integral_constant<int, 0>{});
    block(std::integral_constant<int, 1>{});
    if (tid == 0) {  // finish_mll_kernel's arithmetic (quick_inverse.py:38 / mcmc_record_mll.py:73)
        double v = -quad_sum - 2.0 * logsum_sum;
        if (ob.include_2pi) v = v - (double)p.N * log(2.0 * M_PI);
        ob.mll[b] = 0.5 * v;
        p.info[b] = *ob.fault ? -1 : bad_all;
    }
}

// ---------------------------------------------------------------------------------------------
// multi_block_kernel (round 5): matrices of THREE to SIX block rows (256 < N <= 768) in one launch, eight waves, one workgroup
// per matrix — two_block_kernel's scheme with the off-diagonal GEMMs in the same workgroup.  The multi-launch sweep of 256
// such matrices moves every tile through HBM three times (T written by the row launch, read and rewritten as U by the solve,
// read as a panel by the next row launch: 0.5 GB per pass against 36 MB of L2) and takes 0.52 ms at N = 512; a CU's own MFMA
// share of one matrix is 146 us.  Per block step j:
//   D_j = A_jj (generated) - sum_{a<j} U[a,j]'U[a,j]     mb_update: diag_update's product loop over ALL block rows above
//   factor_tile8 -> W_j ; z_j ; log|D_j|
//   for every block column c > j (a wave owns 16 columns of the tile):
//       T = A_jc (generated) - sum_{r < 128 j} U[r, j-block]' U[r, c-cols]    A fragments from an LDS-DMA stage of the panel
//                                                                            (a ring of three 16-row stages beside the factor image),
//                                                                            B fragments straight from L2 (16 columns a wave)
//       U[j,c] = W_j' T     T in its accumulators IS the B operand of this product (D layout == B layout per 16 x 16 block)
//       y_c -= U[j,c]' z_j  (y and z live in LDS)
// U tiles go to the workspace (L2-resident scratch); nothing else leaves the CU until the MLL.  Per element the k order of every
// sum is ascending as in the sweep; the right-hand-side update sums in two_block_kernel's order.
// ---------------------------------------------------------------------------------------------
constexpr int MB_MAX_NRB = 6;  // block rows of a matrix in multi_block_kernel (LDS: y and z by the matrix's OWN count, the codes of all its points)
constexpr int MB_STAGE = BK * LDS_LD;  // one operand: 16 rows x (128 + 16) doubles

#ifdef BARK_TWO_STAMPS
__device__ unsigned long long g_mb_stamps[64];
#define MBO_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && j == 1 && c == 2) g_mb_stamps[40 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define MBO_STAMP(i) do {} while (0)
#endif
// Tile (j, c), this wave's 16 columns ct: T, then U[j,c] = W_j' T, stored; y_c -= U' z_j in LDS.  All eight waves (barriers inside).
// NOT inlined: inside multi_block_kernel's body its 64 + 64 accumulator registers beside the factorisation's spilled ~290 VGPRs;
// as a function of its own the kernel allocates 248 and spills none.  The LDS arrays arrive as offsets into the kernel's dynamic LDS
// (pointers through a call would be generic: flat loads instead of ds_read).
// a wave-uniform value that arrived in vector registers (an argument of a call), back in scalar registers
__device__ __forceinline__ long uniform_i64(long v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(unsigned long)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)((unsigned long)v >> 32));
    return (long)(((unsigned long)hi << 32) | lo);
}
__device__ __forceinline__ double uniform_f64(double v) { return __longlong_as_double(uniform_i64(__double_as_longlong(v))); }
template <int j>
__device__ __attribute__((noinline)) void mb_offdiag(GenCtx g, long ld_in, int cs_in, int stage_off, int codes_off, double *Ab_generic,
                                                     int c_in, int zj_off, int yc_off, int wave_in, int lane, int lr, int lk) {
    // (arguments of a call arrive in vector registers: the uniform ones go back to scalars — the loop over the code planes in
    // gen_rows4 is then a scalar loop with scalar address arithmetic instead of an exec-masked one with v_mul_lo_u32 per plane)
    const int c = __builtin_amdgcn_readfirstlane(c_in);
    const int cs = __builtin_amdgcn_readfirstlane(cs_in);
    const int wave_u = __builtin_amdgcn_readfirstlane(wave_in);
    // The matrix's generation constants and stride arrive BY VALUE, the matrix pointer is cast back to global memory.  As first
    // written the function took `const Mats &p`: the kernel's argument struct then had to live in scratch memory, every use of a
    // field — here AND in the kernel around the call — became a scratch reload plus a flat access behind an s_waitcnt vmcnt(0),
    // and the ring below never had two tiles in flight (round 5, profiles/r05/small_n.txt).
    typedef __attribute__((address_space(1))) double gdouble;
    gdouble *const Ab = (gdouble *)Ab_generic;
    const long ld = uniform_i64(ld_in);
    g.inv_m = uniform_f64(g.inv_m);
    g.sc = uniform_f64(g.sc);
    g.sh = uniform_f64(g.sh);
    g.jitter = uniform_f64(g.jitter);
    g.rep = __builtin_amdgcn_readfirstlane(g.rep);
    g.nW = __builtin_amdgcn_readfirstlane(g.nW);
    g.N = __builtin_amdgcn_readfirstlane(g.N);
    g.m = __builtin_amdgcn_readfirstlane(g.m);
    g.has_scale = __builtin_amdgcn_readfirstlane(g.has_scale);
    g.has_shift = __builtin_amdgcn_readfirstlane(g.has_shift);
    extern __shared__ __attribute__((aligned(16))) double mb_lds[];
    const double *S = mb_lds;
    double *stage = mb_lds + __builtin_amdgcn_readfirstlane(stage_off);
    const uint32_t *codes = reinterpret_cast<const uint32_t *>(mb_lds + __builtin_amdgcn_readfirstlane(codes_off));
    const double *zj = mb_lds + __builtin_amdgcn_readfirstlane(zj_off);
    double *yc = mb_lds + __builtin_amdgcn_readfirstlane(yc_off);
    MBO_STAMP(0);
    const int ct = wave_u;
    f64x4 tacc[NSB];
#pragma unroll
    for (int kt = 0; kt < NSB; ++kt) tacc[kt] = (f64x4){0.0, 0.0, 0.0, 0.0};
    constexpr int nk = j * (NB / BK);  // the block step is a template argument: the ring below is straight-line code, every
                                       // wait a constant (as a run-time loop the joins of its tail conditions cost a vmcnt(0) a tile)
    if constexpr (nk > 0) {
        // running pointers (the tiles are issued in order): this wave's two rows of the A k-tile (rows w and w + 8) and this lane's B
        // element of row 4 s + lk
        const gdouble *ap = Ab + (size_t)j * NB + (size_t)wave_u * ld + lane * 2;  // U[0 : 128 j, j-block]: rows k, 128 columns
        const gdouble *bp = Ab + (size_t)c * NB + ct * SB + lr + (size_t)lk * ld;
        const long ld4 = 4 * ld, ld8 = 8 * ld;
        // ring of three stages, two in flight: the 32 MFMAs of a k-tile are ~2 K cycles a wave, a DMA round trip is more.  The B
        // elements ride in three register sets by the stage's slot (the loop is unrolled by three: no copies of values in flight)
        double bb[3][4];
        auto issue = [&](int slot) {  // two DMA rows + four B elements: six VM operations a wave
            double *st = stage + slot * MB_STAGE;
            dma_row((const double *)ap, st + wave_u * LDS_LD);
            dma_row((const double *)(ap + ld8), st + (wave_u + 8) * LDS_LD);
            ap += 2 * ld8;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {  // (as assembly: the compiler would guard every later use with a vmcnt(0); the counted wait of
                                              // the step that consumes the slot covers these loads)
                asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(bb[slot][kk]) : "v"(bp) : "memory");
                bp += ld4;
            }
        };
        auto step = [&](int kt, int slot) {  // slot == kt % 3, a compile-time constant at every call
            if (kt + 1 < nk)
                asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (kt + 2 < nk) issue((slot + 2) % 3);  // that slot was read in iteration kt - 1: every wave is past the barrier
            const double *st = stage + slot * MB_STAGE;
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
#pragma unroll
                for (int t8 = 0; t8 < NSB; ++t8)
                    tacc[t8] = __builtin_amdgcn_mfma_f64_16x16x4f64(st[(kk * 4 + lk) * LDS_LD + t8 * SB + lr], bb[slot][kk], tacc[t8], 0, 0, 0);
            }
        };
        issue(0);
        issue(1);
#pragma unroll
        for (int kt = 0; kt < nk; kt += 3) {
            step(kt, 0);
            if (kt + 1 < nk) step(kt + 1, 1);
            if (kt + 2 < nk) step(kt + 2, 2);
        }
        __syncthreads();
    }
    MBO_STAMP(1);
    // T = A_jc - (the sum) in place: tacc[kt] becomes the B fragments of k-tile kt (T[16 kt + 4 kk + lk][16 ct + lr], kk = 0..3)
#pragma unroll
    for (int kt = 0; kt < NSB; ++kt) {
        double av[4];
        const int r0 = j * NB + kt * SB + lk;
        const int gi[4] = {r0, r0 + 4, r0 + 8, r0 + 12};
        gen_rows4(g, codes, cs, gi, c * NB + ct * SB + lr, av);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) tacc[kt][kk] = av[kk] - tacc[kt][kk];
        __builtin_amdgcn_sched_barrier(0);  // one k-tile's generation at a time
    }
    MBO_STAMP(2);
    f64x4 acc[NSB];
#pragma unroll
    for (int rt = 0; rt < NSB; ++rt) acc[rt] = (f64x4){0.0, 0.0, 0.0, 0.0};
    gdouble *U = Ab + (size_t)j * NB * ld + (size_t)c * NB + (size_t)lk * ld + ct * SB + lr;  // this lane's element of row lk
    const long ld4 = 4 * ld;
    double sum = 0.0;
#pragma unroll
    for (int kt = 0; kt < NSB; ++kt) {  // k ascending for every element; W_j is upper triangular: row tiles rt >= kt only
#pragma unroll
        for (int rt = 0; rt < NSB; ++rt) {
            if (rt >= kt) {
                const double *wb = S + blk_off(kt, rt);
#pragma unroll
                for (int kk = 0; kk < SB / 4; ++kk)
                    acc[rt] = __builtin_amdgcn_mfma_f64_16x16x4f64(wb[(kk * 4 + lk) * SB + lr], tacc[kt][kk], acc[rt], 0, 0, 0);
            }
        }
        // row tile kt is complete (its last k-tile was kt): out it goes while the later row tiles multiply — a CU drains its 128 KB
        // of stores a tile in ~10 K cycles, which used to follow the product
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int r = kt * SB + lk + 4 * v;
            U[(size_t)(kt * 4 + v) * ld4] = acc[kt][v];
            sum = fma(acc[kt][v], zj[r], sum);
        }
    }
    MBO_STAMP(3);
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    if (lk == 0) yc[ct * SB + lr] -= sum;
    MBO_STAMP(4);
}

__global__ __launch_bounds__(512, 1) void multi_block_kernel(Mats p, OneBlock ob) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int tid = threadIdx.x, b = blockIdx.x;
    const Lane q = lane_of(tid);
    const int wave = tid >> 6, lane = tid & 63, lr = q.lr, lk = q.lk;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int nrb = p.nrb, cs = nrb * NB;  // points per code plane == npad
    double *Ab = p.A + (size_t)b * p.bstride;
    double *S = lds;                        // packed upper block triangle, NBLK x [16][16]
    double *vec = lds + NBLK * SB * SB;     // [2][128] y | upper-half partial sums
    double *red = vec + 2 * NB;             // [8]
    double *ylds = red + 8;                 // [nrb][128] the right-hand side, updated in place
    double *zlds = ylds + nrb * NB;         // [nrb][128] z_j
    double *stage = zlds + nrb * NB;        // [3][16][LDS_LD] A panel stages of the off-diagonal GEMMs (a ring)
    uint32_t *codes = reinterpret_cast<uint32_t *>(stage + 3 * MB_STAGE);  // [nW][cs]
    {
        const uint32_t *lb = p.leafx + (size_t)b * p.nW * cs;
        for (int e = tid; e < p.nW * cs; e += 512) codes[e] = lb[e];
        for (int e = tid; e < cs; e += 512) ylds[e] = e < p.N ? ob.y[e] : 0.0;
    }
    __syncthreads();
    double quad_sum = 0.0, logsum_sum = 0.0;  // over the blocks (thread 0: wave 0 ran factor16)
    int bad_all = 0;
#ifdef BARK_TWO_STAMPS
    int stamp_n = 0;
#define MB_STAMP() do { if (tid == 0 && b == 0) g_mb_stamps[stamp_n++] = __builtin_readcyclecounter(); } while (0)
#else
#define MB_STAMP() do {} while (0)
#endif
    MB_STAMP();
    const GenCtx gctx = gen_ctx(p, b, ob.rep);
    // (a block step per compile-time index: as a run-time loop the body — mb_update, factor_tile8, mb_offdiag — spilled 150 VGPRs)
    auto block = [&](auto JJ) {
        constexpr int j = decltype(JJ)::value;
        if (j >= nrb) return;  // workgroup-uniform
        double logsum = 0.0, pacc = 1.0;
        int bad = 0;
        const double *col0 = Ab + (size_t)j * NB;  // U[0 : 128 j, j-block]
        switch (wave_u) {  // (wave-uniform; every wave runs the same barriers)
            case 0: mb_update<0, 0, 5>(col0, p.ld, j, lds, S, wave_u, lane, lr, lk, p, b, ob.rep, codes, cs, j * NB); break;
            case 1: mb_update<1, 0, 5>(col0, p.ld, j, lds, S, wave_u, lane, lr, lk, p, b, ob.rep, codes, cs, j * NB); break;
            case 2: mb_update<2, 0, 5>(col0, p.ld, j, lds, S, wave_u, lane, lr, lk, p, b, ob.rep, codes, cs, j * NB); break;
            case 3: mb_update<3, 0, 5>(col0, p.ld, j, lds, S, wave_u, lane, lr, lk, p, b, ob.rep, codes, cs, j * NB); break;
            case 4: mb_update<0, 5, 9>(col0, p.ld, j, lds, S, wave_u, lane, lr, lk, p, b, ob.rep, codes, cs, j * NB); break;
            case 5: mb_update<1, 5, 9>(col0, p.ld, j, lds, S, wave_u, lane, lr, lk, p, b, ob.rep, codes, cs, j * NB); break;
            case 6: mb_update<2, 5, 9>(col0, p.ld, j, lds, S, wave_u, lane, lr, lk, p, b, ob.rep, codes, cs, j * NB); break;
            default: mb_update<3, 5, 9>(col0, p.ld, j, lds, S, wave_u, lane, lr, lk, p, b, ob.rep, codes, cs, j * NB); break;
        }
        __syncthreads();
        MB_STAMP();
        const int left = p.N - j * NB;
        const int nsb = left >= NB ? NSB : (left + SB - 1) / SB;  // live sub-blocks of this block; the rest is identity padding
        factor_tile8<true>(S, nsb, false, wave_u, lane, lr, lk, logsum, pacc, bad);
        MB_STAMP();
        // z_j = W_j' y_j ; |z_j|^2 (diag_kernel's last phase on the first four waves)
        {
            const double *yj = ylds + j * NB;
            const int c = tid & (NB - 1), half = tid >> 7, cbk = c >> 4, cc = c & 15;  // (half = 2, 3 do nothing)
            double part = 0.0;
#pragma unroll
            for (int i = 0; i < NSB / 2; ++i) {
                const int rbk = half * (NSB / 2) + i;
                if (rbk <= cbk && half < 2) {
                    const double *col = S + blk_off(rbk, cbk) + cc;
#pragma unroll
                    for (int rr = 0; rr < SB; ++rr) part = fma(col[rr * SB], yj[rbk * SB + rr], part);
                }
            }
            if (half == 1) vec[NB + c] = part;
            __syncthreads();
            double zz = 0.0;
            if (!half) {
                const double z = part + vec[NB + c];
                zlds[j * NB + c] = z;
                zz = z * z;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) zz += __shfl_xor(zz, off);
            if (lane == 0) red[wave] = zz;
        }
        __syncthreads();
        quad_sum += red[0] + red[1];
        logsum_sum += logsum;
        bad_all = (bad && !bad_all) ? j * NB + bad : bad_all;
        MB_STAMP();
#pragma unroll 1
        for (int c = j + 1; c < nrb; ++c) {
            // (no barrier between the tiles of a block row: a tile reads none of its predecessor's results, and its first write to a
            // stage slot comes after this wave passed the barrier that ended the predecessor's GEMM.  The call itself begins with an
            // s_waitcnt vmcnt(0): with the c loop inside the function the stores could drain under the next tile, but that
            // body spills 78 VGPRs)
            mb_offdiag<j>(gctx, p.ld, cs, (int)(stage - lds), (int)(reinterpret_cast<double *>(codes) - lds), Ab, c,
                          (int)(zlds - lds) + j * NB, (int)(ylds - lds) + c * NB, wave_u, lane, lr, lk);
        }
        __syncthreads();  // U[j, :] (global: this workgroup's own stores) and y are visible to every wave
        MB_STAMP();
    };
    block(std::integral_constant<int, 0>{});
    block(std::integral_constant<int, 1>{});
    block(std::integral_constant<int, 2>{});
    block(std::integral_constant<int, 3>{});
    block(std::integral_constant<int, 4>{});
    block(std::integral_constant<int, 5>{});
    if (tid == 0) {  // finish_mll_kernel's arithmetic (quick_inverse.py:38 / mcmc_record_mll.py:73)
        double v = -quad_sum - 2.0 * logsum_sum;
        if (ob.include_2pi) v = v - (double)p.N * log(2.0 * M_PI);
        ob.mll[b] = 0.5 * v;
        p.info[b] = *ob.fault ? -1 : bad_all;
    }
}

// D = P - sum_k U[k,j]'U[k,j] over the nkb block rows above, on the 36 sub-blocks of the upper block triangle, in place in
// the stored diagonal tile: diag_update's arithmetic (per element the same MFMA sequence: block rows in order, k ascending,
// four k per MFMA — identical bits) by 9 workgroups x 4 waves per matrix, one sub-block per wave, operands straight from
// L2 (solve_kernel just wrote the panel), instead of inside diag_kernel, where it is 8-16 us of a one-workgroup kernel that
// sits on the critical path of chain-bound chunks (MFMA-bound on ONE CU: 36 x 32 MFMAs over four pipes).  diag_kernel then
// runs with nkb = 0.  publish: store the block step in p.sync[3] (the helper streams' gates wait for it), as diag_kernel
// does when it is the first kernel after solve(j-1).
__global__ __launch_bounds__(THREADS) void diag_pre_kernel(Mats p, int j, int nkb, int publish) {
    const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
    if (publish && blockIdx.x == 0 && b == 0 && threadIdx.x == 0)
        __hip_atomic_store(p.sync + 3, j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int rb = 0, rem = blockIdx.x * (THREADS / 64) + wave;  // sub-block index 0 .. 35, row-major over the upper block triangle
    while (rem >= NSB - rb) {
        rem -= NSB - rb;
        ++rb;
    }
    const int cb = rb + rem;
    double *Ab = p.A + (size_t)b * p.bstride;
    double *tile = Ab + (size_t)j * NB * p.ld + (size_t)j * NB;
    double pre[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) pre[v] = tile[(size_t)(rb * 16 + lk + 4 * v) * p.ld + cb * 16 + lr];
    f64x4 acc = {0.0, 0.0, 0.0, 0.0};
    for (int a = 0; a < nkb; ++a) {
        const double *panel = Ab + (size_t)(j - nkb + a) * NB * p.ld + (size_t)j * NB + (size_t)lk * p.ld + lr;
#pragma unroll 8
        for (int ks = 0; ks < NB / 4; ++ks) {
            const double av = panel[(size_t)(ks * 4) * p.ld + rb * 16], bv = panel[(size_t)(ks * 4) * p.ld + cb * 16];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) tile[(size_t)(rb * 16 + lk + 4 * v) * p.ld + cb * 16 + lr] = pre[v] - acc[v];
}

// the row stream `slot` has finished everything up to block step `value` (one thread; runs after the row kernels
// of that step in stream order, so their stores are complete and released when it starts)
__global__ void sync_publish_kernel(int32_t *sync, int slot, int value) {
    __hip_atomic_store(sync + slot, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Gate of a helper stream: one lane waits (bounded: ~2 s, then sync[2]++ and on it goes — the call ends with info = -3)
// until the caller's stream has reached block step `value` (diag_kernel publishes it when it starts); the row kernels
// behind the gate in stream order start once it retires.  The diag_kernel it waits for is enqueued AFTER this gate (the host
// runs ahead): progress relies on the caller's stream running beside this one, the bound covers the case that it cannot.
__global__ void sync_gate_kernel(int32_t *sync, int slot, int value) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
    while (__hip_atomic_load(sync + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < value) {
        if (__hip_atomic_load(sync + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) break;  // sticky: see diag_kernel
        __builtin_amdgcn_s_sleep(4);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {
            atomicAdd(sync + 2, 1);
            break;
        }
    }
}

}  // namespace
}  // namespace bark
