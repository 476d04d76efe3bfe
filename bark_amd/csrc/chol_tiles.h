// chol_tiles.h — part of the dense sweep's one translation unit (chol.hip includes it; kernels and helpers live in the
// anonymous namespace of that unit).  The 128 x 128 tile machinery every kernel of the sweep shares: MFMA lane maps, LDS-DMA staging, the k-major
// GEMM loops, the XCD-aware workgroup map, and Mats (the chunk's workspace as the kernels see it).
#pragma once
#include "common.h"

namespace bark {
namespace {

// Tuning builds only (-DBARK_DIAG_STAMPS; tools/ab/diag_stamps.py): cycle stamps of thread 0 of workgroup 0 of the diagonal-block
// kernels.  Without the flag the macro is empty.
#ifdef BARK_DIAG_STAMPS
__device__ unsigned long long g_diag_stamps[64];
#define DIAG_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_diag_stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define DIAG_STAMP(i) do {} while (0)
#endif

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

constexpr int NB = TILE;         // 128
constexpr int BK = 16;           // k rows per LDS stage
constexpr int LDS_LD = NB + 16;  // padded row (doubles)
constexpr int STAGE = 2 * BK * LDS_LD;  // A rows then B rows, doubles
constexpr int GEMM_LDS_DOUBLES = 2 * STAGE;
constexpr int THREADS = 256;

struct Lane {
    int wr, wc, lr, lk;
};

__device__ __forceinline__ Lane lane_of(int tid) {
    Lane q;
    const int wave = tid >> 6, l = tid & 63;
    q.wr = wave >> 1;
    q.wc = wave & 1;
    q.lr = l & 15;
    q.lk = l >> 4;
    return q;
}

// element (row, col) inside the 128x128 tile held by acc[mt][nt][v] of this lane
__device__ __forceinline__ int acc_row(const Lane &q, int mt, int v) { return q.wr * 64 + mt * 16 + q.lk + 4 * v; }
__device__ __forceinline__ int acc_col(const Lane &q, int nt) { return q.wc * 64 + nt * 16 + q.lr; }

__device__ __forceinline__ void zero_acc(f64x4 (&acc)[4][4]) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = (f64x4){0.0, 0.0, 0.0, 0.0};
}

// one k-tile (16 deep) of MFMAs from LDS stage `st`
__device__ __forceinline__ void mma_stage(f64x4 (&acc)[4][4], const double *st, const Lane &q) {
    const double *As = st;
    const double *Bs = st + BK * LDS_LD;
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
        const double *ar = As + (kk * 4 + q.lk) * LDS_LD + q.wr * 64 + q.lr;
        const double *br = Bs + (kk * 4 + q.lk) * LDS_LD + q.wc * 64 + q.lr;
        const double a0 = ar[0], a1 = ar[16], a2 = ar[32], a3 = ar[48];
        const double b0 = br[0], b1 = br[16], b2 = br[32], b3 = br[48];
#define BARK_MFMA(mt, nt, av, bv) acc[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[mt][nt], 0, 0, 0)
        BARK_MFMA(0, 0, a0, b0); BARK_MFMA(0, 1, a0, b1); BARK_MFMA(0, 2, a0, b2); BARK_MFMA(0, 3, a0, b3);
        BARK_MFMA(1, 0, a1, b0); BARK_MFMA(1, 1, a1, b1); BARK_MFMA(1, 2, a1, b2); BARK_MFMA(1, 3, a1, b3);
        BARK_MFMA(2, 0, a2, b0); BARK_MFMA(2, 1, a2, b1); BARK_MFMA(2, 2, a2, b2); BARK_MFMA(2, 3, a2, b3);
        BARK_MFMA(3, 0, a3, b0); BARK_MFMA(3, 1, a3, b1); BARK_MFMA(3, 2, a3, b2); BARK_MFMA(3, 3, a3, b3);
#undef BARK_MFMA
    }
}

// ---- LDS-DMA staging ---------------------------------------------------------------------------
// One wave-instruction (global_load_lds_dwordx4) copies a whole k-row of a panel (128 doubles = 1 KiB,
// lane l supplies the address of its 16 bytes) from L2/HBM straight into the LDS stage image: the
// destination is wave-uniform base + 16*lane, i.e. exactly one padded row As[k][0..127].  No VGPR
// staging and no ds_write_b128 bursts (measured: those bursts, not HBM, cost the register-staged
// pipeline ~12 % of the MFMA rate).  Wave w moves rows w, w+4, w+8, w+12 of both operands.
typedef __attribute__((address_space(3))) void lds_ptr_t;
typedef const __attribute__((address_space(1))) void glb_ptr_t;

__device__ __forceinline__ void dma_row(const double *g, double *l) {
    __builtin_amdgcn_global_load_lds((glb_ptr_t *)g, (lds_ptr_t *)l, 16, 0, 0);
}

__device__ __forceinline__ void stage_dma(const double *__restrict__ A, long lda, const double *__restrict__ B,
                                          long ldb, int kt, double *st, int wave, int lane) {
    const double *a = A + ((long)kt * BK + wave) * lda + lane * 2;
    const double *b = B + ((long)kt * BK + wave) * ldb + lane * 2;
    double *as = st + wave * LDS_LD;
    double *bs = as + BK * LDS_LD;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        dma_row(a + (long)(4 * p) * lda, as + 4 * p * LDS_LD);
        dma_row(b + (long)(4 * p) * ldb, bs + 4 * p * LDS_LD);
    }
}

// acc[r][c] += sum_{k<K} A[k][r] * B[k][c] for a 128x128 tile; A, B k-major panels (row stride lda/ldb, 128 contiguous
// doubles per row, 16-byte aligned), K % 16 == 0.  All 256 threads; ends with a barrier.  Tile t+1 is in flight into
// the other LDS stage while tile t is multiplied; the wait + barrier at the end of the iteration publishes it.
__device__ __forceinline__ void gemm_kmajor_dma(f64x4 (&acc)[4][4], const double *__restrict__ A, long lda,
                                                const double *__restrict__ B, long ldb, int K, double *lds, int tid,
                                                const Lane &q) {
    const int nk = K / BK;
    if (nk == 0) return;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    stage_dma(A, lda, B, ldb, 0, lds, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) stage_dma(A, lda, B, ldb, kt + 1, lds + ((kt + 1) & 1) * STAGE, wave, lane);
        mma_stage(acc, lds + (kt & 1) * STAGE, q);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

// Row-tile map of the triangular solve: W is upper triangular, so output row tile rt (16 rows) only needs
// k-tiles kt <= rt.  Wave-row 0 owns row tiles {0,3,4,7}, wave-row 1 owns {1,2,5,6}: 18 (k-tile, row-tile)
// products each instead of 32, perfectly balanced.

// acc[rt(mt)][nt] += sum_k P[k][r] T[k][c] for the A-operand panel P (row stride 128) = DEF dense 128-row blocks
// followed by the upper triangular W: in the W block the k-tiles above a row tile are skipped.  T: (DEF + 1) * 128
// k-rows, row stride ldt.  Same LDS-DMA staging/pipeline as gemm_kmajor_dma.
template <int DEF>
__device__ __forceinline__ void gemm_upper_tri(f64x4 (&acc)[4][4], const int (&rt)[4], const double *__restrict__ P,
                                               const double *__restrict__ T, long ldt, double *lds, int tid,
                                               const Lane &q) {
    constexpr int nd = DEF * (NB / BK), nk = nd + NB / BK;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    stage_dma(P, NB, T, ldt, 0, lds, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nd; ++kt) {  // dense block(s): every row tile takes every k-tile
        stage_dma(P, NB, T, ldt, kt + 1, lds + ((kt + 1) & 1) * STAGE, wave, lane);
        const double *As = lds + (kt & 1) * STAGE;
        const double *Bs = As + BK * LDS_LD;
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            const double *br = Bs + (kk * 4 + q.lk) * LDS_LD + q.wc * 64 + q.lr;
            const double b0 = br[0], b1 = br[16], b2 = br[32], b3 = br[48];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const double a = As[(kk * 4 + q.lk) * LDS_LD + rt[mt] * 16 + q.lr];
                acc[mt][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, acc[mt][0], 0, 0, 0);
                acc[mt][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, acc[mt][1], 0, 0, 0);
                acc[mt][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b2, acc[mt][2], 0, 0, 0);
                acc[mt][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b3, acc[mt][3], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#pragma unroll
    for (int kw = 0; kw < NB / BK; ++kw) {  // k-tile kw of W (nd is even: stage parity == kw & 1)
        const int kt = nd + kw;
        if (kt + 1 < nk) stage_dma(P, NB, T, ldt, kt + 1, lds + ((kw + 1) & 1) * STAGE, wave, lane);
        const double *As = lds + (kw & 1) * STAGE;
        const double *Bs = As + BK * LDS_LD;
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            const double *br = Bs + (kk * 4 + q.lk) * LDS_LD + q.wc * 64 + q.lr;
            const double b0 = br[0], b1 = br[16], b2 = br[32], b3 = br[48];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                if (kw <= rt[mt]) {  // wave-uniform
                    const double a = As[(kk * 4 + q.lk) * LDS_LD + rt[mt] * 16 + q.lr];
                    acc[mt][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0, acc[mt][0], 0, 0, 0);
                    acc[mt][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b1, acc[mt][1], 0, 0, 0);
                    acc[mt][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b2, acc[mt][2], 0, 0, 0);
                    acc[mt][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b3, acc[mt][3], 0, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

// Workgroup -> (matrix, tile) map for the panel/solve grids.  Blocks are dealt round-robin over the 8
// XCDs (id % 8 labels the XCD group; speed only, never correctness), and every tile of block row j of
// one matrix streams the same A panel U[0:128j, j]: give all tiles of matrix b ids == b (mod 8), in one
// contiguous run of the per-XCD sequence, so that panel is fetched into ONE 4 MiB L2 once and shared.
// With fewer than 8 resident matrices that would leave XCDs idle, so each matrix is split into
// R "virtual matrices" holding every R-th tile; virtual matrix v goes to XCD v % 8 (R: xcd_rep — Bc R is a multiple of 8).
// Grid = 8 * ceil(Bc R / 8) * ceil(ntiles / R); ids that fall outside exit.
constexpr int NXCD = 8;
// ... and so would a chunk size that is not a multiple of 8 while it is small (12 matrices: four XCDs with two, four with one —
// the launch took as long as 16: N = 4096 x 9 / 12 / 16 ran 6.1 / 7.0 / 7.2 ms): then R = 8 / gcd(Bc, 8) virtual matrices per
// matrix make Bc R a multiple of 8 (4.9 / 5.9 / 7.2 ms).  From 10 % imbalance down (Bc > 72) the locality of one matrix per XCD
// is worth more.  The same for 3, 5, 6, 7 matrices, which used to get ceil(8 / Bc) virtual matrices each — 9, 10, 12, 14 on 8 XCDs
// (N = 8192 x 3 14.2 -> 10.5 ms, N = 4096 x 5 / 6 4.10 / 4.33 -> 3.44 / 3.59).
__host__ __device__ __forceinline__ int xcd_rep(int Bc) {
    const int rounds = (Bc + NXCD - 1) / NXCD;
    if (Bc % NXCD == 0 || (Bc > NXCD && rounds * NXCD * 10 < Bc * 11)) return 1;
    return (Bc % 4 == 0) ? 2 : (Bc % 2 == 0) ? 4 : 8;  // 8 / gcd(Bc, 8); fewer than 8 matrices: 3, 5, 6, 7 of them used to get ceil(8 / Bc)
}
__host__ __device__ __forceinline__ bool xcd_map(int id, int ntiles, int Bc, int &b, int &tile) {
    const int R = xcd_rep(Bc), ntv = (ntiles + R - 1) / R;
    const int x = id % NXCD, q = id / NXCD;
    const int v = (q / ntv) * NXCD + x;  // virtual matrix
    b = v / R;
    tile = (q % ntv) * R + (v - b * R);
    return b < Bc && tile < ntiles;
}
__host__ __device__ inline unsigned xcd_grid(int ntiles, int Bc) {
    const int R = xcd_rep(Bc), ntv = (ntiles + R - 1) / R;
    return (unsigned)(NXCD * ((Bc * R + NXCD - 1) / NXCD) * ntv);
}

struct Mats {
    double *A;            // (Bc, Npad, ld)
    long ld, bstride;
    double *W;            // (Bc, 256, 128)  rows 128..255: W_j = inverse of the current diagonal factor; rows 0..127:
                          //                 -U[j-1,j] W_j (diag_kernel's epilogue; pipelined schedule only)
    double *yz;           // (Bc, Npad)      y on entry, z = U^-T y on exit
    double *accum;        // (Bc, 2)         quad, logdet
    int32_t *info;        // (Bc,)
    int nrb;              // row blocks  (Npad / 128)
    int ncb;              // column blocks incl. candidate blocks
    int Bc;               // matrices in this chunk
    // fused Gram generation (MLL-only path): tiles of A = [scale*] K + (1e-6+noise) I are produced in the
    // panel epilogue from the byte-packed leaf ids instead of being read back from HBM
    const uint32_t *leafx;  // (Bc, W, npad) or nullptr when A is materialised
    const double *scale;    // (Bc,) or nullptr
    const double *shift;    // (Bc,) or nullptr (no-null kernel)
    const double *noise;    // (Bc,)
    int nW, m, N;  // dwords of leaf ids per point, trees, real points
    // device-side hand-over of row-launch completion to the caller's stream (chain-bound schedules, see Sweep):
    // sync[0], sync[1] = progress counters of the two row streams, sync[2] = timed-out waits, sync[3] = progress of
    // the caller's stream (diag_kernel(j) stores j + 1 when it starts: solve(j-1) has retired).  Zeroed per chunk.
    int32_t *sync;
};

constexpr size_t W_STRIDE = (size_t)2 * NB * NB;  // doubles per matrix in Mats::W
__device__ __forceinline__ double *w_block(const Mats &p, int b) { return p.W + (size_t)b * W_STRIDE + (size_t)NB * NB; }

}  // namespace
}  // namespace bark
