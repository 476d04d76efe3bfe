// Batched GP marginal log-likelihood + posterior on gfx950.
//
// Reference semantics (paths under /root/reference):
//   examples/mcmc/mcmc_record_mll.py:57-74      K_s = K + (1e-6+noise) I ; 0.5(-y'K_s^-1 y - log|K_s| - n log 2pi)
//   src/bark/fitting/bark_sampler.py:153-162    K_s = scale K + (1e-6+noise) I ; quick_inverse.mll (:37-38)
//   src/bark/tree_kernels/tree_gps.py:80-113    mu = K_xX K_s^-1 y ; var = scale - diag(K_xX K_s^-1 K_Xx)
// The reference does LU `inv` + LU `slogdet` per forest; here every forest sample of a chunk is
// factorised as K_s = U'U (U upper triangular, row-major) by a LEFT-LOOKING blocked Cholesky
// whose kernels advance all Bc matrices of the chunk in lock step (block size 128).  Block column j:
//
//   diag_kernel  (1 workgroup / matrix)   D = P_jj - U[j-1,j]'U[j-1,j]; U_jj = chol(D); W_j = U_jj^-1;
//                                         z_j = W_j' y_j; logdet += 2 sum log diag; quad += |z_j|^2
//        ||  (concurrently, helper stream)
//   row_kernel   (1 workgroup / 128x128 tile) T[j,i] = A[j,i] - sum_{k<j} U[k,j]' U[k,i]   (fp64 MFMA, K = 128 j)
//                                         plus the partial diagonal tile P_{j+1,j+1} (same sum, k < j).
//                                         Panels are staged by LDS-DMA; in MLL-only sweeps A[j,i] is generated
//                                         in the epilogue from the leaf codes (the Gram is never materialised).
//                                         Under-filled steps use panel_split_kernel + panel_reduce_kernel (split-K).
//        then
//   solve_kernel (1 workgroup / tile)     U[j,i] = W_j' T[j,i]  (MFMA, K = 128, zero k-tiles of W_j skipped);
//                                         y_i -= U[j,i]' z_j
//
// The candidate block of the posterior (K_Xx, N x C) — or an identity block for the explicit inverse — is
// appended as extra block columns, so the same sweep yields V = U^-T K_Xx; then mu = V'z,
// var = scale - colsumsq(V), and optionally the full covariance / K_s^-1 = V'V (vtv_kernel).
//
// Three schedules of those kernels (Sweep, below), chosen per chunk of resident matrices:
//   plain       diag(j) || row(j), then solve(j).  Chunks whose size is a multiple of the 256 CUs (every round of row
//               workgroups full) and matrices of fewer than 8 block rows.
//   pipelined   row(j) covers k < j-1 only and is launched two steps ahead on the helper streams; the last block row is
//               applied by its consumers: solve_kernel<1> (one K = 256 product with the stacked [-U[j-1,j] W_j ; W_j])
//               and diag_kernel (two block rows, and the -U[j-1,j] W_j block in its epilogue).  Under-filled launches
//               split K.  Everything else that is not bound by its critical path.
//   look-ahead  split-K layout (A materialised, slab scratch): the bulk of step j+2's K range (split into slabs) is
//               launched after solve(j); diag(j) || the last block row's slab, reduce, solve(j) remain on the critical
//               path.  Few small matrices (one N = 4096 matrix: 2.5 ms).
//
// Why left-looking: every U tile is written once and each trailing tile is accumulated in
// registers over the whole K range, instead of a read-modify-write of the trailing matrix per
// step.  Why lock step over the batch: the serial 128x128 potrf/inverse of one matrix occupies
// one CU; with Bc >= 256 matrices resident in the 288 GB of HBM all CUs do it at once, and the
// MFMA panel kernel always has Bc x (tiles per block row) workgroups.
//
// MFMA: v_mfma_f64_16x16x4_f64.  Operand maps (lane l): A[i = l&15][k = l>>4], B[k = l>>4][j = l&15],
// D reg v: row = (l>>4) + 4v, col = l&15.  Both GEMM operands are "k-major" panels of U (rows = k,
// 128 contiguous columns), so a panel row is one 1 KiB coalesced wave load and the LDS image
// As[k][128(+16 pad)] is read conflict-free by ds_read_b64 (row stride 1152 B == 128 mod 256).
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <vector>

#include "common.h"

namespace bark {

int walk_one_hot(const void *packed, const bark_pack_info *info, const double *X, int64_t N, int64_t d, int words,
                 uint32_t *out, int32_t *fault, hipStream_t stream);
int walk_codes(const void *packed, const bark_pack_info *info, const double *X, int64_t N, int64_t d, uint32_t *out,
               int32_t *fault, hipStream_t stream);
int leafspace_prepare(const uint32_t *codes, int W, int npad, unsigned long long *planes, int R, int Rpad,
                      const double *noise, const double *scale, int m, int bc, double *A, long ld, long bstride,
                      const double *y, int N, double *yz, double *accum, int32_t *info, hipStream_t s);
int leafspace_sumsq(const double *y, int N, double *out, hipStream_t s);
int leafspace_predict(const uint32_t *ccodes, int W, int cpad, int C, const double *w, const double *Minv, int R,
                      const double *noise, const double *scale, int m, int bc, double *mu, double *var, hipStream_t s);
int leafspace_inverse(const uint32_t *codes, int W, int npad, int N, const double *Minv, const double *w, int R,
                      const double *y, const double *noise, const double *scale, int m, int bc, double *Wm, double *kinv,
                      double *kinv_y, hipStream_t s);
int leafspace_finish(const double *accum, const double *yy, const double *noise, const double *scale, int m, int bc, int N,
                     int include_2pi, double *mll, hipStream_t s);

int launch_gram(const uint32_t *leaf1, int npad1, const uint32_t *leaf2, int npad2, int64_t B, int64_t m, int N, int M,
                int Nout, int Mout, const double *shift, const double *scale, const double *noise, double *out, int64_t ld,
                int64_t batch_stride, bool pad_identity, bool upper_only, int rep, int words, hipStream_t stream);

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
__device__ __forceinline__ void factor16(double *blk, int lane, int base_index, double &pacc, int &bad) {
    const int c = lane & 15, g = lane >> 4;
    f64x4 e, f;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        e[v] = blk[(g + 4 * v) * SB + c];
        f[v] = (g + 4 * v == c) ? 1.0 : 0.0;
    }
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
        double val;
        if (gi[v] < g.N && gj < g.N) {
            const int agree = g.rep == REP_BITS ? (int)cnt[v] : g.m - (int)cnt[v];
            val = g.inv_m * (double)agree;
            if (g.has_shift) val = val - g.sh;
            if (g.has_scale) val = g.sc * val;
            if (gi[v] == gj) val = val + g.jitter;
        } else {
            val = gi[v] == gj ? 1.0 : 0.0;  // identity padding
        }
        out[v] = val;
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
        const uint32_t *lb = p.leafx + (size_t)b * p.nW * NB;  // npad == 128
        for (int e = tid; e < p.nW * NB; e += NW * 64) codes[e] = lb[e];
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
        if (wsel == 0)
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
        if (wave_u == 0)
            diag_update<0, true>(nullptr, p.ld, Ab + NB, blk, lds, S, lane, lr, lk, p, b, ob.rep, codes, CS, blk * NB);
        else if (wave_u == 1)
            diag_update<1, true>(nullptr, p.ld, Ab + NB, blk, lds, S, lane, lr, lk, p, b, ob.rep, codes, CS, blk * NB);
        else if (wave_u == 2)
            diag_update<2, true>(nullptr, p.ld, Ab + NB, blk, lds, S, lane, lr, lk, p, b, ob.rep, codes, CS, blk * NB);
        else if (NW == 4 || wave_u == 3)
            diag_update<3, true>(nullptr, p.ld, Ab + NB, blk, lds, S, lane, lr, lk, p, b, ob.rep, codes, CS, blk * NB);
        else  // waves 4-7: the barriers of the update's product loop (diag_update: 1 + 8 for the one block row of the second block)
            for (int a = blk * (1 + NB / BK); a > 0; --a) __syncthreads();
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

constexpr int RED_ROWS = 8;  // rows of a 128 x 128 tile per panel_reduce_kernel workgroup (4 doubles per thread)
// T = A - sum of the tile's S slabs (fixed order: bit-reproducible) for the n_tiles tiles from t_off on.
// GEN == 0: A is read from (and T written to) the materialised matrix; GEN == 1 + LeafRep: A is generated from the
// leaf codes exactly as form_tile does (MLL-only sweeps never materialise the Gram matrix).
template <int GEN>
__global__ __launch_bounds__(THREADS) void panel_reduce_kernel(Mats p, int j, int n_right, int t_off, int n_tiles, int S,
                                                               const double *slabs) {
    const int tl = blockIdx.x / (NB / RED_ROWS), part = blockIdx.x % (NB / RED_ROWS), b = blockIdx.y;
    const int t = t_off + tl;
    const int rb = t < n_right ? j : j + 1, cb = t < n_right ? j + 1 + t : j + 1;
    const int e = part * RED_ROWS * NB + 4 * threadIdx.x;  // four consecutive entries of one tile row
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
        return;
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
    const double inv_m = 1.0 / (double)p.m;
    const bool has_scale = p.scale != nullptr, has_shift = p.shift != nullptr;
    const double sc = has_scale ? p.scale[b] : 1.0, sh = has_shift ? p.shift[b] : 0.0, jitter = 1e-6 + p.noise[b];
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

constexpr int SPLITK_SLOTS = 512;  // workgroup slots split-K aims to fill (2 per CU)
#ifndef BARK_SPLITK_MAX
#define BARK_SPLITK_MAX 32
#endif
constexpr int SPLITK_MAX = BARK_SPLITK_MAX;
#ifndef BARK_LA_SLOTS
#define BARK_LA_SLOTS 448
#endif
#ifndef BARK_LA_SLOTS_CHAIN
#define BARK_LA_SLOTS_CHAIN 384
#endif
#ifndef BARK_LA_BULK_WORK
#define BARK_LA_BULK_WORK 2400
#endif
// Workgroups the look-ahead bulk of a split-K step aims at (nearest integer split factor, never more than
// SPLITK_SLOTS).  The bulk runs beside the previous step's critical-path kernels, and resident workgroups are never
// pre-empted: a bulk that fills every slot for its ~250 us starves them (solve_kernel 20 -> 200 us in the kernel
// timeline of one N = 16384 matrix) and delays its own successor.  So an eighth of the slots stays free — and a quarter
// in steps whose bulk is shorter than the critical path anyway ((tiles x matrices) x block rows < LA_BULK_WORK, ~150 us
// of MFMA work).  Measured (448 | 384 vs 512 everywhere): N = 16384 lone 30.2 -> 28.7 ms, N = 8192 x 2 10.7 -> 9.4,
// N = 4096 x 8 5.59 -> 5.2.
constexpr int LA_SLOTS = BARK_LA_SLOTS, LA_SLOTS_CHAIN = BARK_LA_SLOTS_CHAIN;
// ... of an under-filled bulk launch of the pipelined schedule (two are in flight, and the chain kernels need slots beside
// them): same box, 384 | 448 | 320 | 512 — N = 4096 x 8 4.29 | 4.42 | 4.30 | 4.56 ms, x 16 6.98 | 7.07 | 7.11 | 7.13,
// N = 8192 x 2 7.99 | 8.25 | 8.04 | 8.46, N = 16384 x 1 25.4 | 25.6 | 26.4 | 25.9
#ifndef BARK_PIPE_BULK_SLOTS
#define BARK_PIPE_BULK_SLOTS 384
#endif
constexpr int PIPE_BULK_SLOTS = BARK_PIPE_BULK_SLOTS;
constexpr long LA_BULK_WORK = BARK_LA_BULK_WORK;
// slabs (128 x 128) of one set: a split step has fewer than SPLITK_SLOTS / 2 tiles x matrices, each with S slabs
// (tiles x matrices x S <= the slots aimed at) plus one for the last block row
constexpr size_t SLAB_SET_TILES = (LA_SLOTS > SPLITK_SLOTS ? LA_SLOTS : SPLITK_SLOTS) + SPLITK_SLOTS / 2;
// look-ahead from this much bulk work on ((tiles x matrices) x block rows; see Sweep::lookahead).  Re-measured with this round's
// shorter chain, same box, 450 | 600: N = 4096 x 2 2.28 | 2.45 ms, N = 6000 x 1 3.27 | 3.64, N = 5600 x 1 3.07 | 3.18, nothing
// else moves; from 200 down the lone N = 4096 matrix and N = 2048 x 4 get look-ahead steps and lose 7-15 %
#ifndef BARK_LA_MIN_WORK
#define BARK_LA_MIN_WORK 450
#endif
constexpr long LA_MIN_WORK = BARK_LA_MIN_WORK;
// Critical-path split (the plain split of an under-filled step: lone and few matrices): up to SPLIT_FINE slabs per block row of the
// K range, while the launch stays within SPLIT_FINE_MAX_WGS workgroups.  A split workgroup on an idle chip is a chain of DMA round
// trips (~2.7 us per 16-row k-tile, 8 per block row: 22 us for the one block row of a slab), and since round 5's eight-wave
// diag_kernel that chain — not the diagonal factor beside it — bounds the block step of a lone matrix of a few block rows; halving
// it costs a second slab per block row in the reduce launch.  Round 5, same box, one process per variant, ms (1 | 2 slabs per block
// row up to 256 workgroups | ... up to 384 | 3 slabs up to 256): N = 1024 x 1 0.371 | 0.358 | 0.357 | 0.359, N = 2048 x 1 0.762 | 0.693 |
// 0.695 | 0.713, N = 1500 x 1 0.568 | 0.536 | 0.537 | 0.536, N = 3000 x 1 1.170 | 1.128 | 1.156 | 1.147, N = 4096 x 1 1.696 | 1.668 | 1.694 | 1.682,
// N = 1024 x 4 0.391 | 0.368 | 0.370 | 0.368, x 8 0.420 | 0.398 | 0.403 | 0.413, N = 2048 x 4 0.923 | 0.913 | 0.923 | 0.923; without the
// workgroup bound (4 slabs, or 2 everywhere) N = 4096 x 1 lost 8 % and N = 2048 x 4 5 % to the longer reduce
// (profiles/r05/chain_bound_levers.txt item 4; round 4 had measured the finer slabs beside the four-wave diag_kernel: nothing).
#ifndef BARK_SPLIT_FINE
#define BARK_SPLIT_FINE 2
#endif
constexpr int SPLIT_FINE = BARK_SPLIT_FINE;
#ifndef BARK_SPLIT_FINE_MAX_WGS
#define BARK_SPLIT_FINE_MAX_WGS 256
#endif
constexpr long SPLIT_FINE_MAX_WGS = BARK_SPLIT_FINE_MAX_WGS;  // ... while the launch stays within this many workgroups (more slabs cost more in the reduce launch than they save)
#ifndef BARK_TAIL_MAX_WGS
#define BARK_TAIL_MAX_WGS 192
#endif
constexpr long TAIL_MAX_WGS = BARK_TAIL_MAX_WGS;  // ragged_tail: largest last round (workgroups) that is split over K
#ifndef BARK_SPLITK_LAYOUT_MAX_TILES
#define BARK_SPLITK_LAYOUT_MAX_TILES 600
#endif
#ifndef BARK_SPLITK_LAYOUT_MAX_WORK
#define BARK_SPLITK_LAYOUT_MAX_WORK 3000
#endif
// Which chunks get the split-K layout (slab scratch reserved, A materialised, look-ahead schedule of Sweep::step) and
// which the pipelined schedule (Sweep::step_pipelined, which splits K in its under-filled launches too).  The
// look-ahead schedule has the shorter critical path per block step (diag + solve against a longer diag + a K = 256 solve),
// the pipelined one keeps the chip full; so the split-K layout is for sweeps bound by their critical path: fewer than
// MAX_WORK (matrices x block columns) x block rows.  Build-time tuning constants; split-K layout | pipelined, ms, with
// the narrow solve kernels in both:
//   N = 4096:  B = 2 2.81 | 3.12, B = 3 4.00 | 3.49, B = 4 3.55 | 3.49, B = 5 5.31 | 4.29, B = 8 4.89 | 4.57, B = 16 9.1 | 7.2
//   N = 8192:  B = 1 6.80 | 6.65, B = 2 9.08 | 8.41     N = 5000: B = 2 4.00 | 4.10, B = 3 6.17 | 5.12
//   N = 3000:  B = 4 2.28 | 2.40, B = 6 3.08 | 2.70     N = 2048: B = 8 1.47 | 1.54, B = 12 1.97 | 1.80, B = 16 1.93 | 1.87
//   N = 1024:  B = 32 0.76 | 0.87, B = 40 0.87 | 0.87, B = 64 1.11 | 1.14
// i.e. the crossover sits near 3000 for every N (it was 5600 before the narrow solves shortened the pipelined chain).
// Below 8 block rows (no pipelining) the rule is the tile count alone.
constexpr int SPLITK_LAYOUT_MAX_TILES = BARK_SPLITK_LAYOUT_MAX_TILES;
constexpr int64_t SPLITK_LAYOUT_MAX_WORK = BARK_SPLITK_LAYOUT_MAX_WORK;
// Build-time tuning constants of the pipelined schedule (numbers only: every on/off alternative that was measured and
// lost is gone from the sources, with its figures left in the comment next to the code that won).
#ifndef BARK_DIAG_WAVES8
#define BARK_DIAG_WAVES8 1  // chain-bound diag launches (a CU per matrix) with eight waves: factor_tile8
#endif
constexpr int DIAG8_MAX_BC = 256;  // one-launch kernels (N <= 256): eight waves up to this many matrices per chunk
#ifndef BARK_TWO_BLOCK
#define BARK_TWO_BLOCK 1  // 128 < N <= 256, MLL only: the one-launch evaluation by two_block_kernel (0: the multi-launch sweep)
#endif
// ... for chunks of TWO_MIN_BC .. TWO_MAX_BC matrices, and larger chunks up to TWO_ANY_BC_MAX_N points (plan_chunk has the table)
constexpr int TWO_MIN_BC = 16, TWO_MAX_BC = 384, TWO_ANY_BC_MAX_N = 224;
#ifndef BARK_PIPE_MIN_NRB
#define BARK_PIPE_MIN_NRB 8
#endif
constexpr int PIPE_MIN_NRB = BARK_PIPE_MIN_NRB;  // fewer block rows: plain schedule (no difference measured at N = 512..896)
#ifndef BARK_PLAIN_CHUNK_MULTIPLE
#define BARK_PLAIN_CHUNK_MULTIPLE 256
#endif
constexpr int PLAIN_CHUNK_MULTIPLE = BARK_PLAIN_CHUNK_MULTIPLE;  // chunks of a multiple of this many matrices (and >= PLAIN_MIN_NRB block rows): plain
// round 4 (diagonal tile as a SYRK after the square tiles, in both schedules) — plain | pipelined, ms, same box: N = 2200 x 256
// 18.21 | 18.44, N = 3000 x 256 40.70 | 41.10, N = 4096 x 256 92.45 | 93.05, N = 1536 x 256 (12 block rows) 6.29 | 6.24,
// N = 1536 x 512 11.97 | 12.24, N = 1100 x 256 3.12 | 3.13.  The boundary itself — chunks of EXACTLY 16 block rows (N = 1921..2048)
// take the plain / paired schedule (nrb >= PLAIN_MIN_NRB), 15 block rows and fewer the pipelined one — measured in round 5 with the
// paired launches, paired | pipelined (-DBARK_PLAIN_MIN_NRB=17), same box: N = 2048 x 256 13.31 | 13.36, N = 1930 x 256 13.26 | 13.27,
// N = 2048 x 512 26.00 | 26.24 (profiles/r05/nrb16_boundary_ab.txt); parity at that shape: tests/test_gpu_configs.py::
// test_sixteen_block_rows_b256_is_the_boundary_of_the_paired_schedule
#ifndef BARK_PLAIN_MIN_NRB
#define BARK_PLAIN_MIN_NRB 16
#endif
#ifndef BARK_PIPE_SYRK_MODE
#define BARK_PIPE_SYRK_MODE 2  // pipelined schedule's row launches: 1 = SYRK workgroup in its matrix's run of tiles, 2 = after all square tiles
#endif
#ifndef BARK_PLAIN_PAIRS
#define BARK_PLAIN_PAIRS 1  // lock-step chunks of the plain schedule: two block rows per row launch (Sweep::step_paired)
#endif
#ifndef BARK_PLAIN_SYRK_MODE
#define BARK_PLAIN_SYRK_MODE 2  // 0: square diagonal tile; 2: SYRK workgroups after all square tiles (launch_rows)
#endif
constexpr int PLAIN_MIN_NRB = BARK_PLAIN_MIN_NRB;
#ifndef BARK_SOLVE_NARROW_MAX_WGS
#define BARK_SOLVE_NARROW_MAX_WGS 256
#endif
#ifndef BARK_PIPE_NARROW_MAX_WGS
#define BARK_PIPE_NARROW_MAX_WGS 256
#endif
// a solve of at most this many workgroups after sharing its tiles' columns out goes narrow (look-ahead / pipelined schedule)
constexpr long SOLVE_NARROW_MAX_WGS = BARK_SOLVE_NARROW_MAX_WGS, PIPE_NARROW_MAX_WGS = BARK_PIPE_NARROW_MAX_WGS;
#ifndef BARK_SOLVE_DIRECT_MAX_WGS
#define BARK_SOLVE_DIRECT_MAX_WGS 512
#endif
constexpr long SOLVE_DIRECT_MAX_WGS = BARK_SOLVE_DIRECT_MAX_WGS;  // ... at most this many after sharing out over 8: solve_direct_kernel

#ifndef BARK_DEVWAIT_MAX_BC
#define BARK_DEVWAIT_MAX_BC 32
#endif
// chunks of at most this many matrices hand row-launch completion over to the caller's stream by a device-side counter
// (diag_kernel waits at its end) instead of an event: their block steps are bound by diag -> solve -> diag, and the
// waiting workgroups (one per matrix) hold few slots
constexpr int DEVWAIT_MAX_BC = BARK_DEVWAIT_MAX_BC;

struct Layout {
    int64_t npad, cpad, ncols, ld, W;
    bool splitk;  // chunk too small to fill the chip with one workgroup per tile: slab scratch reserved
    size_t off_A, off_W, off_yz, off_acc, off_sync, off_leafx, off_leafc, off_slab, total;
};

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

Layout make_layout(int64_t N, int64_t C, int64_t m, int64_t Bc) {
    Layout L;
    L.npad = round_up(N, NB);
    L.cpad = C > 0 ? round_up(C, NB) : 0;
    L.ncols = L.npad + L.cpad;
    L.ld = L.ncols + 16;  // +128 B: consecutive rows of a tile do not alias the same HBM channel set
    L.W = MAX_LEAF_WORDS;  // leaf-code planes are sized for the widest code any forest can need
    size_t o = 0;
    L.off_A = o;
    o = align256(o + (size_t)Bc * L.npad * L.ld * sizeof(double));
    L.off_W = o;
    o = align256(o + (size_t)Bc * W_STRIDE * sizeof(double));
    L.off_yz = o;
    o = align256(o + (size_t)Bc * L.npad * sizeof(double));
    L.off_acc = o;
    o = align256(o + (size_t)Bc * 2 * sizeof(double));
    L.off_sync = o;
    o = align256(o + 64);
    L.off_leafx = o;
    o = align256(o + (size_t)Bc * L.W * L.npad * sizeof(uint32_t));
    L.off_leafc = o;
    o = align256(o + (size_t)Bc * L.W * L.cpad * sizeof(uint32_t));
    L.off_slab = o;
    {
        const int64_t nrb = L.npad / NB, tiles = Bc * (L.ncols / NB);
        L.splitk = nrb >= 4 && (nrb < PIPE_MIN_NRB ? tiles < SPLITK_LAYOUT_MAX_TILES : tiles * nrb < SPLITK_LAYOUT_MAX_WORK);
    }
    // two slab sets (look-ahead: the bulk of step j+1 is accumulated while step j is reduced); a split step has fewer
    // than SPLITK_SLOTS / 2 tiles x matrices, S of at most SPLITK_SLOTS / that, plus one slab for the last block row
    o = align256(o + (size_t)2 * SLAB_SET_TILES * NB * NB * sizeof(double));  // every chunk: ragged last rounds split K too
    L.total = o;
    return L;
}

// (74.3 KiB: two of these workgroups fit a CU — chunks of more than 256 small matrices — or one beside a row workgroup; the four
// per-wave transpose scratch blocks that used to sit between the factor image and `vec` went with x_entry)
constexpr size_t DIAG_LDS = (size_t)(NBLK * SB * SB + 2 * NB + 8) * sizeof(double);
// With few matrices resident the diag workgroup IS the critical path of the sweep, and a split-K / row workgroup that
// lands on its CU stretches it from 52 to 62-77 us (kernel timeline of a lone N = 4096 matrix).  Asking for the whole
// 160 KiB of LDS keeps the CU to itself (lone N = 4096: 2.83 -> 2.59 ms; neutral from 8 matrices on, harmful at 64:
// 26.7 -> 27.5 ms); with many matrices the kernel is hidden and must share (DIAG_LDS).
#ifndef BARK_DIAG_EXCLUSIVE_MAX_BC
#define BARK_DIAG_EXCLUSIVE_MAX_BC 16
#endif
constexpr size_t DIAG_LDS_EXCLUSIVE = 160 * 1024;
constexpr int DIAG_EXCLUSIVE_MAX_BC = BARK_DIAG_EXCLUSIVE_MAX_BC;
constexpr size_t GEMM_LDS = (size_t)GEMM_LDS_DOUBLES * sizeof(double);
static_assert(DIAG_LDS >= GEMM_LDS, "diag kernel reuses its LDS for the K=128 GEMM stage");

// The diag kernel runs concurrently with row launches (Sweep::step; speed only, every kernel's inputs are ordered by
// events): diag stays on the caller's stream (its 82 KiB of LDS leave room for a row workgroup beside it on a CU) and
// the bulk row launches go to a helper stream.  The helper stream and the events are created once and reused; the
// pattern is fork/join, so it is capturable.
// (Tried and rejected: splitting the resident matrices into two independently advancing lanes so that
// one lane's panel kernel covers the other's diag/solve phases — 5 % slower at B = 256, 4 % at B = 64.)
}  // namespace

int set_lds_limits() {
    // kernels using more than 64 KiB of dynamic LDS need the limit raised once per device
    static std::once_flag once[64];
    static int status[64];
    int dev = 0;
    BARK_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(BARK_ERR_ARG, "device index %d out of range", dev);
    std::call_once(once[dev], [dev]() {
        auto set = [](const void *fn, size_t bytes) {
            return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        };
        hipError_t e = set(reinterpret_cast<const void *>(diag_kernel<false, 4>), DIAG_LDS_EXCLUSIVE);
        if (e == hipSuccess) e = set(reinterpret_cast<const void *>(diag_kernel<true, 4>), DIAG_LDS_EXCLUSIVE);
        if (e == hipSuccess) e = set(reinterpret_cast<const void *>(two_block_kernel<4>), DIAG_LDS_EXCLUSIVE);
        if (e == hipSuccess) e = set(reinterpret_cast<const void *>(two_block_kernel<8>), DIAG_LDS_EXCLUSIVE);
        if (e == hipSuccess) e = set(reinterpret_cast<const void *>(diag_kernel<false, 8>), DIAG_LDS_EXCLUSIVE);
        if (e == hipSuccess) e = set(reinterpret_cast<const void *>(diag_kernel<true, 8>), DIAG_LDS_EXCLUSIVE);
        if (e == hipSuccess) e = set(reinterpret_cast<const void *>(row_kernel<0>), GEMM_LDS);
        if (e == hipSuccess) e = set(reinterpret_cast<const void *>(row_kernel<1>), GEMM_LDS);
        if (e == hipSuccess) e = set(reinterpret_cast<const void *>(row_kernel<2>), GEMM_LDS);
        if (e == hipSuccess) e = set(reinterpret_cast<const void *>(row_kernel<3>), GEMM_LDS);
        if (e == hipSuccess) e = set(reinterpret_cast<const void *>(panel_split_kernel), GEMM_LDS);
        if (e == hipSuccess) e = set(reinterpret_cast<const void *>(vtv_kernel), GEMM_LDS);
        if (e == hipSuccess) e = set(reinterpret_cast<const void *>(solve_kernel<0>), GEMM_LDS);
        if (e == hipSuccess) e = set(reinterpret_cast<const void *>(solve_kernel<1>), GEMM_LDS);
        if (e == hipSuccess) e = set(reinterpret_cast<const void *>(solve_narrow_kernel<1>), GEMM_LDS);
        if (e == hipSuccess) e = set(reinterpret_cast<const void *>(solve_narrow_kernel<2>), GEMM_LDS);
        if (e == hipSuccess) e = set(reinterpret_cast<const void *>(solve_narrow_kernel<1, 1>), GEMM_LDS);
        if (e == hipSuccess) e = set(reinterpret_cast<const void *>(solve_narrow_kernel<2, 1>), GEMM_LDS);
        status[dev] = (int)e;
    });
    if (status[dev] != 0)
        return fail(BARK_ERR_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed: %s",
                    hipGetErrorString((hipError_t)status[dev]));
    return BARK_OK;
}

namespace {

// ---------------------------------------------------------------------------------------------
// Sweep: the factorisation of one chunk of resident matrices, shared by the dense MLL / posterior entry
// point and the leaf-space entry point.  The caller fills the matrices (and right-hand sides), then calls
// step(j) for every block column.
// ---------------------------------------------------------------------------------------------
struct Sweep {
    Mats p;
    hipStream_t main = nullptr, panel = nullptr;  // panel == main: no overlap
    hipStream_t la_stream = nullptr;              // look-ahead launches of the split-K bulk (null: no look-ahead)
    hipStream_t la_stream2 = nullptr;             // ... of the odd steps (consecutive bulks do not queue behind each other)
    int nrb_steps = 0;                            // block columns that get a step() (== nrb)
    bark_ctx *res = nullptr;
    int nrb = 0, ncb = 0;
    bool fused = false, splitk = false, pipelined = false;
    bool paired = false;  // plain schedule with two block rows per row launch (step_paired)
    // chain-bound chunks (few matrices): diag_kernel(j) itself waits, at its end, for the row launch solve(j) depends on
    // (device-side progress counter) instead of an event wait on the caller's stream; see diag_kernel
    bool dev_wait = false;
    bool pre_update = false;  // ... and the rank-128 / 256 update of the diagonal tile runs as diag_pre_kernel
    // MLL-only sweeps: the last diag_kernel launch of the chunk writes the MLL itself (fin_mll != nullptr); `finished` says it did
    double *fin_mll = nullptr;
    const int32_t *fin_fault = nullptr;
    int fin_2pi = 0;
    bool finished = false;
    bool dev_gate = false;  // ... and the helper streams are released by gate kernels instead of an event record
    int rep = 0;
    double *slabs = nullptr;
    // timing mode only: one event pair per launch, recorded on the stream of the launch
    bool timed = false;
    std::vector<hipEvent_t> ev;
    std::vector<size_t> gram_marks, diag_marks, panel_marks, solve_marks;
    double panel_flops = 0.0, solve_flops = 0.0;

    // Helper streams this call has put work on (bit i: res->helper / helper2 / helper3).  Whatever way the call ends, every
    // one of them is joined back into the caller's stream (rejoin_helpers): under stream capture a forked stream that is not
    // rejoined leaves the capture unjoined (hipStreamEndCapture then fails — or, on this ROCm, crashes: profiles/r04/
    // capture_unjoined_probe.txt), and outside capture the caller's stream must not run ahead of — or the caller free the
    // workspace under — kernels still queued on a helper.
    unsigned touched = 0;
    void touch(hipStream_t st) {
        if (!res || st == main) return;
        if (st == res->helper) touched |= 1u;
        if (st == res->helper2) touched |= 2u;
        if (st == res->helper3) touched |= 4u;
    }
    int after(hipStream_t st, hipEvent_t e) {  // st proceeds once e has completed
        touch(st);
        BARK_HIP_CHECK(hipStreamWaitEvent(st, e, 0));
        return BARK_OK;
    }
    // Everything enqueued on helper stream `st` so far has been ordered in front of what follows on the caller's stream (an event
    // recorded at its tail is awaited there, directly or through a stream that is joined next): nothing of it is left to rejoin.
    // Work put on it later touches it again.  With this the closing rejoin_helpers() of a call whose schedule joined by events
    // finds nothing to do — it used to record and await up to three more events on the caller's stream (5-13 us each on a
    // chain-bound call, and as many extra edges in a captured graph).
    void joined(hipStream_t st) {
        if (!res || st == main) return;
        if (st == res->helper) touched &= ~1u;
        if (st == res->helper2) touched &= ~2u;
        if (st == res->helper3) touched &= ~4u;
    }
    // everything enqueued on the touched helper streams so far precedes what follows on the caller's stream.  Best effort,
    // keeps the thread's error message: it also runs on the way out of a failed call.
    void rejoin_helpers() {
        if (!res) return;
        hipStream_t hs[3] = {res->helper, res->helper2, res->helper3};
        for (int i = 0; i < 3; ++i)
            if ((touched >> i & 1u) && hs[i] && res->rejoin[i]) {
                if (hipEventRecord(res->rejoin[i], hs[i]) == hipSuccess) (void)hipStreamWaitEvent(main, res->rejoin[i], 0);
            }
        touched = 0;
        (void)hipGetLastError();
    }
    ~Sweep() {
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);  // timing events of a call that did not reach report()
    }

    int mark_on(hipStream_t s) {
        if (!timed) return BARK_OK;
        hipEvent_t e;
        BARK_HIP_CHECK(hipEventCreate(&e));
        ev.push_back(e);
        BARK_HIP_CHECK(hipEventRecord(e, s));
        return BARK_OK;
    }

    // syrk != 0: the partial diagonal tile as a SYRK (syrk_tile) — 7/16 less work in one workgroup of every matrix.
    //   1  the SYRK workgroup sits in its matrix's run of tiles.  In the PLAIN schedule of a chunk that is a multiple of the CU
    //      count every workgroup of a launch is equally long and the launch advances in lock step (the co-running tiles of a
    //      matrix stream their shared A panel through L2 at the same k); one short workgroup per matrix staggers every later
    //      round: N = 4096 x 256 94.5 -> 97.4 ms.  Pipelined launches (no lock step to lose) gained 1-2 % from it.
    //   2  the SYRK workgroups of all matrices come LAST in the launch, after every square tile (needs the diagonal tile in
    //      the launch: n_tiles == n_right + 1): the square tiles keep their lock step and the short workgroups fill — or are —
    //      the last round.  Plain, same box, square diagonal tile | this: N = 4096 x 256 95.3 | 92.4 ms, x 512 189.3 | 183.6,
    //      N = 4200 x 256 104.6 | 101.5, N = 8192 x 256 715 | 703; pipelined, mode 1 | 2: N = 4096 x 64 24.04 | 23.67, x 192
    //      70.6 | 69.9, N = 2200 x 256 18.54 | 18.32.  Shipped in both schedules (BARK_PLAIN_SYRK_MODE, BARK_PIPE_SYRK_MODE).
    //   (a third form — the SYRK workgroups as a launch of their own on another stream, released with solve(j), which does not
    //   need them — measured slower: 92.7-96.0 against 91.4-92.5 ms; profiles/r04/headline_power_wall.txt item 6)
    int launch_rows(hipStream_t st, int j, int kdone, int n_right, int n_tiles, int syrk) {
        if (syrk == 2 && n_tiles != n_right + 1) syrk = 0;
        const unsigned diag_wgs = (unsigned)(NXCD * ((p.Bc + NXCD - 1) / NXCD));
        const unsigned grid = syrk == 2 ? xcd_grid(n_right, p.Bc) + diag_wgs : xcd_grid(n_tiles, p.Bc);
        const dim3 g(grid), blk(THREADS);
        if (!fused)
            hipLaunchKernelGGL(row_kernel<0>, g, blk, GEMM_LDS, st, p, j, kdone, n_right, n_tiles, syrk);
        else if (rep == REP_BITS)
            hipLaunchKernelGGL(row_kernel<1 + REP_BITS>, g, blk, GEMM_LDS, st, p, j, kdone, n_right, n_tiles, syrk);
        else if (rep == REP_BYTES7)
            hipLaunchKernelGGL(row_kernel<1 + REP_BYTES7>, g, blk, GEMM_LDS, st, p, j, kdone, n_right, n_tiles, syrk);
        else
            hipLaunchKernelGGL(row_kernel<1 + REP_BYTES8>, g, blk, GEMM_LDS, st, p, j, kdone, n_right, n_tiles, syrk);
        BARK_LAUNCH_CHECK();
        // executed flops: with syrk the partial diagonal tile (present when n_tiles > n_right) takes 36 of 64 sub-block products
        panel_flops += 2.0 * NB * NB * (double)(kdone * NB) * ((double)n_right + (n_tiles > n_right ? (syrk ? 36.0 / 64.0 : 1.0) : 0.0)) * (double)p.Bc;
        return BARK_OK;
    }

    // rows j and j+1 over k < j in one launch (row_kernel, syrk == 4); j + 1 < nrb
    int launch_row_pair(hipStream_t st, int j) {
        const int nA = ncb - j - 1, nB = nA - 1, n_sq = nA + nB, n_syrk = 1 + ((j + 2 < nrb) ? 1 : 0);
        const unsigned grid = xcd_grid(n_sq, p.Bc) + (unsigned)(n_syrk * NXCD * ((p.Bc + NXCD - 1) / NXCD));
        const dim3 g(grid), blk(THREADS);
        if (!fused)
            hipLaunchKernelGGL(row_kernel<0>, g, blk, GEMM_LDS, st, p, j, j, nA, n_sq, 4);
        else if (rep == REP_BITS)
            hipLaunchKernelGGL(row_kernel<1 + REP_BITS>, g, blk, GEMM_LDS, st, p, j, j, nA, n_sq, 4);
        else if (rep == REP_BYTES7)
            hipLaunchKernelGGL(row_kernel<1 + REP_BYTES7>, g, blk, GEMM_LDS, st, p, j, j, nA, n_sq, 4);
        else
            hipLaunchKernelGGL(row_kernel<1 + REP_BYTES8>, g, blk, GEMM_LDS, st, p, j, j, nA, n_sq, 4);
        BARK_LAUNCH_CHECK();
        panel_flops += 2.0 * NB * NB * (double)(j * NB) * ((double)n_sq + n_syrk * 36.0 / 64.0) * (double)p.Bc;
        return BARK_OK;
    }

    // ---- paired plain schedule (lock-step chunks: a multiple of the CU count, not split-K, not pipelined) -----------------------
    // Two block rows per row launch.  Every tile of block row j streams the B panel of its column block from HBM (a block row's
    // tiles share only their A panel, through L2): 197 of the sweep's 222 GB of fetches, ~96 W per TB/s on a chip that runs this
    // sweep at its power cap (profiles/r04/headline_power_wall.txt).  With block rows j and j+1 in ONE launch over the same K
    // range k < j, the tiles (j, c) and (j+1, c) run side by side and share that panel.  Block row j is then complete
    // (diag(j) with the two block rows its partial diagonal tile lacks, plain solve(j)); block row j+1 lacks k = j, which
    // its consumers apply as the pipelined schedule does: diag_kernel(j+1) with one block row and the G block, solve_kernel<1>.
    //   even j:  [diag(j) || rows(j, j+1)] -> solve<0>(j)          odd j:  diag(j) -> solve<1>(j)
    // A last unpaired block row (nrb odd) has no tiles to the right unless there are candidate columns: a plain single launch.
    int step_paired(int j) {
        hipStream_t s = main, ps = panel;
        const int bc = p.Bc, n_right = ncb - j - 1;
        const bool even = (j & 1) == 0, pair = even && j + 1 < nrb;
        int r;
        const bool has_rows = even && (pair || n_right > 0) && (j >= 1 || fused);
        if (has_rows) {
            BARK_HIP_CHECK(hipEventRecord(res->events[6 * j], s));
            if ((r = after(ps, res->events[6 * j]))) return r;
        }
        // blocks the stored P[j,j] lacks: it came from block row j-1's part of a pair launch (k < j-2 if j-1 is the second row of
        // its pair, k < j-1 if it is the first)
        const int nkb = j == 0 ? 0 : (even ? 2 : 1);
        const bool deferred = !even;  // T'[j,.] lacks k = j-1
        if ((r = launch_diag(j, nkb, deferred && n_right > 0, -2, j + 1))) return r;
        if (has_rows) {
            if (timed) panel_marks.push_back(ev.size());
            if ((r = mark_on(ps))) return r;
            if (pair) {
                if ((r = launch_row_pair(ps, j))) return r;
            } else if ((r = launch_rows(ps, j, j, n_right, n_right, 0))) {  // (j + 1 == nrb: candidate columns only, no diagonal tile)
                return r;
            }
            if ((r = mark_on(ps))) return r;
            if ((r = join(6 * j + 1))) return r;
        }
        if (n_right > 0) {
            if (timed) solve_marks.push_back(ev.size());
            if ((r = mark_on(s))) return r;
            const dim3 g(xcd_grid(n_right, bc)), blk(THREADS);
            if (deferred)
                hipLaunchKernelGGL(solve_kernel<1>, g, blk, GEMM_LDS, s, p, j, n_right);
            else
                hipLaunchKernelGGL(solve_kernel<0>, g, blk, GEMM_LDS, s, p, j, n_right);
            BARK_LAUNCH_CHECK();
            if ((r = mark_on(s))) return r;
            solve_flops += ((deferred ? 32.0 : 0.0) + 18.0) / 32.0 * 2.0 * NB * NB * (double)NB * (double)n_right * (double)bc;
        }
        return BARK_OK;
    }

    int publish(hipStream_t st, int slot, int value) {
        touch(st);
        hipLaunchKernelGGL(sync_publish_kernel, dim3(1), dim3(1), 0, st, p.sync, slot, value);
        BARK_LAUNCH_CHECK();
        return BARK_OK;
    }

    int gate(hipStream_t st, int value) {  // st proceeds once the caller's stream has started diag(value - 1)
        touch(st);
        hipLaunchKernelGGL(sync_gate_kernel, dim3(1), dim3(1), 0, st, p.sync, 3, value);
        BARK_LAUNCH_CHECK();
        return BARK_OK;
    }

    // wait_slot: -2 no device-side hand-over; -1 publish the start only; 0 / 1: also wait for that row stream at the end
    // the whole evaluation of a chunk of one-block-row matrices (N <= 128) in one launch: see OneBlock
    int launch_one_block(const double *y, double *mll, const int32_t *fault, int include_2pi) {
        const OneBlock ob{y, mll, fault, include_2pi, rep};
        const size_t lds_bytes = DIAG_LDS + (size_t)p.nW * NB * sizeof(uint32_t);
        // eight waves (factor_tile8) while every matrix of the chunk has a CU to itself; beyond that two four-wave workgroups share one
        const bool w8 = BARK_DIAG_WAVES8 && p.Bc <= DIAG8_MAX_BC;
        if (nrb == 2) {  // two_block_kernel: codes of 256 points + z_0 + U_01' z_0 behind the factor image
            const size_t lds2 = DIAG_LDS + (size_t)p.nW * 2 * NB * sizeof(uint32_t) + 2 * NB * sizeof(double);
            if (w8)
                hipLaunchKernelGGL(two_block_kernel<8>, dim3((unsigned)p.Bc), dim3(512), lds2, main, p, ob);
            else
                hipLaunchKernelGGL(two_block_kernel<4>, dim3((unsigned)p.Bc), dim3(THREADS), lds2, main, p, ob);
        } else if (w8) {
            hipLaunchKernelGGL((diag_kernel<true, 8>), dim3((unsigned)p.Bc), dim3(512), lds_bytes, main, p, 0, 0, 0, -2, 0, ob, 0);
        } else {
            hipLaunchKernelGGL((diag_kernel<true, 4>), dim3((unsigned)p.Bc), dim3(THREADS), lds_bytes, main, p, 0, 0, 0, -2, 0, ob, 0);
        }
        BARK_LAUNCH_CHECK();
        return BARK_OK;
    }

    int launch_diag(int j, int nkb, int want_g = 0, int wait_slot = -2, int wait_value = 0) {
        int r;
        if (timed) diag_marks.push_back(ev.size());
        if ((r = mark_on(main))) return r;
        // a whole CU only while one can be had: beside a heavy look-ahead bulk (448 workgroups resident for ~250 us) no CU
        // is empty, and the request would wait for the bulk to drain (kernel timeline of one N = 16384 matrix: diag 50 ->
        // 200-300 us in the middle steps); there it asks for its 83 KiB and lands beside a single bulk workgroup.  (The
        // pipelined schedule's row launches retire workgroups continuously: there the whole-CU request stays the
        // better choice — one N = 16384 matrix 26.6 against 28.3 ms, N = 4096 x 8 4.87 against 5.34.)
        const bool exclusive = p.Bc <= DIAG_EXCLUSIVE_MAX_BC && !lookahead(j + 1);
        int publish = wait_slot >= -1 ? 1 : 0;
        // chain-bound chunks: the update of the diagonal tile by its block rows above as a launch of its own in front of
        // the one-workgroup kernel (diag_pre_kernel)
        if (pre_update && nkb > 0) {
            hipLaunchKernelGGL(diag_pre_kernel, dim3(NBLK / (THREADS / 64), (unsigned)p.Bc), dim3(THREADS), 0, main, p, j, nkb, publish);
            BARK_LAUNCH_CHECK();
            nkb = 0;
            publish = 0;
        }
        // the last block step of an MLL-only sweep also writes the MLL (no row launch left to wait for there: wait_slot < 0)
        const OneBlock fin = (fin_mll && j == nrb_steps - 1 && wait_slot < 0) ? OneBlock{nullptr, fin_mll, fin_fault, fin_2pi, rep} : OneBlock{};
        if (fin.mll) finished = true;
        if (exclusive && BARK_DIAG_WAVES8)  // a CU to itself: eight waves (factor_tile8)
            hipLaunchKernelGGL((diag_kernel<false, 8>), dim3((unsigned)p.Bc), dim3(512), DIAG_LDS_EXCLUSIVE, main, p, j, nkb, want_g, wait_slot,
                               wait_value, fin, publish);
        else
            hipLaunchKernelGGL((diag_kernel<false, 4>), dim3((unsigned)p.Bc), dim3(THREADS), exclusive ? DIAG_LDS_EXCLUSIVE : DIAG_LDS, main, p, j, nkb,
                               want_g, wait_slot, wait_value, fin, publish);
        BARK_LAUNCH_CHECK();
        return mark_on(main);
    }

    // everything enqueued on `main` so far precedes what follows on the panel stream
    int fork(int slot) {
        if (panel == main) return BARK_OK;
        BARK_HIP_CHECK(hipEventRecord(res->events[slot], main));
        return after(panel, res->events[slot]);
    }
    // everything enqueued on the panel stream so far precedes what follows on `main`
    int join(int slot) {
        if (panel == main) return BARK_OK;
        BARK_HIP_CHECK(hipEventRecord(res->events[slot], panel));
        BARK_HIP_CHECK(hipStreamWaitEvent(main, res->events[slot], 0));
        joined(panel);
        return BARK_OK;
    }

    // split-K factor of step j over nkb block rows: fill ~SPLITK_SLOTS workgroup slots, >= 1 block row per slab
    int split_factor(int j, int nkb, int slots = SPLITK_SLOTS) const {
        const int n_tiles = (ncb - j - 1) + ((j + 1 < nrb) ? 1 : 0);
        if (!splitk || j < 1 || n_tiles <= 0 || n_tiles * p.Bc >= SPLITK_SLOTS / 2 || nkb < 1) return 1;
        int S = slots / (n_tiles * p.Bc);
        if (slots != SPLITK_SLOTS) {  // look-ahead bulk: aim at `slots` workgroups (nearest S), never more than SPLITK_SLOTS
            S = (2 * slots + n_tiles * p.Bc) / (2 * n_tiles * p.Bc);
            while (S > 1 && S * n_tiles * p.Bc > SPLITK_SLOTS) --S;
        }
        if (S < 1) S = 1;
        // a slab on the critical path (the plain split of an under-filled step) may be shorter than a block row: its workgroup is
        // bound by the latency of its k-tiles' DMA stages (~2.7 us each on an idle chip, 8 per block row), not by their MFMAs
        int cap = nkb;
        if (slots == SPLITK_SLOTS && (long)n_tiles * p.Bc * nkb * SPLIT_FINE <= SPLIT_FINE_MAX_WGS) cap = nkb * SPLIT_FINE;  // (see SPLIT_FINE)
        if (S > cap) S = cap;
        if (S > SPLITK_MAX) S = SPLITK_MAX;
        return S;
    }
    // look-ahead step: split layout, under-filled, and at least one block row besides the last (j >= 2)
    // ... and a bulk worth a launch of its own: (tiles x matrices) x block rows >= LA_MIN_WORK, i.e. ~40 us of MFMA
    // work (4.2 MFLOP per tile and block row); below that the step is bound by diag_kernel and the extra launches and
    // events only cost (lone N = 4096: 2.88 ms without, 3.11 ms with look-ahead everywhere)
    bool lookahead(int j) const {
        if (la_stream == nullptr || j < 2 || j >= nrb_steps || split_factor(j, j) <= 1) return false;
        const long n_tiles = (ncb - j - 1) + ((j + 1 < nrb) ? 1 : 0);
        return n_tiles * p.Bc * (long)(j - 1) >= LA_MIN_WORK;
    }
    int la_slots(int j) const {  // look-ahead step j: workgroups its bulk (block rows [0, j-1)) aims at
        const long n_tiles = (ncb - j - 1) + ((j + 1 < nrb) ? 1 : 0);
        return n_tiles * p.Bc * (long)(j - 1) >= LA_BULK_WORK ? LA_SLOTS : LA_SLOTS_CHAIN;
    }
    double *slab_set(int j) const { return slabs + (size_t)(j & 1) * SLAB_SET_TILES * NB * NB; }

    // split-K over block rows [kb_lo, kb_hi) for the tiles [t_off, t_off + nt) of block row j (nt < 0: all of them)
    int launch_split(hipStream_t st, int j, int kb_lo, int kb_hi, int S, int s_off, int S_tot, int t_off = 0, int nt = -1) {
        const int n_right = ncb - j - 1, n_all = n_right + ((j + 1 < nrb) ? 1 : 0);
        if (nt < 0) nt = n_all - t_off;
        hipLaunchKernelGGL(panel_split_kernel, dim3(xcd_grid(nt * S, p.Bc)), dim3(THREADS), GEMM_LDS, st, p, j, n_right, t_off, nt,
                           kb_lo, kb_hi, S, s_off, S_tot, slab_set(j));
        BARK_LAUNCH_CHECK();
        panel_flops += 2.0 * NB * NB * (double)((kb_hi - kb_lo) * NB) * (double)nt * (double)p.Bc;
        return BARK_OK;
    }
    int launch_reduce(hipStream_t st, int j, int S_tot, int t_off = 0, int nt = -1) {
        const int n_right = ncb - j - 1, n_all = n_right + ((j + 1 < nrb) ? 1 : 0);
        if (nt < 0) nt = n_all - t_off;
        const dim3 g((unsigned)(nt * (NB / RED_ROWS)), (unsigned)p.Bc), blk(THREADS);
        if (!fused)
            hipLaunchKernelGGL(panel_reduce_kernel<0>, g, blk, 0, st, p, j, n_right, t_off, nt, S_tot, slab_set(j));
        else if (rep == REP_BITS)
            hipLaunchKernelGGL(panel_reduce_kernel<1 + REP_BITS>, g, blk, 0, st, p, j, n_right, t_off, nt, S_tot, slab_set(j));
        else if (rep == REP_BYTES7)
            hipLaunchKernelGGL(panel_reduce_kernel<1 + REP_BYTES7>, g, blk, 0, st, p, j, n_right, t_off, nt, S_tot, slab_set(j));
        else
            hipLaunchKernelGGL(panel_reduce_kernel<1 + REP_BYTES8>, g, blk, 0, st, p, j, n_right, t_off, nt, S_tot, slab_set(j));
        BARK_LAUNCH_CHECK();
        return BARK_OK;
    }
    // Ragged last round (chunks that are NOT in the split-K layout): n_tiles x Bc workgroups rarely fill whole rounds
    // of the chip's SPLITK_SLOTS slots, and the stragglers of the last round run one per CU for a full tile time
    // (measured at B = 64: up to 35 % per step when one tile in nine is left over).  The tiles beyond the last full
    // round are split over K instead, so that they finish in a fraction of a round: -> first tile of the tail and its
    // split factor (tail == n_tiles: nothing to split).
    void ragged_tail(int j, int n_tiles, int &tail, int &S) const {
        tail = n_tiles;
        S = 1;
        if (splitk || j < 2 || n_tiles <= 0) return;
        const long bc = p.Bc, slots = SPLITK_SLOTS;
        const long rounds = (n_tiles * bc) / slots;
        const long n_plain = rounds * slots / bc;  // tiles that fill whole rounds
        if (n_plain * bc != rounds * slots) return;  // rounds do not end on a tile boundary (Bc does not divide the slots)
        const long m = n_tiles - n_plain;
        // worth it for a sparse last round only (a slab round trip and two more launches): measured at N = 4096,
        // B = 64 27.8 -> 27.0 ms, B = 48 22.8 -> 21.8 ms; a half-filled last round (B = 256, odd tile counts) gains nothing
        if (m <= 0 || m * bc > TAIL_MAX_WGS) return;
        long s = slots / (m * bc);
        if (s > j) s = j;
        if (s > SPLITK_MAX) s = SPLITK_MAX;
        if (s < 3) return;  // a 2-way split of a third-filled round measured slower than leaving it (B = 64, j = 5, 13, 21)
        tail = (int)n_plain;
        S = (int)s;
    }

    // Block column j of the current chunk (p.Bc matrices): diag(j) || rows(j), then solve(j).  diag(j) and rows(j)
    // both depend only on solve(j-1): diag stays on the caller's stream (dispatched the moment solve(j-1) retires,
    // one slot per CU, 82 KiB of LDS leave room for a row workgroup beside it), the rows go to the helper stream
    // and are joined before solve(j).
    // Look-ahead (split-K steps, j >= 2): all of T[j,.]'s K range but its last block row only needs rows < j-1, so that
    // bulk (S slabs) is launched on a third stream right after solve(j-2) and runs beside the whole of step j-1; on
    // the critical path of step j remain diag(j) || the rank-128 slab of block row j-1, the reduce and the solve.
    int step(int j) {
        if (pipelined) return step_pipelined(j);
        if (paired) return step_paired(j);
        hipStream_t s = main, ps = panel;
        const int bc = p.Bc;
        const int n_right = ncb - j - 1;
        const int n_diag = (j + 1 < nrb) ? 1 : 0;
        const int n_tiles = n_right + n_diag;
        int r;
        const bool la = lookahead(j);
        const int S = la ? split_factor(j, j - 1, la_slots(j)) : split_factor(j, j);
        // j == 0: with a materialised A the tiles T = A are in place; in fused-Gram sweeps the K = 0 launch writes them
        const bool has_rows = (j >= 1 || fused) && n_tiles > 0;
        // rows(j) on the panel stream and the look-ahead bulk of step j+1 (rows <= j-1 are final) both start once
        // solve(j-1) has retired.  Event form: ONE event recorded after solve(j-1) (every record between two kernels of
        // the caller's stream costs ~3-7 us there).  Device form (dev_wait): nothing on the caller's stream at all —
        // diag_kernel(j) publishes its start in sync[3] and a one-lane gate kernel heads the helper streams' work.
        const bool bulk_next = lookahead(j + 1);
        const bool rows_off_stream = has_rows && ps != s;
        if (dev_gate && j == 0) {  // the helper streams' gates must not read sync before the prologue has zeroed it
            BARK_HIP_CHECK(hipEventRecord(res->events[5], s));
            if ((r = after(ps, res->events[5])) || (r = after(la_stream, res->events[5])) || (r = after(la_stream2, res->events[5]))) return r;
        }
        if (!dev_gate && (rows_off_stream || bulk_next)) BARK_HIP_CHECK(hipEventRecord(res->events[6 * j], s));
        if (rows_off_stream) {
            if (dev_gate) {
                if ((r = gate(ps, j + 1))) return r;
            } else {
                if ((r = after(ps, res->events[6 * j]))) return r;
            }
        }
        hipStream_t bulk_next_stream = nullptr;
        if (bulk_next) {
            const int j2 = j + 1, S2 = split_factor(j2, j2 - 1, la_slots(j2));
            // bulk-bound steps alternate between two streams, so that a bulk does not queue behind the last workgroups
            // of its predecessor (one N = 16384 matrix 28.7 -> 28.1 ms, with 10^4 candidates 73.5 -> 69.3); in
            // critical-path-bound steps two resident bulks would only take slots from the critical path
            // (N = 4096, B = 8: 5.04 -> 5.26 ms)
            hipStream_t ls = ((j2 & 1) && la_slots(j2) == LA_SLOTS) ? la_stream2 : la_stream;
            bulk_next_stream = ls;
            if (dev_gate) {
                if ((r = gate(ls, j + 1))) return r;
            } else {
                if ((r = after(ls, res->events[6 * j]))) return r;
            }
            if ((r = launch_split(ls, j2, 0, j2 - 1, S2, 0, S2 + 1))) return r;
            BARK_HIP_CHECK(hipEventRecord(res->events[6 * j2 + 2], ls));
        }
        const bool wait_in_diag = dev_wait && rows_off_stream;
        if ((r = launch_diag(j, j > 0 ? 1 : 0, 0, wait_in_diag ? 0 : (dev_wait ? -1 : -2), j + 1))) return r;
        if (has_rows) {
            if (timed) panel_marks.push_back(ev.size());
            if ((r = mark_on(ps))) return r;
            if (la) {  // the last block row of the K range; the bulk [0, j-1) was launched after solve(j-2)
                if ((r = launch_split(ps, j, j - 1, j, 1, S, S + 1))) return r;
                if ((r = after(ps, res->events[6 * j + 2]))) return r;  // the bulk slabs of step j
                {  // the stream of bulk(j) is covered by the join of ps below — unless bulk(j + 1) has just gone onto it as well
                    hipStream_t lsj = ((j & 1) && la_slots(j) == LA_SLOTS) ? la_stream2 : la_stream;
                    if (!wait_in_diag && lsj != bulk_next_stream) joined(lsj);
                }
                if ((r = launch_reduce(ps, j, S + 1))) return r;
            } else if (S > 1) {
                if ((r = launch_split(ps, j, 0, j, S, 0, S))) return r;
                if ((r = launch_reduce(ps, j, S))) return r;
            } else {
                int tail, St;
                ragged_tail(j, n_tiles, tail, St);
                if (tail > 0) {
                    const int lockstep = (tail == n_tiles && p.Bc % PLAIN_CHUNK_MULTIPLE == 0) ? BARK_PLAIN_SYRK_MODE : 0;
                    if ((r = launch_rows(ps, j, j, n_right, tail, lockstep))) {
                        return r;
                    }
                }
                if (tail < n_tiles) {
                    if ((r = launch_split(ps, j, 0, j, St, 0, St, tail, n_tiles - tail))) return r;
                    if ((r = launch_reduce(ps, j, St, tail, n_tiles - tail))) return r;
                }
            }
            if ((r = mark_on(ps))) return r;
            if (wait_in_diag) {  // diag_kernel(j) waits for this counter before it ends: solve(j) follows it back to back
                if ((r = publish(ps, 0, j + 1))) return r;
            } else if ((r = join(6 * j + 1))) {  // solve(j) (and diag(j+1)) need the row's tiles
                return r;
            }
        }
        if (n_right > 0) {
            if (timed) solve_marks.push_back(ev.size());
            if ((r = mark_on(s))) return r;
            // few tiles (the critical path of lone matrices): share a tile's columns out over 4 or 2 workgroups
            if ((long)n_right * bc * 8 <= SOLVE_DIRECT_MAX_WGS)
                hipLaunchKernelGGL(solve_direct_kernel<0>, dim3(xcd_grid(n_right * 8, bc)), dim3(THREADS), 0, s, p, j, n_right);
            else if ((long)n_right * bc * 4 <= SOLVE_NARROW_MAX_WGS)
                hipLaunchKernelGGL(solve_narrow_kernel<1>, dim3(xcd_grid(n_right * 4, bc)), dim3(THREADS), GEMM_LDS, s, p, j, n_right);
            else if ((long)n_right * bc * 2 <= SOLVE_NARROW_MAX_WGS)
                hipLaunchKernelGGL(solve_narrow_kernel<2>, dim3(xcd_grid(n_right * 2, bc)), dim3(THREADS), GEMM_LDS, s, p, j, n_right);
            else
                hipLaunchKernelGGL(solve_kernel<0>, dim3(xcd_grid(n_right, bc)), dim3(THREADS), GEMM_LDS, s, p, j, n_right);
            BARK_LAUNCH_CHECK();
            if ((r = mark_on(s))) return r;
            // 18 of the 32 (k-tile, row-tile) products per wave are executed (zero k-tiles of W_j skipped)
            solve_flops += (18.0 / 32.0) * 2.0 * NB * NB * (double)NB * (double)n_right * (double)bc;
        }
        return BARK_OK;
    }

    // ---- pipelined schedule (chunks with enough matrices to fill the chip, i.e. not in the split-K layout) -----------
    // The row launch of block row j covers only the block rows k < j-1 of its K range ("bulk"), which are final once
    // solve(j-2) has retired; the last block row is applied by the consumers — solve_kernel<1>(j) (one K = 256 product,
    // see there) and diag_kernel(j+1) (two block rows instead of one).  So bulk(j+2) runs beside diag(j+1)
    // and the HBM-bound solve(j+1) instead of waiting for them, and — alternating between the two helper streams —
    // beside the ragged last round of bulk(j+1):
    //   caller's stream  diag(j) -> [wait bulk(j)] solve(j) -> diag(j+1) -> ...
    //   helper streams   bulk(j+2) after solve(j)
    // Every bulk launch is awaited on the caller's stream at its own step, so the pattern stays fork/join (capturable).
    int kdone(int j) const { return j > 0 ? j - 1 : 0; }
    int tiles_of(int j) const { return (ncb - j - 1) + ((j + 1 < nrb) ? 1 : 0); }
    // a K = 0 launch only generates A (fused sweeps); with a materialised A there is nothing to do
    bool has_bulk(int j) const { return j < nrb_steps && tiles_of(j) > 0 && (kdone(j) > 0 || fused); }
    int launch_bulk(int j) {  // everything enqueued on `main` so far precedes it
        if (!has_bulk(j)) return BARK_OK;
        hipStream_t st = (j & 1) ? la_stream : panel;  // one bulk stream only: B = 256 at N = 4096 94 -> 100 ms
        int r;
        if (dev_gate && j >= 2) {  // called right after solve(j-2): diag_kernel(j-1), next on `main`, publishes j when it starts
            if ((r = gate(st, j))) return r;
        } else {
            BARK_HIP_CHECK(hipEventRecord(res->events[6 * j + 3], main));
            if ((r = after(st, res->events[6 * j + 3]))) return r;
        }
        if (timed) panel_marks.push_back(ev.size());
        if ((r = mark_on(st))) return r;
        // the last block rows have few tiles and the longest K: below half a round of workgroups the launch splits K
        // (slabs + generating reduce, both on the bulk stream, off the critical path; slab sets alternate with j like
        // the bulk streams).  N = 4096, B = 32: the last 8 steps took 3.5 of 13.9 ms.
        const int k = kdone(j), nt = tiles_of(j);
        int S = 1;
        if (k >= 2 && nt * p.Bc < SPLITK_SLOTS / 2) {
            S = (2 * PIPE_BULK_SLOTS + nt * p.Bc) / (2 * nt * p.Bc);
            if (S > k) S = k;
            if (S > SPLITK_MAX) S = SPLITK_MAX;
            while (S > 1 && S * nt * p.Bc > SPLITK_SLOTS) --S;
        }
        if (S >= 2) {
            if ((r = launch_split(st, j, 0, k, S, 0, S))) return r;
            if ((r = launch_reduce(st, j, S))) return r;
        } else if ((r = launch_rows(st, j, k, ncb - j - 1, nt, BARK_PIPE_SYRK_MODE))) {
            return r;
        }
        if ((r = mark_on(st))) return r;
        if (dev_wait) return publish(st, j & 1, j + 1);  // diag_kernel(j) waits for the counter of this bulk stream
        BARK_HIP_CHECK(hipEventRecord(res->events[6 * j + 2], st));
        return BARK_OK;
    }
    // (Tried for sweeps bound by their critical path and dropped: applying the last block row by a K = 128 row launch
    // that updates the stored tiles in place beside diag(j), followed by the plain solve — max(diag, update) + solve<0>
    // looked shorter than diag + solve<1>, but the update launch competes with the resident row launches like the
    // solve does: N = 4096, B = 16 7.27 -> 7.98 ms, one N = 16384 matrix 26.3 -> 27.7.)
    int step_pipelined(int j) {
        const int bc = p.Bc, n_right = ncb - j - 1;
        int r;
        if (j == 0) {
            if (dev_gate) {  // the bulk streams' gates must not read sync before the prologue has zeroed it
                BARK_HIP_CHECK(hipEventRecord(res->events[5], main));
                if ((r = after(panel, res->events[5])) || (r = after(la_stream, res->events[5]))) return r;
            }
            if ((r = launch_bulk(0))) return r;
            if ((r = launch_bulk(1))) return r;
        }
        // the stored P_jj comes from the row launch of block row j-1: block rows kdone(j-1) .. j-1 are still to apply
        const bool deferred = j > kdone(j);
        // solve(j) (and the next diag) need bulk(j): awaited by event, or — few matrices — inside diag_kernel(j)
        const bool wait_in_diag = dev_wait && has_bulk(j);
        if ((r = launch_diag(j, j > 0 ? j - kdone(j - 1) : 0, deferred && n_right > 0, wait_in_diag ? (j & 1) : (dev_wait ? -1 : -2),
                             j + 1)))
            return r;
        if (has_bulk(j) && !wait_in_diag) {
            BARK_HIP_CHECK(hipStreamWaitEvent(main, res->events[6 * j + 2], 0));
            joined((j & 1) ? la_stream : panel);  // bulk(j)'s stream; bulk(j + 2) touches it again
        }
        if (n_right > 0) {
            if (timed) solve_marks.push_back(ev.size());
            if ((r = mark_on(main))) return r;
            const long wgs = (long)n_right * bc;
            const int parts = wgs * 4 <= PIPE_NARROW_MAX_WGS ? 4 : wgs * 2 <= PIPE_NARROW_MAX_WGS ? 2 : 1;
            const dim3 g(xcd_grid(n_right * parts, bc)), blk(THREADS);
            if (deferred && parts == 4)
                hipLaunchKernelGGL((solve_narrow_kernel<1, 1>), g, blk, GEMM_LDS, main, p, j, n_right);
            else if (deferred && parts == 2)
                hipLaunchKernelGGL((solve_narrow_kernel<2, 1>), g, blk, GEMM_LDS, main, p, j, n_right);
            else if (deferred)
                hipLaunchKernelGGL(solve_kernel<1>, g, blk, GEMM_LDS, main, p, j, n_right);
            else if (parts == 4)
                hipLaunchKernelGGL((solve_narrow_kernel<1, 0>), g, blk, GEMM_LDS, main, p, j, n_right);
            else if (parts == 2)
                hipLaunchKernelGGL((solve_narrow_kernel<2, 0>), g, blk, GEMM_LDS, main, p, j, n_right);
            else
                hipLaunchKernelGGL(solve_kernel<0>, g, blk, GEMM_LDS, main, p, j, n_right);
            BARK_LAUNCH_CHECK();
            if ((r = mark_on(main))) return r;
            solve_flops += ((deferred ? 32.0 : 0.0) + 18.0) / 32.0 * 2.0 * NB * NB * (double)NB * (double)n_right * (double)bc;
        }
        return launch_bulk(j + 2);
    }

    // fill *t from the recorded events (synchronises); [t_begin, t_end] bracket the whole call on `caller`
    int report(bark_mll_timing *t, size_t t_begin, size_t t_end, hipStream_t caller) {
        BARK_HIP_CHECK(hipStreamSynchronize(caller));
        if (res && res->helper) BARK_HIP_CHECK(hipStreamSynchronize(res->helper));
        if (res && res->helper2) BARK_HIP_CHECK(hipStreamSynchronize(res->helper2));
        if (res && res->helper3) BARK_HIP_CHECK(hipStreamSynchronize(res->helper3));
        int r;
        auto span = [&](size_t a, size_t b_, float *acc) -> int {
            float ms = 0.f;
            BARK_HIP_CHECK(hipEventElapsedTime(&ms, ev[a], ev[b_]));
            *acc += ms;
            return BARK_OK;
        };
        t->total_ms = t->gram_ms = t->chol_ms = t->diag_ms = t->panel_ms = t->solve_ms = 0.f;
        if ((r = span(t_begin, t_end, &t->total_ms))) return r;
        for (size_t a : gram_marks)
            if ((r = span(a, a + 1, &t->gram_ms))) return r;
        for (size_t a : diag_marks)
            if ((r = span(a, a + 1, &t->diag_ms))) return r;
        for (size_t a : panel_marks)
            if ((r = span(a, a + 1, &t->panel_ms))) return r;
        for (size_t a : solve_marks)
            if ((r = span(a, a + 1, &t->solve_ms))) return r;
        t->chol_ms = t->total_ms - t->gram_ms;
        t->n_diag_launches = (int64_t)diag_marks.size();
        t->n_panel_launches = (int64_t)panel_marks.size();
        t->n_solve_launches = (int64_t)solve_marks.size();
        t->panel_flops = panel_flops;
        t->solve_flops = solve_flops;
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
        ev.clear();
        return BARK_OK;
    }
};

// Which schedule a chunk of bc resident matrices takes — the ONE place that decides it: bark_mll_batched_hip configures its
// Sweep from this, and bark_mll_plan_query reports it (DESIGN.md section 4 has the table for the BASELINE configs; a -m gpu
// test asserts it, so that a tuning constant cannot silently move the headline shape onto another schedule).
//   sw: splitk / fused / nrb / ncb / nrb_steps / la_stream set; p.Bc is set here (lookahead() reads the chunk size).
struct ChunkPlan {
    bool one_block, pipelined, paired, dev_wait, dev_gate, pre_update;
    int lookahead_steps, splitk_steps;
};
ChunkPlan plan_chunk(Sweep &sw, int64_t bc, int64_t C, bool timing, bool dev_wait_ok, bool two_streams) {
    ChunkPlan c{};
    const int nrb = sw.nrb;
    const bool splitk = sw.splitk;
    // Pipelined schedule: pays whenever the plain schedule leaves ragged rounds of workgroups (measured at N = 4096:
    // B = 40 +15 %, 64 +7 %, 96 +9 %, 160 +5 %, 192 +4 %; N = 8192, B = 32 +7 %; N = 2048, B = 128..192 +5 %).  When Bc
    // is a multiple of the 256 CUs every round of the plain schedule is full or exactly half full; the two then tie at
    // N = 4096 (93.6 | 93.7 ms at B = 256), the plain one wins beyond (B = 512: 186.6 | 188.6; N = 8192, B = 256:
    // 696 | 711) and the pipelined one up to 16 block rows (N = 2048: 14.1 | 13.8, N = 1024: 2.65 | 2.56).  Fewer than 8
    // block rows: no difference measured (N = 512..896), plain.
    const bool pipeline_ok = !splitk && nrb >= PIPE_MIN_NRB;
    c.pipelined = pipeline_ok && ((bc % PLAIN_CHUNK_MULTIPLE) != 0 || nrb < PLAIN_MIN_NRB);
    // N <= 128: leaf walk + ONE launch per chunk (OneBlock); 128 < N <= 256: the same with the two-block kernel (two_block_kernel),
    // while the codes of 256 points fit behind the factor image in LDS (up to 83 code words per point)
    // ... and while it is the faster form (a workgroup runs its matrix's phases one after the other; the multi-launch sweep spreads a
    // lone matrix over the chip and overlaps the phases of many): round 5, same box, sweep | two_block_kernel, ms —
    //   N = 256:  x 1..8 0.103-0.105 | 0.107-0.109,  x 256 0.158 | 0.133,  x 512 0.247 | 0.257,  x 1024 0.486 | 0.502,  x 2048 0.88 | 0.96
    //   N = 200:  x 32 0.105 | 0.097,  x 256 0.156 | 0.118,  x 512 0.245 | 0.227,  x 1024 0.473 | 0.447      N = 144 x 1024  0.468 | 0.400
    const bool two_ok = nrb == 2 && DIAG_LDS + (size_t)sw.p.nW * 2 * NB * sizeof(uint32_t) + 2 * NB * sizeof(double) <= DIAG_LDS_EXCLUSIVE &&
                        bc >= TWO_MIN_BC && (bc <= TWO_MAX_BC || sw.p.N <= TWO_ANY_BC_MAX_N);
    c.one_block = (nrb == 1 || (BARK_TWO_BLOCK && two_ok)) && sw.fused && C == 0 && !timing;
    sw.p.Bc = (int)bc;  // lookahead() / split_factor() read the chunk size
    if (c.one_block) {
        c.pipelined = false;
        return c;
    }
    c.paired = BARK_PLAIN_PAIRS && !c.pipelined && !splitk && nrb >= PLAIN_MIN_NRB && bc % PLAIN_CHUNK_MULTIPLE == 0 && two_streams;
    c.dev_wait = dev_wait_ok && (splitk || c.pipelined) && bc <= DEVWAIT_MAX_BC;
    // gate kernels in place of the event record that releases the row streams: measured (one process per variant,
    // gates | end-of-diag wait only | events, ms): pipelined N = 4096 x 8 4.44 | 4.59 | 4.64, N = 8192 x 2 8.18 | 8.35 |
    // 8.48; split-K layout N = 4096 x 1 2.19 | 2.14 | 2.22, N = 1024 x 1 0.575 | 0.520 | 0.534 — pipelined only
    // ... split-K layout: only for sweeps with look-ahead steps (N = 6900 x 1 5.02 -> 4.59 with gates; without look-ahead
    // they cost: N = 4096 x 1 2.04 -> 2.28, N = 2048 x 4 1.09 -> 1.21)
    c.dev_gate = c.dev_wait && c.pipelined;
    if (!c.pipelined && !c.paired) {
        for (int jj = 1; jj < nrb; ++jj) {
            const bool la = sw.lookahead(jj);
            c.lookahead_steps += la ? 1 : 0;
            c.splitk_steps += (la || sw.split_factor(jj, jj) > 1) ? 1 : 0;
        }
        if (c.dev_wait && c.lookahead_steps > 0) c.dev_gate = true;
    } else if (c.pipelined) {
        for (int jj = 0; jj < nrb; ++jj) {
            const int k = sw.kdone(jj), nt = sw.tiles_of(jj);
            if (sw.has_bulk(jj) && k >= 2 && nt * (int)bc < SPLITK_SLOTS / 2 && (2 * PIPE_BULK_SLOTS + nt * (int)bc) / (2 * nt * (int)bc) >= 2)
                ++c.splitk_steps;
        }
    }
    // split-K layout only: same box, this | inside diag_kernel, ms — N = 4096 x 1 2.035 | 2.130, N = 1024 x 1 0.492 | 0.514,
    // N = 2048 x 4 1.082 | 1.125; in the pipelined schedule the extra launch queues for slots behind the resident row
    // workgroups like every kernel of the chain does (N = 4096 x 8 4.77 | 4.46, x 16 7.61 | 7.11, N = 16384 x 1 27.2 | 25.4)
    c.pre_update = splitk && bc <= DEVWAIT_MAX_BC;
    return c;
}
int plan_code(const ChunkPlan &c, bool splitk, int nrb) {
    if (c.one_block) return nrb == 2 ? BARK_SCHED_TWO_BLOCK : BARK_SCHED_ONE_BLOCK;
    if (c.paired) return BARK_SCHED_PAIRED;
    if (c.pipelined) return BARK_SCHED_PIPELINED;
    if (splitk) return c.lookahead_steps > 0 ? BARK_SCHED_SPLITK_LOOKAHEAD : BARK_SCHED_SPLITK;
    return BARK_SCHED_PLAIN;
}

}  // namespace
}  // namespace bark

using namespace bark;

namespace bark {
static bool env_set(const char *name) {
    const char *v = std::getenv(name);
    return v && v[0] && !(v[0] == '0' && v[1] == 0);
}
std::atomic<bool> &device_wait_enabled() {
    // off when asked, and when the environment serialises kernel dispatch: every device-side wait would run into its bound
    static std::atomic<bool> on{std::getenv("BARK_NO_DEVICE_WAIT") == nullptr && !env_set("AMD_SERIALIZE_KERNEL") &&
                                !env_set("HIP_LAUNCH_BLOCKING") && !env_set("CUDA_LAUNCH_BLOCKING")};
    return on;
}
}  // namespace bark

extern "C" {

int bark_xcd_map_selftest(int ntiles, int Bc) {
    if (ntiles < 1 || Bc < 1 || (long)ntiles * Bc > (1L << 26)) return -1;
    std::vector<int> hits((size_t)ntiles * Bc, 0);
    const unsigned grid = bark::xcd_grid(ntiles, Bc);
    for (unsigned id = 0; id < grid; ++id) {
        int b, t;
        if (bark::xcd_map((int)id, ntiles, Bc, b, t)) ++hits[(size_t)b * ntiles + t];
    }
    int wrong = 0;
    for (int h : hits) wrong += h != 1;
    return wrong;
}

int bark_device_wait(int on) {
    const bool prev = device_wait_enabled().exchange(on != 0);
    return prev ? 1 : 0;
}

size_t bark_mll_workspace_bytes(int64_t N, int64_t C, int64_t m, int64_t Bc) {
    if (N < 1 || C < 0 || m < 1 || Bc < 1) return 0;
    return make_layout(N, C, m, Bc).total;
}

// Which schedule bark_mll_batched_hip takes for (N, C, m, B, Bc): plan_chunk's decisions for the first and the last chunk.
// No GPU needed (the device-side hand-over is reported as the process switch stands, outside stream capture).
int bark_mll_plan_query(int64_t N, int64_t C, int64_t m, int64_t B, int64_t Bc, int leaf_words, int timing, bark_mll_plan *out) {
    error_buffer()[0] = 0;
    if (!out || N < 1 || C < 0 || m < 1 || B < 1 || Bc < 1 || leaf_words < 1)
        return fail(BARK_ERR_ARG, "bark_mll_plan_query: bad argument");
    if (Bc > B) Bc = B;
    if (Bc > 65535) Bc = 65535;
    const Layout L = make_layout(N, C, m, Bc);
    Sweep sw;
    sw.nrb = sw.nrb_steps = (int)(L.npad / NB);
    sw.ncb = (int)(L.ncols / NB);
    sw.splitk = L.splitk;
    sw.fused = !L.splitk && C == 0 && (size_t)2 * leaf_words * NB * sizeof(uint32_t) <= GEMM_LDS;
    sw.la_stream = sw.la_stream2 = reinterpret_cast<hipStream_t>(&sw);  // non-null: look-ahead possible (never dereferenced)
    sw.p.nW = leaf_words;
    sw.p.N = (int)N;
    const bool dw = device_wait_enabled().load();
    const int64_t last = B % Bc ? B % Bc : Bc;
    const ChunkPlan lastp = plan_chunk(sw, last, C, timing != 0, dw, true);
    const ChunkPlan first = plan_chunk(sw, Bc, C, timing != 0, dw, true);
    out->n_chunks = (int32_t)((B + Bc - 1) / Bc);
    out->chunk = (int32_t)Bc;
    out->last_chunk = (int32_t)last;
    out->schedule = plan_code(first, L.splitk, sw.nrb);
    out->last_schedule = plan_code(lastp, L.splitk, sw.nrb);
    out->splitk_layout = L.splitk ? 1 : 0;
    out->fused_gram = sw.fused ? 1 : 0;
    out->dev_wait = first.dev_wait ? 1 : 0;
    out->dev_gate = first.dev_gate ? 1 : 0;
    out->pre_update = first.pre_update ? 1 : 0;
    out->lookahead_steps = first.lookahead_steps;
    out->splitk_steps = first.splitk_steps;
    out->nrb = sw.nrb;
    out->ncb = sw.ncb;
    return BARK_OK;
}

int bark_mll_batched_hip(bark_ctx *ctx, const void *packed, const bark_pack_info *info, const double *X, int64_t N, int64_t d,
                         const double *y, const double *noise, const double *scale, const double *shift, int flags,
                         const double *cand, int64_t C, double *mll_out, double *mu_out, double *var_out,
                         double *cov_out, int32_t *info_out,
                         void *workspace, size_t workspace_bytes, int64_t Bc, bark_mll_timing *timing, void *stream_) {
    error_buffer()[0] = 0;
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!packed || !info || !X || !y || !noise || !mll_out || !info_out || !workspace)
        return fail(BARK_ERR_ARG, "bark_mll_batched_hip: null argument");
    const int64_t B = info->B, m = info->m;
    if (N < 1 || d < 1 || B < 1 || C < 0 || Bc < 1 || N > (1 << 24) || C > (1 << 24))
        return fail(BARK_ERR_ARG, "bark_mll_batched_hip: bad shape N=%lld d=%lld B=%lld C=%lld Bc=%lld", (long long)N,
                    (long long)d, (long long)B, (long long)C, (long long)Bc);
    const bool rhs_identity = (flags & BARK_MLL_RHS_IDENTITY) != 0;
    if (rhs_identity) {
        if (C != N || !mu_out) return fail(BARK_ERR_ARG, "BARK_MLL_RHS_IDENTITY needs C == N and mu_out");
    } else if (C > 0 && (!cand || !mu_out || !var_out || !scale || !(flags & BARK_MLL_INCLUDE_SCALE))) {
        return fail(BARK_ERR_ARG, "posterior predictive needs cand, mu_out, var_out, scale and BARK_MLL_INCLUDE_SCALE");
    }
    if (cov_out && C == 0) return fail(BARK_ERR_ARG, "cov_out without candidates");
    if ((flags & BARK_MLL_INCLUDE_SCALE) && !scale) return fail(BARK_ERR_ARG, "BARK_MLL_INCLUDE_SCALE without scale");
    if (Bc > B) Bc = B;
    if (Bc > 65535) Bc = 65535;
    const Layout L = make_layout(N, C, m, Bc);
    if (workspace_bytes < L.total)
        return fail(BARK_ERR_WORKSPACE, "workspace too small: %zu < %zu bytes", workspace_bytes, L.total);
    if (reinterpret_cast<uintptr_t>(workspace) & 255) return fail(BARK_ERR_ARG, "workspace must be 256-byte aligned");
    if ((rc = set_lds_limits())) return rc;

    hipStream_t caller = static_cast<hipStream_t>(stream_);
    const int rep = (int)leaf_rep(info);
    const int words = (int)bark_leaf_words(info);
    if (words > MAX_LEAF_WORDS) return fail(BARK_ERR_ARG, "forest needs %d leaf-code words per point (max %d)", words, MAX_LEAF_WORDS);
    const bool use_scale = (flags & BARK_MLL_INCLUDE_SCALE) != 0;
    // MLL-only sweeps generate A inside the row kernels; only the tile (0, 0) is materialised (input of diag(0)).
    // With candidates (or split-K) the whole matrix is filled up front.
    const bool splitk = L.splitk;  // then A is materialised (the reduce kernel reads it)
    const bool fused = !splitk && C == 0 && (size_t)2 * words * NB * sizeof(uint32_t) <= GEMM_LDS;
    double *slabs = reinterpret_cast<double *>(static_cast<char *>(workspace) + L.off_slab);
    const int nrb = (int)(L.npad / NB), ncb = (int)(L.ncols / NB);
    if ((rc = ctx_events(ctx, (size_t)6 * nrb + 6))) return rc;

    Sweep sw;
    sw.res = ctx;
    sw.nrb = nrb;
    sw.nrb_steps = nrb;
    sw.la_stream = ctx->helper2;
    sw.la_stream2 = ctx->helper3;
    sw.ncb = ncb;
    sw.fused = fused;
    sw.splitk = splitk;
    sw.rep = rep;
    sw.slabs = slabs;
    sw.timed = timing != nullptr;
    sw.main = caller;
    sw.panel = ctx->helper;
    uint32_t *leafx, *leafc;
    {
        char *ws = static_cast<char *>(workspace);
        Mats &p = sw.p;
        p.A = reinterpret_cast<double *>(ws + L.off_A);
        p.ld = L.ld;
        p.bstride = L.npad * L.ld;
        p.W = reinterpret_cast<double *>(ws + L.off_W);
        p.yz = reinterpret_cast<double *>(ws + L.off_yz);
        p.accum = reinterpret_cast<double *>(ws + L.off_acc);
        p.sync = reinterpret_cast<int32_t *>(ws + L.off_sync);
        p.nrb = nrb;
        p.ncb = ncb;
        p.nW = words;
        p.m = (int)m;
        p.N = (int)N;
        leafx = reinterpret_cast<uint32_t *>(ws + L.off_leafx);
        leafc = reinterpret_cast<uint32_t *>(ws + L.off_leafc);
    }

    auto prologue = [&](int64_t c0, int64_t bc) -> int {  // leaf walk, Gram fill, right-hand sides of one chunk
        Mats &p = sw.p;
        hipStream_t s = sw.main;
        bark_pack_info sub = *info;
        sub.B = bc;
        const char *packed_c = static_cast<const char *>(packed) + (size_t)c0 * m * info->stride * 16;
        p.info = info_out + c0;
        p.Bc = (int)bc;
        p.leafx = fused ? leafx : nullptr;
        p.scale = use_scale ? scale + c0 : nullptr;
        p.shift = shift ? shift + c0 : nullptr;
        p.noise = noise + c0;
        int r;
        if (sw.timed) sw.gram_marks.push_back(sw.ev.size());
        if ((r = sw.mark_on(s))) return r;
        if ((r = walk_codes(packed_c, &sub, X, N, d, leafx, ctx->fault, s))) return r;
        const int fill = fused ? NB : (int)L.npad;
        r = launch_gram(leafx, (int)L.npad, leafx, (int)L.npad, bc, m, (int)N, (int)N, fill, fill,
                        p.shift, p.scale, p.noise, p.A, L.ld, p.bstride, true, true, rep, words, s);
        if (r) return r;
        if (rhs_identity) {
            dim3 g((unsigned)((L.cpad + 255) / 256), (unsigned)L.npad, (unsigned)bc);
            hipLaunchKernelGGL(identity_rhs_kernel, g, dim3(256), 0, s, p, (int)N, (int)L.cpad);
            BARK_LAUNCH_CHECK();
        } else if (C > 0) {
            if ((r = walk_codes(packed_c, &sub, cand, C, d, leafc, ctx->fault, s))) return r;
            r = launch_gram(leafx, (int)L.npad, leafc, (int)L.cpad, bc, m, (int)N, (int)C, (int)L.npad,
                            (int)L.cpad, p.shift, scale + c0, nullptr, p.A + L.npad, L.ld, p.bstride, false, false, rep, words, s);
            if (r) return r;
        }
        dim3 g((unsigned)((L.npad + 255) / 256), (unsigned)bc);
        hipLaunchKernelGGL(init_rhs_kernel, g, dim3(256), 0, s, y, (int)N, (int)L.npad, p.yz, p.accum, p.info, p.sync);
        BARK_LAUNCH_CHECK();
        return sw.mark_on(s);
    };

    auto epilogue = [&](int64_t c0, int64_t bc) -> int {  // MLL and posterior reductions of one chunk
        Mats &p = sw.p;
        hipStream_t s = sw.main;
        if (!sw.finished) {
            hipLaunchKernelGGL(finish_mll_kernel, dim3((unsigned)((bc + 255) / 256)), dim3(256), 0, s, p.accum, (int)bc, (int)N,
                               (flags & BARK_MLL_INCLUDE_2PI) ? 1 : 0, mll_out + c0, ctx->fault, p.info, sw.dev_wait ? p.sync : nullptr);
            BARK_LAUNCH_CHECK();
        }
        if (C > 0) {
            const int prc = launch_predict_reduce(p, (int)N, (int)C, (int)bc, rhs_identity ? nullptr : scale + c0,
                                                  mu_out + (size_t)c0 * C, var_out ? var_out + (size_t)c0 * C : nullptr,
                                                  L.splitk ? sw.slabs : nullptr, s);
            if (prc) return prc;
            if (cov_out) {
                const int nct = (int)(L.cpad / NB);
                hipLaunchKernelGGL(vtv_kernel, dim3(xcd_grid(nct * nct, (int)bc)), dim3(THREADS), GEMM_LDS, s, p, nct,
                                   (int)C, rhs_identity ? nullptr : scale + c0, rhs_identity ? 1.0 : -1.0,
                                   rhs_identity ? 1 : 0, cov_out + (size_t)c0 * C * C);
                BARK_LAUNCH_CHECK();
            }
        }
        return BARK_OK;
    };

    // device-side waits need the helper streams to run BESIDE the caller's stream: not under stream capture (a graph's
    // branches may be replayed in any order) and not when switched off (bark_device_wait)
    bool dev_wait_ok = device_wait_enabled().load();
    if (dev_wait_ok) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        BARK_HIP_CHECK(hipStreamIsCapturing(caller, &cap));
        dev_wait_ok = cap == hipStreamCaptureStatusNone;
    }
    const size_t t_begin = sw.ev.size();
    if ((rc = sw.mark_on(caller))) return rc;
    auto chunks = [&]() -> int {
    for (int64_t c0 = 0; c0 < B; c0 += Bc) {  // chunks of Bc resident matrices, one after the other
        const int64_t bc = (B - c0 < Bc) ? (B - c0) : Bc;
        const ChunkPlan plan = plan_chunk(sw, bc, C, timing != nullptr, dev_wait_ok, sw.panel != sw.main);  // (sets sw.p.Bc)
        sw.pipelined = plan.pipelined;
        if (plan.one_block) {  // N <= 128: leaf walk + ONE launch per chunk (OneBlock)
            Mats &p1 = sw.p;
            bark_pack_info sub = *info;
            sub.B = bc;
            p1.info = info_out + c0;
            p1.Bc = (int)bc;
            p1.leafx = leafx;
            p1.scale = use_scale ? scale + c0 : nullptr;
            p1.shift = shift ? shift + c0 : nullptr;
            p1.noise = noise + c0;
            if ((rc = walk_codes(static_cast<const char *>(packed) + (size_t)c0 * m * info->stride * 16, &sub, X, N, d, leafx, ctx->fault,
                                 caller)))
                return rc;
            if ((rc = sw.launch_one_block(y, mll_out + c0, ctx->fault, (flags & BARK_MLL_INCLUDE_2PI) ? 1 : 0))) return rc;
            continue;
        }
        sw.paired = plan.paired;
        sw.dev_wait = plan.dev_wait;
        sw.dev_gate = plan.dev_gate;
        sw.pre_update = plan.pre_update;
        sw.finished = false;
        sw.fin_mll = (C == 0 && !timing) ? mll_out + c0 : nullptr;
        sw.fin_fault = ctx->fault;
        sw.fin_2pi = (flags & BARK_MLL_INCLUDE_2PI) ? 1 : 0;
        if ((rc = prologue(c0, bc))) return rc;
        for (int j = 0; j < nrb; ++j)
            if ((rc = sw.step(j))) return rc;
        // device-side hand-over: nothing has joined the helper streams to the caller's yet (the last diag_kernel waited for
        // their counters; after a time-out it did not).  One event join per touched stream and chunk, so that neither the next
        // chunk's prologue nor the caller's next use of the workspace can overtake a row launch of this one.
        if (sw.dev_wait) sw.rejoin_helpers();
        if ((rc = epilogue(c0, bc))) return rc;
    }
    return BARK_OK;
    };
    rc = chunks();
    if (rc) {  // an error return from the middle of a chunk: helper streams may be forked (under capture: unjoined)
        // a gate kernel of a helper stream may be waiting for a diag_kernel that will now never be launched: raise the sticky
        // time-out word first, so that it gives up at once instead of holding the rejoin for its 2 s bound (best effort)
        if (sw.dev_gate && sw.p.sync) {
            hipLaunchKernelGGL(sync_publish_kernel, dim3(1), dim3(1), 0, caller, sw.p.sync, 2, 1);
            (void)hipGetLastError();
        }
        sw.rejoin_helpers();
        return rc;
    }
    sw.rejoin_helpers();  // whatever is still marked touched (Sweep::joined clears a stream's mark when an event join covers it)
    const size_t t_end = sw.ev.size();
    if ((rc = sw.mark_on(caller))) return rc;
    if (timing) return sw.report(timing, t_begin, t_end, caller);
    return BARK_OK;
}

// ---------------------------------------------------------------------------------------------
// Leaf-space MLL (kernels in leafspace.hip): factorises I_R + c Z'Z (R x R) instead of K_s (N x N).
// ---------------------------------------------------------------------------------------------
struct LeafLayout {
    Layout L;         // the R x R sweep workspace (N := R; candidates := R identity columns for the posterior)
    int64_t R, Rpad, W, npad, Q, cpad;
    size_t off_codes, off_planes, off_yy, off_ccodes, off_minv, off_w, off_wm, total;
};

// C > 0: posterior at C candidates; want_inverse: explicit K_s^-1.  Either needs M^-1 (identity columns in the sweep).
static LeafLayout make_leaf_layout(int64_t N, int64_t max_bits, int64_t m, int64_t Bc, int64_t C = 0,
                                   bool want_inverse = false) {
    LeafLayout g;
    const bool want_minv = C > 0 || want_inverse;
    g.R = max_bits;
    g.Rpad = round_up(max_bits, NB);
    g.W = (max_bits + 31) / 32;
    g.npad = round_up(N, NB);
    g.Q = g.npad / 64;
    g.L = make_layout(max_bits, want_minv ? max_bits : 0, m, Bc);
    g.cpad = C > 0 ? round_up(C, NB) : 0;
    size_t o = g.L.total;
    g.off_codes = o;
    o = align256(o + (size_t)Bc * g.W * g.npad * sizeof(uint32_t));
    g.off_planes = o;
    o = align256(o + (size_t)Bc * 32 * g.W * g.Q * sizeof(unsigned long long));
    g.off_yy = o;
    o = align256(o + 64);
    g.off_ccodes = o;
    o = align256(o + (size_t)Bc * g.W * g.cpad * sizeof(uint32_t));
    g.off_minv = o;
    if (want_minv) o = align256(o + (size_t)Bc * max_bits * max_bits * sizeof(double));
    g.off_w = o;
    if (want_minv) o = align256(o + (size_t)Bc * max_bits * sizeof(double));
    g.off_wm = o;
    if (want_inverse) o = align256(o + (size_t)Bc * N * max_bits * sizeof(double));
    g.total = o;
    return g;
}

size_t bark_kernel_inverse_leafspace_workspace_bytes(int64_t N, int64_t max_bits, int64_t m, int64_t Bc) {
    if (N < 1 || max_bits < 1 || m < 1 || Bc < 1) return 0;
    return make_leaf_layout(N, max_bits, m, Bc, 0, true).total;
}

size_t bark_mll_leafspace_workspace_bytes(int64_t N, int64_t max_bits, int64_t m, int64_t Bc, int64_t C) {
    if (N < 1 || max_bits < 1 || m < 1 || Bc < 1 || C < 0) return 0;
    return make_leaf_layout(N, max_bits, m, Bc, C).total;
}

}  // extern "C"

// shared driver of the leaf-space entry points: MLL always; posterior when C > 0; explicit inverse when kinv_out
static int leafspace_run(bark_ctx *ctx, const void *packed, const bark_pack_info *info, const double *X, int64_t N, int64_t d,
                         const double *y, const double *noise, const double *scale, int flags, const double *cand,
                         int64_t C, double *mll_out, double *mu_out, double *var_out, double *kinv_out,
                         double *kinv_y_out, int32_t *info_out, void *workspace, size_t workspace_bytes, int64_t Bc,
                         void *stream_) {
    error_buffer()[0] = 0;
    int rc = check_ctx(ctx);
    if (rc) return rc;
    if (!packed || !info || !X || !y || !noise || !mll_out || !info_out || !workspace)
        return fail(BARK_ERR_ARG, "leaf-space entry: null argument");
    const int64_t B = info->B, m = info->m;
    if (N < 1 || d < 1 || B < 1 || Bc < 1 || C < 0 || N > (1 << 24) || C > (1 << 24))
        return fail(BARK_ERR_ARG, "bark_mll_leafspace_hip: bad shape N=%lld d=%lld B=%lld Bc=%lld C=%lld", (long long)N,
                    (long long)d, (long long)B, (long long)Bc, (long long)C);
    if (C > 0 && (!cand || !mu_out || !var_out || !scale || !(flags & BARK_MLL_INCLUDE_SCALE)))
        return fail(BARK_ERR_ARG, "leaf-space posterior needs cand, mu_out, var_out, scale and BARK_MLL_INCLUDE_SCALE");
    if ((flags & BARK_MLL_INCLUDE_SCALE) && !scale) return fail(BARK_ERR_ARG, "BARK_MLL_INCLUDE_SCALE without scale");
    if (flags & BARK_MLL_RHS_IDENTITY) return fail(BARK_ERR_ARG, "leaf-space path computes the MLL only");
    if (info->max_bits > 8192) return fail(BARK_ERR_ARG, "leaf-space path supports at most 8192 leaves per forest");
    if (Bc > B) Bc = B;
    if (Bc > 65535) Bc = 65535;
    const bool want_minv = C > 0 || kinv_out != nullptr;
    if (kinv_out && info->max_bits > 65535) return fail(BARK_ERR_ARG, "leaf-space inverse: too many leaves");
    const LeafLayout g = make_leaf_layout(N, info->max_bits, m, Bc, C, kinv_out != nullptr);
    if (workspace_bytes < g.total) return fail(BARK_ERR_WORKSPACE, "workspace too small: %zu < %zu bytes", workspace_bytes, g.total);
    if (reinterpret_cast<uintptr_t>(workspace) & 255) return fail(BARK_ERR_ARG, "workspace must be 256-byte aligned");
    if ((rc = set_lds_limits())) return rc;
    hipStream_t caller = static_cast<hipStream_t>(stream_);
    const int nrb = (int)(g.Rpad / NB);
    if ((rc = ctx_events(ctx, (size_t)6 * nrb + 6))) return rc;

    char *ws = static_cast<char *>(workspace);
    const int ncb = (int)(g.L.ncols / NB);  // posterior: R identity columns appended (M^-1 and w = M^-1 v)
    Sweep sw;
    sw.res = ctx;
    sw.nrb = nrb;
    sw.nrb_steps = nrb;
    sw.la_stream = ctx->helper2;
    sw.la_stream2 = ctx->helper3;
    sw.ncb = ncb;
    sw.fused = false;
    sw.splitk = g.L.splitk;
    sw.slabs = reinterpret_cast<double *>(ws + g.L.off_slab);
    sw.main = caller;
    sw.panel = ctx->helper;
    Mats &p = sw.p;
    p.A = reinterpret_cast<double *>(ws + g.L.off_A);
    p.ld = g.L.ld;
    p.bstride = g.L.npad * g.L.ld;
    p.W = reinterpret_cast<double *>(ws + g.L.off_W);
    p.yz = reinterpret_cast<double *>(ws + g.L.off_yz);
    p.accum = reinterpret_cast<double *>(ws + g.L.off_acc);
    p.sync = nullptr;  // R x R systems: a block row or two, event joins are fine (sw.dev_wait stays false)
    p.nrb = nrb;
    p.ncb = ncb;
    p.leafx = nullptr;
    p.scale = p.shift = p.noise = nullptr;
    p.nW = 0;
    p.m = (int)m;
    p.N = (int)g.R;
    uint32_t *codes = reinterpret_cast<uint32_t *>(ws + g.off_codes);
    unsigned long long *planes = reinterpret_cast<unsigned long long *>(ws + g.off_planes);
    double *yy = reinterpret_cast<double *>(ws + g.off_yy);
    uint32_t *ccodes = reinterpret_cast<uint32_t *>(ws + g.off_ccodes);
    double *Minv = reinterpret_cast<double *>(ws + g.off_minv);
    double *wvec = reinterpret_cast<double *>(ws + g.off_w);
    const bool use_scale = (flags & BARK_MLL_INCLUDE_SCALE) != 0;

    if ((rc = leafspace_sumsq(y, (int)N, yy, caller))) return rc;
    auto chunks = [&]() -> int {
    for (int64_t c0 = 0; c0 < B; c0 += Bc) {
        const int64_t bc = (B - c0 < Bc) ? (B - c0) : Bc;
        bark_pack_info sub = *info;
        sub.B = bc;
        const char *packed_c = static_cast<const char *>(packed) + (size_t)c0 * m * info->stride * 16;
        p.info = info_out + c0;
        p.Bc = (int)bc;
        sw.pipelined = !g.L.splitk && nrb >= PIPE_MIN_NRB && ((bc % PLAIN_CHUNK_MULTIPLE) != 0 || nrb < PLAIN_MIN_NRB);  // plan_chunk's rule
        if ((rc = walk_one_hot(packed_c, &sub, X, N, d, (int)g.W, codes, ctx->fault, caller))) return rc;
        rc = leafspace_prepare(codes, (int)g.W, (int)g.npad, planes, (int)g.R, (int)g.Rpad, noise + c0,
                               use_scale ? scale + c0 : nullptr, (int)m, (int)bc, p.A, p.ld, p.bstride, y, (int)N, p.yz,
                               p.accum, p.info, caller);
        if (rc) return rc;
        if (want_minv) {  // right-hand side block := I_R, so the sweep also yields V = U^-T
            dim3 gi((unsigned)((g.L.cpad + 255) / 256), (unsigned)g.L.npad, (unsigned)bc);
            hipLaunchKernelGGL(identity_rhs_kernel, gi, dim3(256), 0, caller, p, (int)g.R, (int)g.L.cpad);
            BARK_LAUNCH_CHECK();
        }
        for (int j = 0; j < nrb; ++j)
            if ((rc = sw.step(j))) return rc;
        rc = leafspace_finish(p.accum, yy, noise + c0, use_scale ? scale + c0 : nullptr, (int)m, (int)bc, (int)N,
                              (flags & BARK_MLL_INCLUDE_2PI) ? 1 : 0, mll_out + c0, caller);
        if (rc) return rc;
        if (want_minv) {
            // w = M^-1 v = V'z and M^-1 = V'V (the same kernels the dense posterior / inverse export use)
            const int R = (int)g.R;
            if ((rc = launch_predict_reduce(p, R, R, (int)bc, nullptr, wvec, nullptr, g.L.splitk ? sw.slabs : nullptr, caller)))
                return rc;
            const int nct = (int)(g.L.cpad / NB);
            hipLaunchKernelGGL(vtv_kernel, dim3(xcd_grid(nct * nct, (int)bc)), dim3(THREADS), GEMM_LDS, caller, p, nct, R,
                               (const double *)nullptr, 1.0, 1, Minv);
            BARK_LAUNCH_CHECK();
            if (C > 0) {
                if ((rc = walk_one_hot(packed_c, &sub, cand, C, d, (int)g.W, ccodes, ctx->fault, caller))) return rc;
                rc = leafspace_predict(ccodes, (int)g.W, (int)g.cpad, (int)C, wvec, Minv, R, noise + c0, scale + c0, (int)m,
                                       (int)bc, mu_out + (size_t)c0 * C, var_out + (size_t)c0 * C, caller);
                if (rc) return rc;
            }
            if (kinv_out) {
                rc = leafspace_inverse(codes, (int)g.W, (int)g.npad, (int)N, Minv, wvec, R, y, noise + c0,
                                       use_scale ? scale + c0 : nullptr, (int)m, (int)bc,
                                       reinterpret_cast<double *>(ws + g.off_wm), kinv_out + (size_t)c0 * N * N,
                                       kinv_y_out ? kinv_y_out + (size_t)c0 * N : nullptr, caller);
                if (rc) return rc;
            }
        }
        hipLaunchKernelGGL(fault_info_kernel, dim3((unsigned)((bc + 255) / 256)), dim3(256), 0, caller, ctx->fault, p.info,
                           (int)bc);
        BARK_LAUNCH_CHECK();
    }
    return BARK_OK;
    };
    rc = chunks();
    sw.rejoin_helpers();  // also on an error return from the middle of a sweep: no helper stream stays forked
    return rc;
}

extern "C" {

int bark_mll_leafspace_hip(bark_ctx *ctx, const void *packed, const bark_pack_info *info, const double *X, int64_t N, int64_t d,
                           const double *y, const double *noise, const double *scale, int flags, const double *cand,
                           int64_t C, double *mll_out, double *mu_out, double *var_out, int32_t *info_out,
                           void *workspace, size_t workspace_bytes, int64_t Bc, void *stream_) {
    return leafspace_run(ctx, packed, info, X, N, d, y, noise, scale, flags, cand, C, mll_out, mu_out, var_out, nullptr, nullptr,
                         info_out, workspace, workspace_bytes, Bc, stream_);
}

int bark_kernel_inverse_leafspace_hip(bark_ctx *ctx, const void *packed, const bark_pack_info *info, const double *X, int64_t N, int64_t d,
                                      const double *y, const double *noise, const double *scale, int flags,
                                      double *mll_out, double *kinv_out, double *kinv_y_out, int32_t *info_out,
                                      void *workspace, size_t workspace_bytes, int64_t Bc, void *stream_) {
    if (!kinv_out) {
        error_buffer()[0] = 0;
        return fail(BARK_ERR_ARG, "bark_kernel_inverse_leafspace_hip: kinv_out is null");
    }
    return leafspace_run(ctx, packed, info, X, N, d, y, noise, scale, flags, nullptr, 0, mll_out, nullptr, nullptr, kinv_out,
                         kinv_y_out, info_out, workspace, workspace_bytes, Bc, stream_);
}

// out[b] = alpha * sum_i A[b][i] * y[i] + beta * c[b]  (one wave per row; fixed order; c may be null)
__global__ __launch_bounds__(64) void rowdot_kernel(const double *__restrict__ A, const double *__restrict__ y, int64_t N,
                                                     int64_t lda, double alpha, const double *__restrict__ c, double beta,
                                                     double *__restrict__ out) {
    const double *row = A + (size_t)blockIdx.x * lda;
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < N; i += 64) s = fma(row[i], y[i], s);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (threadIdx.x == 0) out[blockIdx.x] = c ? alpha * s + beta * c[blockIdx.x] : alpha * s;
}

int bark_rowdot_hip(const double *A, int64_t B, int64_t N, int64_t lda, const double *y, double alpha, const double *c,
                    double beta, double *out, void *stream_) {
    error_buffer()[0] = 0;
    if (!A || !y || !out || B < 1 || N < 1 || lda < N || B > (1 << 30)) return fail(BARK_ERR_ARG, "bark_rowdot_hip: bad argument");
    hipLaunchKernelGGL(rowdot_kernel, dim3((unsigned)B), dim3(64), 0, static_cast<hipStream_t>(stream_), A, y, N, lda, alpha, c,
                       beta, out);
    BARK_LAUNCH_CHECK();
    return BARK_OK;
}

int bark_quadform_hip(const double *K_inv, const double *y, int64_t N, double *out, void *stream_) {
    error_buffer()[0] = 0;
    if (!K_inv || !y || !out || N < 1 || N > (1 << 30)) return fail(BARK_ERR_ARG, "bark_quadform_hip: bad argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    BARK_HIP_CHECK(hipMemsetAsync(out, 0, sizeof(double), stream));
    const int grid = (int)(N < 1024 ? N : 1024);
    hipLaunchKernelGGL(quadform_kernel, dim3(grid), dim3(THREADS), 0, stream, K_inv, y, (int)N, out);
    BARK_LAUNCH_CHECK();
    return BARK_OK;
}

}  // extern "C"

#ifdef BARK_DIAG_STAMPS
extern "C" int bark_debug_diag_stamps(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(bark::g_diag_stamps), sizeof(unsigned long long) * 64);
}
#endif
#ifdef BARK_TWO_STAMPS
extern "C" int bark_debug_two_stamps(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(bark::g_two_stamps), sizeof(unsigned long long) * 16);
}
#endif
