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

}  // namespace bark

#include "chol_solve.h"  // -> chol_rows.h -> chol_diag.h -> chol_tiles.h

namespace bark {
namespace {

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
#ifndef BARK_WALK_IN_KERNEL
#define BARK_WALK_IN_KERNEL 1  // N <= 128 (eight-wave diag_kernel<true>): the leaf walk inside the kernel, no walk launch
#endif
#ifndef BARK_TWO_BLOCK
#define BARK_TWO_BLOCK 1  // 128 < N <= 256, MLL only: the one-launch evaluation by two_block_kernel (0: the multi-launch sweep)
#endif
// ... for chunks of TWO_MIN_BC .. TWO_MAX_BC matrices, and larger chunks up to TWO_ANY_BC_MAX_N points (plan_chunk has the table)
#ifndef BARK_TWO_MIN_BC
#define BARK_TWO_MIN_BC 1
#endif
constexpr int TWO_MIN_BC = BARK_TWO_MIN_BC, TWO_MAX_BC = 384, TWO_ANY_BC_MAX_N = 224;
#ifndef BARK_MULTI_BLOCK
#define BARK_MULTI_BLOCK 1  // 256 < N <= 768, MLL only: the one-launch evaluation by multi_block_kernel
#endif
// Chunks of at least MB_MIN_BC4 (four block rows) / MB_MIN_BC3 (three) matrices: the kernel runs a matrix's block steps one after the
// other on ONE CU (~0.35 ms at N = 512 whatever the batch), the sweep spreads a matrix over the chip.  Round 5, same box, one process per
// variant, sweep | multi_block_kernel, ms (with the kernel's three-stage ring really in flight; before that the window was 160 .. 320):
//   N = 512:  x 64 0.282 | 0.347,  x 80 0.313 | 0.348,  x 96 0.343 | 0.348,  x 112 0.370 | 0.351,  x 128 0.365 | 0.350,  x 256 0.44-0.52 | 0.385,
//             x 384 0.782 | 0.720,  x 512 0.901 | 0.786,  x 768 1.345 | 1.180,  x 1024 1.741 | 1.588,  x 2048 3.336 | 3.135,  x 4096 6.402 | 6.181
//   N = 384:  x 64 0.187 | 0.193,  x 96 0.202 | 0.199,  x 128 0.210 | 0.200,  x 384 0.440 | 0.402,  x 512 0.512 | 0.410,  x 1000 1.024 | 0.821
//   N = 300:  x 16 0.152 | 0.166,  x 32 0.177 | 0.168,  x 64 0.186 | 0.178,  x 128 0.207 | 0.185,  x 512 0.496 | 0.382,  x 2048 1.863 | 1.588
#ifndef BARK_MB_MIN_BC4
#define BARK_MB_MIN_BC4 112
#endif
#ifndef BARK_MB_MIN_BC3
#define BARK_MB_MIN_BC3 80
#endif
// FIVE and SIX block rows (512 < N <= 768; forests whose leaf codes fit: 7 / 6 words a point — the one-hot codes of prior-sized trees), same
// box, sweep | multi_block_kernel, ms: a matrix now holds its CU for 0.58 / 0.90 ms, so the chunk must nearly fill whole rounds of the 256 CUs
//   N = 640:  x 128 0.527 | 0.584   x 160 0.633 | 0.601   x 192 0.788 | 0.625   x 224 0.819 | 0.649   x 256 0.826 | 0.673   x 384 1.266 | 1.253
//             x 512 1.497 | 1.317   x 2048 5.474 | 5.229
//   N = 768:  x 128 0.784 | 0.903   x 160 0.982 | 0.933   x 192 1.081 | 0.954   x 224 1.190 | 0.998   x 256 1.197 | 1.015   x 384 1.886 | 1.959
//             x 512 2.234 | 2.051   x 1024 4.347 | 4.069          N = 600 x 256 0.814 | 0.666      N = 700 x 1024 4.246 | 3.990
#ifndef BARK_MB_MIN_BC5
#define BARK_MB_MIN_BC5 144
#endif
constexpr int MB_MIN_BC4 = BARK_MB_MIN_BC4, MB_MIN_BC3 = BARK_MB_MIN_BC3, MB_MIN_BC5 = BARK_MB_MIN_BC5;
// Chunks of more matrices than CUs run in rounds, and a round that is mostly empty costs a whole matrix time: sweep | kernel, ms —
//   N = 512:  x 272 0.621 | 0.717   x 300 0.643 | 0.722   x 320 0.664 | 0.723   x 352 0.733 | 0.723   x 544 1.117 | 1.102   x 600 1.169 | 1.104
//   N = 384:  x 272 0.357 | 0.400   x 300 0.376 | 0.398   x 340 0.397 | 0.403          N = 300:  x 272 0.354 | 0.370   x 320 0.371 | 0.373
// -> beyond one round only chunks that fill at least 70 % (five / six block rows: 80 %) of ceil(chunk / CUs) rounds
constexpr int MB_CUS = 256;
inline bool mb_chunk_ok(int nrb, int64_t bc) {
    const int64_t rounds = (bc + MB_CUS - 1) / MB_CUS;
    if (bc < (nrb <= 3 ? MB_MIN_BC3 : nrb == 4 ? MB_MIN_BC4 : MB_MIN_BC5)) return false;
    if (rounds <= 1) return true;
    return nrb <= 4 ? bc * 10 >= rounds * MB_CUS * 7 : bc * 5 >= rounds * MB_CUS * 4;
}
// dynamic LDS of multi_block_kernel: factor image, vec, red, y and z (a block per block row of the matrix each), three A-panel stages, the leaf codes
inline size_t mb_lds_bytes(int nW, int nrb) {
    return (size_t)(NBLK * SB * SB + 2 * NB + 8 + 2 * nrb * NB + 3 * MB_STAGE) * sizeof(double) + (size_t)nW * nrb * NB * sizeof(uint32_t);
}
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
        if (e == hipSuccess) e = set(reinterpret_cast<const void *>(multi_block_kernel), DIAG_LDS_EXCLUSIVE);
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
    // N <= 128 with a CU per matrix: the leaf walk runs inside the kernel (OneBlock::nodes) while the points' rows and the forest's
    // packed nodes fit where the factor image will be
    bool walk_in_kernel(const bark_pack_info *sub, int64_t d) const {
        return BARK_WALK_IN_KERNEL && nrb == 1 && BARK_DIAG_WAVES8 && p.Bc <= DIAG8_MAX_BC &&
               ((size_t)p.N * (d | 1) + 2 + (size_t)sub->m * sub->stride * 2) * sizeof(double) <= (size_t)NBLK * SB * SB * sizeof(double);
    }
    int launch_one_block(const double *y, double *mll, int32_t *fault, int include_2pi, const void *packed_c, const bark_pack_info *sub,
                         const double *X, int64_t d) {
        OneBlock ob{y, mll, fault, include_2pi, rep};
        if (walk_in_kernel(sub, d)) {
            ob.nodes = static_cast<const uint4 *>(packed_c);
            ob.X = X;
            ob.fault_w = fault;
            ob.stride = (int)sub->stride;
            ob.m = (int)sub->m;
            ob.max_depth = (int)sub->max_depth;
            ob.d = (int)d;
        }
        const size_t lds_bytes = DIAG_LDS + (size_t)p.nW * NB * sizeof(uint32_t);
        // eight waves (factor_tile8) while every matrix of the chunk has a CU to itself; beyond that two four-wave workgroups share one
        const bool w8 = BARK_DIAG_WAVES8 && p.Bc <= DIAG8_MAX_BC;
        if (nrb >= 3) {  // multi_block_kernel (256 < N <= 768)
            hipLaunchKernelGGL(multi_block_kernel, dim3((unsigned)p.Bc), dim3(512), mb_lds_bytes(p.nW, nrb), main, p, ob);
        } else if (nrb == 2) {  // two_block_kernel: codes of 256 points + z_0 + U_01' z_0 behind the factor image
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
    //   N = 256:  x 1..8 0.103-0.105 | 0.107-0.109 (four waves; with eight waves and the tile generation over all of them 0.098 | 0.088: every
    //             batch size from 1 on since),  x 256 0.158 | 0.133,  x 512 0.247 | 0.257,  x 1024 0.486 | 0.502,  x 2048 0.88 | 0.96
    //   N = 200:  x 32 0.105 | 0.097,  x 256 0.156 | 0.118,  x 512 0.245 | 0.227,  x 1024 0.473 | 0.447      N = 144 x 1024  0.468 | 0.400
    const bool two_ok = nrb == 2 && DIAG_LDS + (size_t)sw.p.nW * 2 * NB * sizeof(uint32_t) + 2 * NB * sizeof(double) <= DIAG_LDS_EXCLUSIVE &&
                        bc >= TWO_MIN_BC && (bc <= TWO_MAX_BC || sw.p.N <= TWO_ANY_BC_MAX_N);
    c.one_block = (nrb == 1 || (BARK_TWO_BLOCK && two_ok)) && sw.fused && C == 0 && !timing;
    // 256 < N <= 768: multi_block_kernel (three to six block rows in one launch, eight waves), whatever layout the sweep would
    // take for the chunk, while the codes of the matrix's points fit beside the factor image and the GEMM stages in LDS
    if (BARK_MULTI_BLOCK && nrb >= 3 && nrb <= MB_MAX_NRB && C == 0 && !timing && mb_chunk_ok(nrb, bc) &&
        mb_lds_bytes(sw.p.nW, nrb) <= DIAG_LDS_EXCLUSIVE)
        c.one_block = true;
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
    if (c.one_block) return nrb >= 3 ? BARK_SCHED_MULTI_BLOCK : nrb == 2 ? BARK_SCHED_TWO_BLOCK : BARK_SCHED_ONE_BLOCK;
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
            const char *packed_c = static_cast<const char *>(packed) + (size_t)c0 * m * info->stride * 16;
            if (!sw.walk_in_kernel(&sub, d) && (rc = walk_codes(packed_c, &sub, X, N, d, leafx, ctx->fault, caller))) return rc;
            if ((rc = sw.launch_one_block(y, mll_out + c0, ctx->fault, (flags & BARK_MLL_INCLUDE_2PI) ? 1 : 0, packed_c, &sub, X, d))) return rc;
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
extern "C" int bark_debug_mb_stamps(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(bark::g_mb_stamps), sizeof(unsigned long long) * 64);
}
extern "C" int bark_debug_two_stamps(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(bark::g_two_stamps), sizeof(unsigned long long) * 16);
}
#endif
