// Woodbury / matrix-determinant-lemma updates on gfx950 — src/bark/fitting/quick_inverse.py:13-33.
//
//   low_rank_inv_update(K_inv, U, subtract):  K_inv - K_inv U (mul I + U' K_inv U)^-1 U' K_inv      (:13-21)
//   low_rank_det_update(K_inv, U, logdet, s): logdet + log|det(I + mul U' K_inv U)|                 (:24-33)
//
// with mul = -1 for subtract.  In the sampler U is N x r (r = #leaves of one tree, single digits), so
// the work is two streaming passes over the N x N matrix and r x r algebra in between:
//   1. skinny_kernel   Y = K_inv U, plus every workgroup's share of G = U'Y (and Y'y)       (HBM read, 8 N^2 B)
//      colsum_kernel + colsum_finish_kernel: the same for symmetric K_inv (and K_inv'U in general), column form
//   2. small_kernel    G = sum of shares, den = mul I + G, LU with partial pivoting -> den^-1, log|det|
//                      (one workgroup; r <= 64)
//   3. left_factor_kernel   M = Y den^-1                                                   (N r, tiny)
//   4. rank_update_kernel   out = K_inv - M Y'   64x64 tiles, M/Y strips in LDS          (HBM read+write, 16 N^2 B)
// K_inv is treated as a general (not necessarily symmetric) matrix exactly as the reference's formula does:
// the right factor is U' K_inv, computed as its own pass when `symmetric == 0`.
#include <vector>

#include "common.h"

namespace bark {
int walk_one_hot(const void *packed, const bark_pack_info *info, const double *X, int64_t N, int64_t d, int words,
                 uint32_t *out, int32_t *fault, hipStream_t stream);  // traverse.hip

namespace {

constexpr int LR_MAX = 64;  // max rank r
constexpr int LR_THREADS = 256;
#ifndef SK_RW_SMALL
#define SK_RW_SMALL 2  // rows per wave for r <= 16
#endif
#ifndef SK_KL_SMALL
#define SK_KL_SMALL 1  // 128-column groups per chunk for r <= 16
#endif

typedef double lr_double2 __attribute__((ext_vector_type(2)));

// Several independent chains in one launch: the chain index is the highest grid dimension a kernel does not use
// otherwise; chain b's matrices sit `k` doubles apart and its workspace block (identical internal layout) `ws`
// doubles apart.  Single-chain entry points launch with one chain, the strides are then irrelevant.
constexpr int MAX_CHAINS = 64;
struct Chain {
    size_t k, ws;
};
struct ChainInts {
    int v[MAX_CHAINS];
};
struct ChainDoubles {
    double v[MAX_CHAINS];
};

// Y = K U for a skinny U (N x r), K row-major N x N, streamed once at HBM rate.
// A workgroup owns 4*RW rows (RW per wave); a lane owns KL column pairs of every KC = 128*KL column chunk
// (16-byte loads when N is even), the next chunk's K values are in flight while the current one is
// multiplied, and the chunk of U (KC x r) is double-buffered in LDS and shared by all rows.
// Epilogue: besides Y the workgroup writes its share of G = U'Y (r x r) and, with `y`, of v = Y'y (r) to
// `partial[block][r*r + r]`; reduce_shares_kernel adds the shares in block order, so results are reproducible.
template <int RT, bool VEC, int RW, int KL>  // RT: compile-time bound on r; VEC: N even
__global__ __launch_bounds__(LR_THREADS) void skinny_kernel(const double *__restrict__ K, const double *__restrict__ U,
                                                            int N, int r, double *__restrict__ out,
                                                            const double *__restrict__ y,
                                                            double *__restrict__ partial) {
    constexpr int ROWS = 4 * RW, KC = 128 * KL, KCP = KC + 2;  // padded: the transposing stores spread over banks
    extern __shared__ __attribute__((aligned(16))) double us[];  // 2 x [r][KCP]: U chunk transposed, so a lane's
                                                                 // two k values are one conflict-free 16-byte read
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = blockIdx.x * ROWS;
    const double *Krow[RW];
#pragma unroll
    for (int q = 0; q < RW; ++q) Krow[q] = K + (size_t)min(row0 + RW * wave + q, N - 1) * N;  // clamped, not stored
    double acc[RW][RT];
#pragma unroll
    for (int q = 0; q < RW; ++q)
#pragma unroll
        for (int c = 0; c < RT; ++c) acc[q][c] = 0.0;

    auto load2 = [&](const double *row, int k) -> lr_double2 {
        lr_double2 v = {0.0, 0.0};
        if (VEC) {
            if (k < N) v = *reinterpret_cast<const lr_double2 *>(row + k);  // N even: k + 1 < N as well
        } else {
            if (k < N) v.x = row[k];
            if (k + 1 < N) v.y = row[k + 1];
        }
        return v;
    };
    auto stage = [&](int buf, int k0) {
        double *dst = us + (size_t)buf * r * KCP;
        const int rows = min(KC, N - k0);
        const double *src = U + (size_t)k0 * r;  // rows k0.. are contiguous in U: coalesced reads
        for (int e = threadIdx.x; e < KC * r; e += LR_THREADS) {
            const int k = e / r, c = e - k * r;
            dst[c * KCP + k] = (e < rows * r) ? src[e] : 0.0;
        }
    };

    // Workgroups start at different chunks and wrap around: with a power-of-two row pitch, lock-step reads of
    // the same column range from every row would all land on the same few HBM channels.
    const int chunks = (N + KC - 1) / KC;
    const int first = (int)(blockIdx.x % (unsigned)chunks);
    stage(0, first * KC);
    lr_double2 kv[RW][KL], nx[RW][KL];
#pragma unroll
    for (int q = 0; q < RW; ++q)
#pragma unroll
        for (int h = 0; h < KL; ++h) kv[q][h] = load2(Krow[q], first * KC + 128 * h + 2 * lane);
    __syncthreads();
    for (int ch = 0; ch < chunks; ++ch) {
        int next_chunk = first + ch + 1;
        if (next_chunk >= chunks) next_chunk -= chunks;
        const int k_next = next_chunk * KC;
        if (ch + 1 < chunks) {
#pragma unroll
            for (int q = 0; q < RW; ++q)
#pragma unroll
                for (int h = 0; h < KL; ++h) nx[q][h] = load2(Krow[q], k_next + 128 * h + 2 * lane);
            stage((ch + 1) & 1, k_next);
        }
#pragma unroll
        for (int h = 0; h < KL; ++h) {
            const double *u0 = us + (size_t)(ch & 1) * r * KCP + 128 * h + 2 * lane;
#pragma unroll
            for (int c = 0; c < RT; ++c)
                if (c < r) {
                    const lr_double2 u = *reinterpret_cast<const lr_double2 *>(u0 + c * KCP);
#pragma unroll
                    for (int q = 0; q < RW; ++q) acc[q][c] = fma(kv[q][h].y, u.y, fma(kv[q][h].x, u.x, acc[q][c]));
                }
        }
        if (ch + 1 < chunks) {
#pragma unroll
            for (int q = 0; q < RW; ++q)
#pragma unroll
                for (int h = 0; h < KL; ++h) kv[q][h] = nx[q][h];
        }
        __syncthreads();  // the other buffer is complete, this one may be overwritten next iteration
    }
    // wave reduction; every lane ends with the full sums
#pragma unroll
    for (int q = 0; q < RW; ++q)
#pragma unroll
        for (int c = 0; c < RT; ++c)
            if (c < r) {
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) acc[q][c] += __shfl_xor(acc[q][c], off);
            }
    // Y rows -> global and LDS (for the r x r share); LDS is free again after the loop's last barrier
    double *ys = us;                 // [ROWS][r]
    double *ur = us + ROWS * r;      // [ROWS][r] rows of U
    double *yv = us + 2 * ROWS * r;  // [ROWS]    entries of y
#pragma unroll
    for (int q = 0; q < RW; ++q) {
        const int lrow = RW * wave + q, row = row0 + lrow;
#pragma unroll
        for (int c = 0; c < RT; ++c)
            if (c < r && lane == (c & 63)) {
                ys[lrow * r + c] = row < N ? acc[q][c] : 0.0;
                if (row < N) out[(size_t)row * r + c] = acc[q][c];
            }
    }
    if (partial) {
        for (int e = threadIdx.x; e < ROWS * r; e += LR_THREADS) {
            const int q = e / r;
            ur[e] = (row0 + q < N) ? U[(size_t)row0 * r + e] : 0.0;
        }
        if ((int)threadIdx.x < ROWS) yv[threadIdx.x] = (y && row0 + (int)threadIdx.x < N) ? y[row0 + threadIdx.x] : 0.0;
        __syncthreads();
        double *dst = partial + (size_t)blockIdx.x * (r * r + r);
        for (int e = threadIdx.x; e < r * r + r; e += LR_THREADS) {
            double g = 0.0;
            if (e < r * r) {
                const int a = e / r, b = e - a * r;
#pragma unroll
                for (int q = 0; q < ROWS; ++q) g = fma(ur[q * r + a], ys[q * r + b], g);
            } else {
                const int b = e - r * r;
#pragma unroll
                for (int q = 0; q < ROWS; ++q) g = fma(yv[q], ys[q * r + b], g);
            }
            dst[e] = g;
        }
    }
}

// Column form for a symmetric K (and K'U for any K):  P[seg][i][c] = sum_{k in segment} K[k][i] * U[k][c].
// A lane owns two adjacent output indices i, so a wave reads 1 KiB of row k per step; U[k][.] is the same for
// the whole wave (scalar loads), nothing goes through LDS inside the loop, there is no barrier and no cross-lane
// reduction, and CS_UNROLL rows are in flight per wave.  grid = (ceil(N/128), ceil(N/seg_rows)); the four waves of a
// workgroup take interleaved rows of the segment and are summed through LDS in wave order.  N must be even.
constexpr int CS_UNROLL = 4;     // rows in flight per wave (measured: 4 > 8 > 16 at N = 4096)
constexpr int CS_FIN_ROWS = 16;  // rows of Y per colsum_finish_kernel workgroup

// rows of K per workgroup: 128 up to N = 4096 (1024 workgroups there; fewer, longer ones measured slower), growing
// with N so that at most 32 segment partials are written and summed
// ... and shrinking below N = 2048, where the launch is a chain of row batches per wave (latency, not bytes: 128 rows were 15.6 us
// at N = 1000 for 8 MB) — 32 rows up to N = 1024, 64 up to 2048: still at most 32 partials
inline int colsum_segment(int64_t N) { return N <= 1024 ? 32 : N <= 2048 ? 64 : 128 * (int)((N + 4095) / 4096); }

// U given as one-hot leaf codes (tree swaps in column form: r <= 16 leaf vectors = the bits of ONE code word per point, scaled
// by the chain's s): the kernels read the word and test a bit instead of loading r doubles of a materialised U — and the
// expand launch in front of them is gone (one dependent launch less per proposal).  Entry U[k][c] = bit c of codes[k] ? s : 0.
struct OneHot {
    const uint32_t *codes;  // (chains, 1, npad) or nullptr: U is a matrix
    int npad;
    ChainDoubles scales;
    int *flag;  // cleared by colsum_kernel's first workgroup per chain (the "singular" flag small_kernel sets later)
};

template <int RT, bool ONEHOT>
__global__ __launch_bounds__(LR_THREADS) void colsum_kernel(const double *__restrict__ K, const double *__restrict__ U,
                                                            int N, int r, int seg_rows, double *__restrict__ P,
                                                            Chain ch, OneHot oh) {
    extern __shared__ __attribute__((aligned(16))) double red[];  // [4][128][r]
    K += blockIdx.z * ch.k;  // chain = blockIdx.z
    U += blockIdx.z * ch.ws;
    P += blockIdx.z * ch.ws;
    const uint32_t *cw = ONEHOT ? oh.codes + (size_t)blockIdx.z * oh.npad : nullptr;
    const double sc = ONEHOT ? oh.scales.v[blockIdx.z] : 0.0;
    if (ONEHOT && oh.flag && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
        *reinterpret_cast<int *>(reinterpret_cast<double *>(oh.flag) + blockIdx.z * ch.ws) = 0;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = blockIdx.x * 128 + 2 * lane;
    const int k_begin = blockIdx.y * seg_rows, k_end = min(N, k_begin + seg_rows);
    const bool live = i < N;  // N even: i + 1 < N as well
    double ax[RT], ay[RT];
#pragma unroll
    for (int c = 0; c < RT; ++c) ax[c] = ay[c] = 0.0;
    for (int k0 = k_begin + wave * CS_UNROLL; k0 < k_end; k0 += 4 * CS_UNROLL) {
        lr_double2 kv[CS_UNROLL];
#pragma unroll
        for (int u = 0; u < CS_UNROLL; ++u) {
            kv[u] = lr_double2{0.0, 0.0};
            if (live && k0 + u < k_end) kv[u] = *reinterpret_cast<const lr_double2 *>(K + (size_t)(k0 + u) * N + i);
        }
#pragma unroll
        for (int u = 0; u < CS_UNROLL; ++u) {
            const double *urow = U + (size_t)min(k0 + u, N - 1) * r;  // wave-uniform address: scalar loads
            const uint32_t word = ONEHOT ? cw[min(k0 + u, N - 1)] : 0u;
#pragma unroll
            for (int c = 0; c < RT; ++c)
                if (c < r) {
                    const double uv = ONEHOT ? (((word >> c) & 1u) ? sc : 0.0) : urow[c];
                    ax[c] = fma(kv[u].x, uv, ax[c]);
                    ay[c] = fma(kv[u].y, uv, ay[c]);
                }
        }
    }
    // the four waves hold disjoint row subsets: sum them in wave order
#pragma unroll
    for (int c = 0; c < RT; ++c)
        if (c < r) {
            red[((size_t)wave * 128 + 2 * lane) * r + c] = ax[c];
            red[((size_t)wave * 128 + 2 * lane + 1) * r + c] = ay[c];
        }
    __syncthreads();
    double *dst = P + ((size_t)blockIdx.y * N + (size_t)blockIdx.x * 128) * r;
    const int cols = min(128, N - blockIdx.x * 128);
    for (int e = threadIdx.x; e < cols * r; e += LR_THREADS)
        dst[e] = ((red[e] + red[128 * r + e]) + red[2 * 128 * r + e]) + red[3 * 128 * r + e];
}

// Y[i][c] = sum over segments of P[seg][i][c] (fixed order); with `partial`, also this block of CS_FIN_ROWS rows'
// share of G = U'Y and v = Y'y, in the layout reduce_shares_kernel expects.
__global__ __launch_bounds__(LR_THREADS) void colsum_finish_kernel(const double *__restrict__ P, int segs,
                                                                   const double *__restrict__ U, int N, int r,
                                                                   double *__restrict__ Y, const double *__restrict__ y,
                                                                   double *__restrict__ partial, Chain ch, OneHot oh) {
    extern __shared__ __attribute__((aligned(16))) double sh[];  // ys[ROWS][r] | ur[ROWS][r] | yv[ROWS]
    constexpr int ROWS = CS_FIN_ROWS;
    P += blockIdx.y * ch.ws;  // chain = blockIdx.y; y is shared by the chains
    U += blockIdx.y * ch.ws;
    Y += blockIdx.y * ch.ws;
    if (partial) partial += blockIdx.y * ch.ws;
    double *ys = sh, *ur = sh + ROWS * r, *yv = sh + 2 * ROWS * r;
    const int row0 = blockIdx.x * ROWS;
    const int rows = min(ROWS, N - row0);
    for (int e = threadIdx.x; e < ROWS * r; e += LR_THREADS) {
        double v = 0.0, uv = 0.0;
        if (e < rows * r) {
            const size_t at = (size_t)row0 * r + e;
#pragma unroll 8
            for (int sg = 0; sg < segs; ++sg) v += P[(size_t)sg * N * r + at];
            Y[at] = v;
            if (partial) {
                if (oh.codes) {
                    const int q = e / r, c = e - q * r;
                    uv = ((oh.codes[(size_t)blockIdx.y * oh.npad + row0 + q] >> c) & 1u) ? oh.scales.v[blockIdx.y] : 0.0;
                } else {
                    uv = U[at];
                }
            }
        }
        ys[e] = v;
        ur[e] = uv;
    }
    if (!partial) return;
    if ((int)threadIdx.x < ROWS) yv[threadIdx.x] = (y && (int)threadIdx.x < rows) ? y[row0 + threadIdx.x] : 0.0;
    __syncthreads();
    double *dst = partial + (size_t)blockIdx.x * (r * r + r);
    for (int e = threadIdx.x; e < r * r + r; e += LR_THREADS) {
        double g = 0.0;
        if (e < r * r) {
            const int a = e / r, b = e - a * r;
#pragma unroll
            for (int q = 0; q < ROWS; ++q) g = fma(ur[q * r + a], ys[q * r + b], g);
        } else {
            const int b = e - r * r;
#pragma unroll
            for (int q = 0; q < ROWS; ++q) g = fma(yv[q], ys[q * r + b], g);
        }
        dst[e] = g;
    }
}

// sums[e] = sum over blocks of partial[block][e], one wave per entry, fixed order
__global__ __launch_bounds__(LR_THREADS) void reduce_shares_kernel(const double *__restrict__ partial, int nblocks,
                                                                   int per, double *__restrict__ sums, Chain ch) {
    partial += blockIdx.y * ch.ws;  // chain = blockIdx.y
    sums += blockIdx.y * ch.ws;
    const int e = blockIdx.x * (LR_THREADS / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= per) return;
    double g = 0.0;
    for (int blk = lane; blk < nblocks; blk += 64) g += partial[(size_t)blk * per + e];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) g += __shfl_xor(g, off);
    if (lane == 0) sums[e] = g;
}

// out[i][c] = sum_k K[k][i] * U[k][c]  (= (U'K)': column i of K) — only for a K_inv not known to be symmetric.
// One wave per output index i, lanes stride k.
template <int RT>
__global__ __launch_bounds__(LR_THREADS) void skinny_t_kernel(const double *__restrict__ K, const double *__restrict__ U,
                                                              int N, int r, double *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double us[];  // [64][r] chunk of U
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * (LR_THREADS / 64) + wave;
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
        if (i < N && k < N) kv = K[(size_t)k * N + i];
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

// One workgroup: G = the summed shares (r x r, then v: r entries) from reduce_shares_kernel, den = C + G with C = diag(+-1) (the first r_neg
// columns carry -1: leaf vectors being removed), Gauss-Jordan with partial pivoting -> den^-1 (written to
// `inv`, r x r) and log|det den|.  With `dquad`: v = sum of the v shares and dquad = v' den^-1 v (the change of
// y'K^-1 y).  `logabsdet` / `dquad` may be null.
// Device-side tree sweeps (bark_tree_sweep_chains_hip) — bound by the HOST's launch rate up to N ~ 1000 (7 us per launch, ten
// launches per tree proposal) — fold two neighbours into this kernel:
//   partial != nullptr: the shares are summed here (reduce_shares_kernel's order: a wave per entry, lanes stride the blocks, xor
//                       butterfly) instead of being read from `sums`;
//   dec.accept_out != nullptr: the Metropolis decision of decide_kernel, for this chain, once dquad and log|det| are known.
struct DecideArgs {
    const double *log_q_prior, *log_u;  // (chains,) of this step
    double *state;                      // (chains, 2) running y'K^-1 y, log|K|
    const int32_t *accept_prev;         // previous step's flags or nullptr
    int32_t *accept_out;                // (chains,) or nullptr: no decision here
};
__global__ __launch_bounds__(LR_THREADS) void small_kernel(const double *__restrict__ sums, int r, ChainInts r_negs,
                                                           double *__restrict__ inv,
                                                           double *__restrict__ logabsdet, int *__restrict__ singular,
                                                           double *__restrict__ dquad, Chain ch, int out_stride,
                                                           const double *__restrict__ partial, int nblocks, DecideArgs dec) {
    __shared__ double aug[LR_MAX][2 * LR_MAX + 1];  // [den | I]
    __shared__ double vsh[LR_MAX];
    __shared__ int piv_row;
    const int tid = threadIdx.x;
    const int r_neg = r_negs.v[blockIdx.x];  // chain = blockIdx.x; scalar outputs are out_stride doubles apart
    sums += blockIdx.x * ch.ws;
    if (inv) inv += blockIdx.x * ch.ws;
    if (singular) singular = reinterpret_cast<int *>(reinterpret_cast<double *>(singular) + blockIdx.x * ch.ws);
    if (logabsdet) logabsdet += (size_t)blockIdx.x * out_stride;
    if (dquad) dquad += (size_t)blockIdx.x * out_stride;
    bool is_singular = false;  // thread 0's view
    auto place = [&](int e, double g) {  // entry e of the summed shares -> [den | I] and v
        if (e < r * r) {
            const int a = e / r, bcol = e - a * r;
            aug[a][bcol] = g + (a == bcol ? (a < r_neg ? -1.0 : 1.0) : 0.0);
            aug[a][r + bcol] = (a == bcol) ? 1.0 : 0.0;
        } else {
            vsh[e - r * r] = g;
        }
    };
    if (partial) {
        partial += blockIdx.x * ch.ws;
        const int per = r * r + r, lane = tid & 63;
        for (int e = tid >> 6; e < per; e += LR_THREADS / 64) {
            double g = 0.0;
            for (int blk = lane; blk < nblocks; blk += 64) g += partial[(size_t)blk * per + e];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) g += __shfl_xor(g, off);
            if (lane == 0) place(e, g);
        }
    } else {
        for (int e = tid; e < r * r + r; e += LR_THREADS) place(e, sums[e]);
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
            if (bv == 0.0) is_singular = true;
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
    if (tid == 0 && (dquad || dec.accept_out)) {
        double q = 0.0;
        for (int a = 0; a < r; ++a)
            for (int b2 = 0; b2 < r; ++b2) q = fma(vsh[a] * aug[a][r + b2], vsh[b2], q);
        if (dquad) *dquad = q;
        if (dec.accept_out) {  // decide_kernel, for chain blockIdx.x
            const int b = blockIdx.x;
            int acc;
            if (is_singular || (dec.accept_prev && dec.accept_prev[b] < 0)) {
                acc = -1;
            } else {
                const double log_alpha = dec.log_q_prior[b] + 0.5 * (q - logsum);
                acc = (dec.log_u[b] <= fmin(log_alpha, 0.0)) ? 1 : 0;  // NaN compares false: reject
            }
            dec.accept_out[b] = acc;
            if (acc > 0) {
                dec.state[2 * b] = dec.state[2 * b] - q;
                dec.state[2 * b + 1] = dec.state[2 * b + 1] + logsum;
            }
        }
    }
    if (inv)
        for (int e = tid; e < r * r; e += LR_THREADS) inv[e] = aug[e / r][r + (e - (e / r) * r)];
}

// M = Y den^-1 (N x r): one thread per entry, den^-1 in LDS.
__global__ __launch_bounds__(LR_THREADS) void left_factor_kernel(const double *__restrict__ Y,
                                                                 const double *__restrict__ inv, int N, int r,
                                                                 double *__restrict__ M, Chain ch, ChainInts accept,
                                                                 const int32_t *__restrict__ accept_dev) {
    __shared__ double is[LR_MAX * LR_MAX];
    // chain = blockIdx.y (uniform per workgroup); the decision comes from the host (accept) or, in a device-side
    // sweep, from decide_kernel's flags (accept_dev, > 0 = accepted)
    if (accept_dev ? accept_dev[blockIdx.y] <= 0 : !accept.v[blockIdx.y]) return;
    Y += blockIdx.y * ch.ws;
    inv += blockIdx.y * ch.ws;
    M += blockIdx.y * ch.ws;
    for (int e = threadIdx.x; e < r * r; e += LR_THREADS) is[e] = inv[e];
    __syncthreads();
    const size_t e = (size_t)blockIdx.x * LR_THREADS + threadIdx.x;
    if (e >= (size_t)N * r) return;
    const int i = (int)(e / r), c = (int)(e - (size_t)i * r);
    double s = 0.0;
    for (int a = 0; a < r; ++a) s = fma(Y[(size_t)i * r + a], is[a * r + c], s);
    M[e] = s;
}

// out[i][j] = K[i][j] - sum_a M[i][a] * R[j][a]     (R = Y for symmetric K_inv, else (U'K_inv)')
// inv != nullptr (device-side sweeps): M is not read — the workgroup forms its 64 rows of M = Y den^-1 itself (left_factor_kernel's
// arithmetic per entry; Y = the rows of `Yrows`), a launch less per tree proposal.
__global__ __launch_bounds__(LR_THREADS) void rank_update_kernel(const double *__restrict__ K,
                                                                 const double *__restrict__ M,
                                                                 const double *__restrict__ R, int N, int r,
                                                                 double *__restrict__ out, Chain ch, ChainInts accept,
                                                                 const int32_t *__restrict__ accept_dev,
                                                                 const double *__restrict__ inv,
                                                                 const double *__restrict__ Yrows) {
    extern __shared__ __attribute__((aligned(16))) double strips[];  // Ms[64][r] | Rs[64][r] (| Ys[64][r] | is[r][r])
    if (accept_dev ? accept_dev[blockIdx.z] <= 0 : !accept.v[blockIdx.z]) return;  // chain = blockIdx.z
    K += blockIdx.z * ch.k;
    out += blockIdx.z * ch.k;
    M += blockIdx.z * ch.ws;
    R += blockIdx.z * ch.ws;
    double *Ms = strips, *Rs = strips + 64 * r;
    const int row0 = blockIdx.y * 64, col0 = blockIdx.x * 64;
    if (inv) {
        inv += blockIdx.z * ch.ws;
        Yrows += blockIdx.z * ch.ws;
        double *Ys = strips + 2 * 64 * r, *is = strips + 3 * 64 * r;
        for (int e = threadIdx.x; e < 64 * r; e += LR_THREADS) {
            const int q = e / r;
            Ys[e] = (row0 + q < N) ? Yrows[(size_t)(row0 + q) * r + (e - q * r)] : 0.0;
        }
        for (int e = threadIdx.x; e < r * r; e += LR_THREADS) is[e] = inv[e];
        __syncthreads();
        for (int e = threadIdx.x; e < 64 * r; e += LR_THREADS) {
            const int q = e / r, c = e - q * r;
            double s = 0.0;
            for (int a = 0; a < r; ++a) s = fma(Ys[q * r + a], is[a * r + c], s);
            Ms[e] = (row0 + q < N) ? s : 0.0;
            Rs[e] = (col0 + q < N) ? R[(size_t)(col0 + q) * r + (e - q * r)] : 0.0;
        }
    } else {
        for (int e = threadIdx.x; e < 64 * r; e += LR_THREADS) {
            const int q = e / r, a = e - q * r;
            Ms[e] = (row0 + q < N) ? M[(size_t)(row0 + q) * r + a] : 0.0;
            Rs[e] = (col0 + q < N) ? R[(size_t)(col0 + q) * r + a] : 0.0;
        }
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

// U[i][c] = s if bit c of point i's one-hot leaf code is set else 0  (codes: [words][npad] planes)
// flag != nullptr: also clears the chain's "singular" flag (small_kernel, later in the stream, sets it) — a memset launch less.
__global__ __launch_bounds__(LR_THREADS) void expand_onehot_kernel(const uint32_t *__restrict__ codes, int words, int npad,
                                                                   int N, int r, ChainDoubles scales,
                                                                   double *__restrict__ U, Chain ch, int *__restrict__ flag) {
    const double s = scales.v[blockIdx.y];  // chain = blockIdx.y; codes are (chains, words, npad) contiguous
    codes += (size_t)blockIdx.y * words * npad;
    U += blockIdx.y * ch.ws;
    if (flag && blockIdx.x == 0 && threadIdx.x == 0) *reinterpret_cast<int *>(reinterpret_cast<double *>(flag) + blockIdx.y * ch.ws) = 0;
    const size_t e = (size_t)blockIdx.x * LR_THREADS + threadIdx.x;
    if (e >= (size_t)N * r) return;
    const int i = (int)(e / r), c = (int)(e - (size_t)i * r);
    U[e] = ((codes[(size_t)(c >> 5) * npad + i] >> (c & 31)) & 1u) ? s : 0.0;
}

// rows per wave / chunk width per rank bound: enough workgroups and bytes in flight at small r, bounded
// registers and LDS at large r
template <int RT>
struct SkinnyCfg {
    static constexpr int RW = RT <= 16 ? SK_RW_SMALL : 2;
    static constexpr int KL = RT <= 16 ? SK_KL_SMALL : 1;
};

template <int RT>
int launch_skinny_rt(hipStream_t stream, const double *K, const double *U, int N, int r, double *out, const double *y,
                     double *partial) {
    constexpr int RW = SkinnyCfg<RT>::RW, KL = SkinnyCfg<RT>::KL, ROWS = 4 * RW;
    const dim3 grid((unsigned)((N + ROWS - 1) / ROWS));
    const size_t lds = (size_t)2 * (128 * KL + 2) * r * sizeof(double);  // 130 KiB at r = 64
    if (N % 2 == 0) {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute((const void *)skinny_kernel<RT, true, RW, KL>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((skinny_kernel<RT, true, RW, KL>), grid, dim3(LR_THREADS), lds, stream, K, U, N, r, out, y,
                           partial);
    } else {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute((const void *)skinny_kernel<RT, false, RW, KL>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((skinny_kernel<RT, false, RW, KL>), grid, dim3(LR_THREADS), lds, stream, K, U, N, r, out, y,
                           partial);
    }
    return (int)grid.x;
}

// returns the number of workgroups (= shares written to `partial`)
int launch_skinny(hipStream_t stream, const double *K, const double *U, int N, int r, double *out, const double *y,
                  double *partial) {
    if (r <= 8) return launch_skinny_rt<8>(stream, K, U, N, r, out, y, partial);
    if (r <= 16) return launch_skinny_rt<16>(stream, K, U, N, r, out, y, partial);
    if (r <= 32) return launch_skinny_rt<32>(stream, K, U, N, r, out, y, partial);
    return launch_skinny_rt<64>(stream, K, U, N, r, out, y, partial);
}

// Column form K'U (== K U for symmetric K): usable when N is even and r <= 16.  `partial` may be null (no shares).
bool colsum_usable(int64_t N, int64_t r) { return N % 2 == 0 && r <= 16; }

int launch_colsum(hipStream_t stream, const double *K, const double *U, int N, int r, double *P, double *out,
                  const double *y, double *partial, int nc = 1, Chain ch = Chain{0, 0}, const OneHot *onehot = nullptr) {
    const int seg_rows = colsum_segment(N), segs = (N + seg_rows - 1) / seg_rows;
    const dim3 grid((unsigned)((N + 127) / 128), (unsigned)segs, (unsigned)nc);
    const size_t lds = (size_t)4 * 128 * r * sizeof(double);  // 64 KiB at r = 16
    const OneHot oh = onehot ? *onehot : OneHot{nullptr, 0, ChainDoubles{}, nullptr};
    if (onehot && r <= 8)
        hipLaunchKernelGGL((colsum_kernel<8, true>), grid, dim3(LR_THREADS), lds, stream, K, U, N, r, seg_rows, P, ch, oh);
    else if (onehot)
        hipLaunchKernelGGL((colsum_kernel<16, true>), grid, dim3(LR_THREADS), lds, stream, K, U, N, r, seg_rows, P, ch, oh);
    else if (r <= 8)
        hipLaunchKernelGGL((colsum_kernel<8, false>), grid, dim3(LR_THREADS), lds, stream, K, U, N, r, seg_rows, P, ch, oh);
    else
        hipLaunchKernelGGL((colsum_kernel<16, false>), grid, dim3(LR_THREADS), lds, stream, K, U, N, r, seg_rows, P, ch, oh);
    const int nblocks = (N + CS_FIN_ROWS - 1) / CS_FIN_ROWS;
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)nblocks, (unsigned)nc), dim3(LR_THREADS),
                       (size_t)(2 * CS_FIN_ROWS * r + CS_FIN_ROWS) * sizeof(double), stream, P, segs, U, N, r, out, y, partial, ch, oh);
    return nblocks;
}

// shares -> sums -> den^-1, log|det|, dquad
// dec != nullptr (device-side sweeps): one launch — small_kernel sums the shares itself and takes the Metropolis decision
void launch_small(hipStream_t stream, const double *partial, int nblocks, int r, const ChainInts &r_neg, double *sums,
                  double *inv, double *logabsdet, int *flag, double *dquad, int nc = 1, Chain ch = Chain{0, 0},
                  int out_stride = 0, const DecideArgs *dec = nullptr) {
    const int per = r * r + r;
    if (dec) {
        hipLaunchKernelGGL(small_kernel, dim3((unsigned)nc), dim3(LR_THREADS), 0, stream, sums, r, r_neg, inv, logabsdet, flag,
                           dquad, ch, out_stride, partial, nblocks, *dec);
        return;
    }
    hipLaunchKernelGGL(reduce_shares_kernel, dim3((unsigned)((per + 3) / 4), (unsigned)nc), dim3(LR_THREADS), 0, stream,
                       partial, nblocks, per, sums, ch);
    hipLaunchKernelGGL(small_kernel, dim3((unsigned)nc), dim3(LR_THREADS), 0, stream, sums, r, r_neg, inv, logabsdet, flag,
                       dquad, ch, out_stride, (const double *)nullptr, 0, DecideArgs{});
}

ChainInts one_int(int v) {
    ChainInts c = {};
    c.v[0] = v;
    return c;
}

// M = Y den^-1, then out = K - M R'  for the chains whose accept entry is set
// fused (device-side sweeps): no left_factor launch — rank_update_kernel forms its rows of M itself
void launch_rewrite(hipStream_t stream, const double *K, int N, int r, const double *Y, const double *inv, double *M,
                    const double *R, double *out, const ChainInts &accept, int nc = 1, Chain ch = Chain{0, 0},
                    const int32_t *accept_dev = nullptr, bool fused = false) {
    const unsigned tiles = (unsigned)((N + 63) / 64);
    if (fused) {
        hipLaunchKernelGGL(rank_update_kernel, dim3(tiles, tiles, (unsigned)nc), dim3(LR_THREADS),
                           (size_t)(3 * 64 * r + r * r) * sizeof(double), stream, K, M, R, N, r, out, ch, accept, accept_dev, inv, Y);
        return;
    }
    hipLaunchKernelGGL(left_factor_kernel, dim3((unsigned)(((size_t)N * r + LR_THREADS - 1) / LR_THREADS), (unsigned)nc),
                       dim3(LR_THREADS), 0, stream, Y, inv, N, r, M, ch, accept, accept_dev);
    hipLaunchKernelGGL(rank_update_kernel, dim3(tiles, tiles, (unsigned)nc), dim3(LR_THREADS),
                       (size_t)2 * 64 * r * sizeof(double), stream, K, M, R, N, r, out, ch, accept, accept_dev,
                       (const double *)nullptr, (const double *)nullptr);
}

// Metropolis decision of one tree proposal per chain on the device (bark_sampler.py:256-264):
//   log_alpha = log_q_prior + (new_mll - cur_mll),  new_mll - cur_mll = 0.5 (dquad - dlogdet)   (scalars = {dquad, dlogdet})
//   accept iff log(u) <= min(log_alpha, 0);  on accept the chain's running y'K^-1 y and log|K| move with it.
// accept_out: 1 / 0, or -1 when the r x r system was singular (the reference raises LinAlgError there).  A chain that
// met a singular system stays at -1 for the rest of the sweep (accept_prev = the previous step's flags, null for the
// first): its K_inv is not rewritten again, so K_inv, quad and logdet of that chain still belong together — the state
// after its last accepted step — when the host raises.
__global__ void decide_kernel(const double *__restrict__ scalars, const double *__restrict__ log_q_prior,
                              const double *__restrict__ log_u, const int *__restrict__ flags, size_t flag_stride_ints,
                              int nc, double *__restrict__ state, const int32_t *__restrict__ accept_prev,
                              int32_t *__restrict__ accept_out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nc) return;
    const double dquad = scalars[2 * b], dlogdet = scalars[2 * b + 1];
    int acc;
    if (flags[(size_t)b * flag_stride_ints] != 0 || (accept_prev && accept_prev[b] < 0)) {
        acc = -1;
    } else {
        const double log_alpha = log_q_prior[b] + 0.5 * (dquad - dlogdet);
        acc = (log_u[b] <= fmin(log_alpha, 0.0)) ? 1 : 0;  // NaN compares false: reject
    }
    accept_out[b] = acc;
    if (acc > 0) {
        state[2 * b] = state[2 * b] - dquad;
        state[2 * b + 1] = state[2 * b + 1] + dlogdet;
    }
}

void launch_skinny_t(hipStream_t stream, const double *K, const double *U, int N, int r, double *out) {
    const dim3 grid((unsigned)((N + LR_THREADS / 64 - 1) / (LR_THREADS / 64)));
    const size_t lds = (size_t)64 * r * sizeof(double);
    if (r <= 8)
        hipLaunchKernelGGL((skinny_t_kernel<8>), grid, dim3(LR_THREADS), lds, stream, K, U, N, r, out);
    else if (r <= 16)
        hipLaunchKernelGGL((skinny_t_kernel<16>), grid, dim3(LR_THREADS), lds, stream, K, U, N, r, out);
    else if (r <= 32)
        hipLaunchKernelGGL((skinny_t_kernel<32>), grid, dim3(LR_THREADS), lds, stream, K, U, N, r, out);
    else
        hipLaunchKernelGGL((skinny_t_kernel<64>), grid, dim3(LR_THREADS), lds, stream, K, U, N, r, out);
}

// workspace: Y (N r) | R (N r) | M (N r) | den^-1 (r r) | sums (r r + r) | shares (<= N/4 blocks x (r r + r)) |
//            P (segments x N r, column form only) | flag
struct LowRankWs {
    double *Y, *R, *M, *inv, *sums, *partial, *P;
    int *flag;
    size_t bytes;
};

LowRankWs lowrank_ws(void *base, int64_t N, int64_t r) {
    LowRankWs w;
    const int64_t max_blocks = (N + 3) / 4, per = r * r + r;
    w.Y = static_cast<double *>(base);
    w.R = w.Y + N * r;
    w.M = w.R + N * r;
    w.inv = w.M + N * r;
    w.sums = w.inv + r * r;
    w.partial = w.sums + per;
    const int64_t seg_rows = colsum_segment(N);
    const int64_t p_doubles = colsum_usable(N, r) ? (N + seg_rows - 1) / seg_rows * N * r : 0;
    w.P = w.partial + (size_t)max_blocks * per;
    w.flag = reinterpret_cast<int *>(w.P + p_doubles);
    w.bytes = (size_t)(3 * N * r + r * r + per + max_blocks * per + p_doubles) * sizeof(double) + 64;
    return w;
}

}  // namespace
}  // namespace bark

using namespace bark;

extern "C" {

size_t bark_lowrank_workspace_bytes(int64_t N, int64_t r) {
    if (N < 1 || r < 1 || r > LR_MAX) return 0;
    return lowrank_ws(nullptr, N, r).bytes;
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
    const LowRankWs w = lowrank_ws(workspace, N, r);
    const int r_neg = subtract ? (int)r : 0;
    const bool colform = colsum_usable(N, r);
    const int nblocks = (symmetric && colform)
                            ? launch_colsum(stream, K_inv, U, (int)N, (int)r, w.P, w.Y, nullptr, w.partial)
                            : launch_skinny(stream, K_inv, U, (int)N, (int)r, w.Y, nullptr, w.partial);
    BARK_LAUNCH_CHECK();
    const double *Rp = w.Y;
    if (K_out && !symmetric) {  // right factor U' K_inv as its own pass
        if (colform)
            launch_colsum(stream, K_inv, U, (int)N, (int)r, w.P, w.R, nullptr, nullptr);
        else
            launch_skinny_t(stream, K_inv, U, (int)N, (int)r, w.R);
        BARK_LAUNCH_CHECK();
        Rp = w.R;
    }
    BARK_HIP_CHECK(hipMemsetAsync(w.flag, 0, sizeof(int), stream));
    launch_small(stream, w.partial, nblocks, (int)r, one_int(r_neg), w.sums, K_out ? w.inv : nullptr, logabsdet_out, w.flag,
                 nullptr);
    BARK_LAUNCH_CHECK();
    if (K_out) {
        launch_rewrite(stream, K_inv, (int)N, (int)r, w.Y, w.inv, w.M, Rp, K_out, one_int(1));
        BARK_LAUNCH_CHECK();
    }
    return BARK_OK;
}

// Reads the `singular` flag the last update / swap evaluation left in `workspace` (synchronises `stream`):
// *singular_out = 1-based column of the first exactly-zero pivot of (mul I + U' K_inv U), or 0.  The reference's
// np.linalg.solve / slogdet raise LinAlgError there (quick_inverse.py:19,31); the Python layer does the same.
int bark_lowrank_status_hip(void *workspace, int64_t N, int64_t r, int32_t *singular_out, void *stream_) {
    error_buffer()[0] = 0;
    if (!workspace || !singular_out || N < 1 || r < 1 || r > LR_MAX) return fail(BARK_ERR_ARG, "bark_lowrank_status_hip: bad argument");
    const LowRankWs w = lowrank_ws(workspace, N, r);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    BARK_HIP_CHECK(hipMemcpyAsync(singular_out, w.flag, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
    BARK_HIP_CHECK(hipStreamSynchronize(stream));
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
    const LowRankWs w = lowrank_ws(workspace, N, r);
    const int nblocks = colsum_usable(N, r) ? launch_colsum(stream, K_inv, U, (int)N, (int)r, w.P, w.Y, y, w.partial)
                                            : launch_skinny(stream, K_inv, U, (int)N, (int)r, w.Y, y, w.partial);
    BARK_LAUNCH_CHECK();
    BARK_HIP_CHECK(hipMemsetAsync(w.flag, 0, sizeof(int), stream));
    launch_small(stream, w.partial, nblocks, (int)r, one_int((int)r_old), w.sums, w.inv, scalars_out + 1, w.flag, scalars_out);
    BARK_LAUNCH_CHECK();
    return BARK_OK;
}

int bark_lowrank_swap_apply_hip(const double *K_inv, int64_t N, int64_t r, void *workspace, double *K_out,
                                void *stream_) {
    error_buffer()[0] = 0;
    if (!K_inv || !K_out || !workspace || N < 1 || r < 1 || r > LR_MAX)
        return fail(BARK_ERR_ARG, "bark_lowrank_swap_apply_hip: bad argument");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const LowRankWs w = lowrank_ws(workspace, N, r);
    launch_rewrite(stream, K_inv, (int)N, (int)r, w.Y, w.inv, w.M, w.Y, K_out, one_int(1));
    BARK_LAUNCH_CHECK();
    return BARK_OK;
}

// ---- the same proposal evaluated straight from the two trees ------------------------------------
// `packed` is the device wire format of ONE forest of TWO trees [old tree, new tree] (bark_forest_pack with
// B = 1, m = 2).  The one-hot leaf code of that pair IS the matrix [U_old U_new] / s of the reference's
// get_leaf_vectors (forest.py:70-75) with one column per leaf of each tree: leaves no training point reaches
// give zero columns, which leave log|det(C + G)| and v'(C + G)^-1 v unchanged.  r_old = leaves of the old tree
// (bark_forest_pack_info of that tree alone: max_bits), s = sqrt(scale / m) (bark_sampler.py:233-236).
// The workspace starts with the bark_lowrank_swap_eval_hip layout, so bark_lowrank_swap_apply_hip(K_inv, N,
// info->max_bits, workspace, ...) commits an accepted proposal.
size_t bark_tree_swap_workspace_bytes(int64_t N, int64_t r) {
    if (N < 1 || r < 1 || r > LR_MAX) return 0;
    const size_t base = round_up((int64_t)lowrank_ws(nullptr, N, r).bytes, 256);
    const size_t codes = (size_t)((r + 31) / 32) * (size_t)bark_leaf_npad(N) * sizeof(uint32_t);
    return base + round_up((int64_t)((size_t)N * r * sizeof(double)), 256) + codes;
}

int bark_tree_swap_eval_hip(bark_ctx *ctx, const double *K_inv, int64_t N, const void *packed, const bark_pack_info *info,
                            const double *X, int64_t d, int64_t r_old, double s, const double *y, double *scalars_out,
                            void *workspace, size_t workspace_bytes, void *stream_) {
    error_buffer()[0] = 0;
    {
        const int crc = check_ctx(ctx);
        if (crc) return crc;
    }
    if (!K_inv || !packed || !info || !X || !y || !scalars_out || !workspace || N < 1 || d < 1 || N > (1 << 24))
        return fail(BARK_ERR_ARG, "bark_tree_swap_eval_hip: bad argument");
    if (info->B != 1 || info->m != 2)
        return fail(BARK_ERR_ARG, "bark_tree_swap_eval_hip: pack the pair [old tree, new tree] as one forest (B = 1, m = 2)");
    const int64_t r = info->max_bits;
    if (r < 2 || r > LR_MAX) return fail(BARK_ERR_ARG, "tree swap supports 2..%d leaves in total (got %lld)", LR_MAX, (long long)r);
    if (r_old < 1 || r_old >= r) return fail(BARK_ERR_ARG, "r_old = %lld is not inside (0, %lld)", (long long)r_old, (long long)r);
    if (workspace_bytes < bark_tree_swap_workspace_bytes(N, r)) return fail(BARK_ERR_WORKSPACE, "tree-swap workspace too small");
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    char *ws = static_cast<char *>(workspace);
    const size_t base = round_up((int64_t)lowrank_ws(nullptr, N, r).bytes, 256);
    double *U = reinterpret_cast<double *>(ws + base);
    uint32_t *codes = reinterpret_cast<uint32_t *>(ws + base + round_up((int64_t)((size_t)N * r * sizeof(double)), 256));
    const int words = (int)((r + 31) / 32);
    int rc = walk_one_hot(packed, info, X, N, d, words, codes, ctx->fault, stream);
    if (rc) return rc;
    ChainDoubles scales = {};
    scales.v[0] = s;
    hipLaunchKernelGGL(expand_onehot_kernel, dim3((unsigned)((N * r + LR_THREADS - 1) / LR_THREADS)), dim3(LR_THREADS), 0,
                       stream, codes, words, (int)bark_leaf_npad(N), (int)N, (int)r, scales, U, Chain{0, 0}, (int *)nullptr);
    BARK_LAUNCH_CHECK();
    return bark_lowrank_swap_eval_hip(K_inv, N, U, r_old, r - r_old, y, scalars_out, workspace, base, stream_);
}

// ---- several chains in one call ---------------------------------------------------------------------
// The chains of the sampler are independent (bark_sampler.py:147 loops over them); one proposal is a short,
// latency-bound launch sequence around a single streaming pass.  Each chain's sequence is enqueued on its own
// stream forked from the caller's, so the sequences overlap on the GPU and the host pays one call and one
// synchronisation for all of them.  K_inv: (nc, N, N); packed/info: nc forests of TWO trees [old, new]
// (bark_forest_pack, B = nc, m = 2); r_old, s: host arrays (nc); scalars_out: device (nc, 2); workspace of
// bark_tree_swap_chains_workspace_bytes(N, info->max_bits, nc) bytes.  With N even and <= 16 leaves per pair the
// chain index is a grid dimension of every kernel (one launch sequence for all chains); otherwise one
// single-chain sequence per chain, each on its own stream.
size_t bark_tree_swap_chains_workspace_bytes(int64_t N, int64_t r, int64_t nc, size_t *chain_stride_bytes) {
    if (N < 1 || r < 1 || r > LR_MAX || nc < 1 || nc > MAX_CHAINS) return 0;
    const size_t stride = (size_t)round_up((int64_t)bark_tree_swap_workspace_bytes(N, r), 256);
    if (chain_stride_bytes) *chain_stride_bytes = stride;
    // nc per-chain blocks, then the one-hot codes of all chains, contiguous (one leaf walk for all of them)
    return stride * (size_t)nc + (size_t)nc * (size_t)((r + 31) / 32) * (size_t)bark_leaf_npad(N) * sizeof(uint32_t);
}

}  // extern "C"

// One proposal per chain: leaf walk of the nc [old, new] pairs, U, Y = K_inv U, the r x r algebra -> scalars_out
// (nc, 2).  Chain b's workspace block starts `stride` bytes after chain b-1's (>= the block size for this r); the
// codes of all chains follow at ws + stride * nc.
// dec != nullptr (bark_tree_sweep_chains_hip, column form): the Metropolis decision is taken inside small_kernel — *decided = true
static int eval_chains(bark_ctx *ctx, const double *K_inv, int64_t N, int64_t nc, const void *packed, const bark_pack_info *info,
                       const double *X, int64_t d, const int64_t *r_old, const double *s, const double *y,
                       double *scalars_out, char *ws, size_t stride, hipStream_t caller, const DecideArgs *dec = nullptr,
                       bool *decided = nullptr) {
    const int64_t r = info->max_bits;
    if (colsum_usable(N, r)) {
        // one launch sequence for all chains: the chain index is a grid dimension of every kernel
        const Chain ch{(size_t)N * (size_t)N, stride / sizeof(double)};
        const LowRankWs w = lowrank_ws(ws, N, r);
        const size_t base = (size_t)round_up((int64_t)w.bytes, 256);
        double *U = reinterpret_cast<double *>(ws + base);
        uint32_t *codes = reinterpret_cast<uint32_t *>(ws + stride * (size_t)nc);
        const int words = (int)((r + 31) / 32);
        int rc = walk_one_hot(packed, info, X, N, d, words, codes, ctx->fault, caller);
        if (rc) return rc;
        ChainDoubles scales = {};
        ChainInts r_negs = {};
        for (int64_t b = 0; b < nc; ++b) {
            scales.v[b] = s[b];
            r_negs.v[b] = (int)r_old[b];
        }
        // r <= 16 here (colsum_usable): the one-hot U is one code word per point — read by the kernels, never materialised
        const OneHot oh{codes, (int)bark_leaf_npad(N), scales, w.flag};
        const int nblocks = launch_colsum(caller, K_inv, U, (int)N, (int)r, w.P, w.Y, y, w.partial, (int)nc, ch, &oh);
        BARK_LAUNCH_CHECK();
        launch_small(caller, w.partial, nblocks, (int)r, r_negs, w.sums, w.inv, scalars_out + 1, w.flag, scalars_out, (int)nc,
                     ch, 2, dec);
        BARK_LAUNCH_CHECK();
        if (decided) *decided = dec != nullptr;
        return BARK_OK;
    }
    // general shapes (odd N, more than 16 leaves): one single-chain sequence per chain, each on its own stream
    int rc = ctx_chain_streams(ctx, (size_t)nc);
    if (rc) return rc;
    BARK_HIP_CHECK(hipEventRecord(ctx->chain_fork, caller));
    bark_pack_info one = *info;
    one.B = 1;
    for (int64_t b = 0; b < nc; ++b) {
        hipStream_t st = ctx->chain_streams[(size_t)b];
        BARK_HIP_CHECK(hipStreamWaitEvent(st, ctx->chain_fork, 0));
        rc = bark_tree_swap_eval_hip(ctx, K_inv + (size_t)b * N * N, N,
                                     static_cast<const char *>(packed) + (size_t)b * 2 * info->stride * 16, &one, X, d,
                                     r_old[b], s[b], y, scalars_out + 2 * b, ws + (size_t)b * stride, stride, st);
        // the join is recorded even after a failure so that the caller's stream never waits on nothing
        BARK_HIP_CHECK(hipEventRecord(ctx->chain_done[(size_t)b], st));
        BARK_HIP_CHECK(hipStreamWaitEvent(caller, ctx->chain_done[(size_t)b], 0));
        if (rc) return rc;
    }
    return BARK_OK;
}

extern "C" {

int bark_tree_swap_eval_chains_hip(bark_ctx *ctx, const double *K_inv, int64_t N, int64_t nc, const void *packed,
                                   const bark_pack_info *info, const double *X, int64_t d, const int64_t *r_old,
                                   const double *s, const double *y, double *scalars_out, void *workspace,
                                   size_t workspace_bytes, void *stream_) {
    error_buffer()[0] = 0;
    {
        const int crc = check_ctx(ctx);
        if (crc) return crc;
    }
    if (!K_inv || !packed || !info || !X || !r_old || !s || !y || !scalars_out || !workspace || N < 1 || d < 1 || nc < 1 ||
        nc > MAX_CHAINS || N > (1 << 24))
        return fail(BARK_ERR_ARG, "bark_tree_swap_eval_chains_hip: bad argument (1 <= chains <= %d)", MAX_CHAINS);
    if (info->B != nc || info->m != 2) return fail(BARK_ERR_ARG, "pack one [old tree, new tree] pair per chain (B = chains, m = 2)");
    const int64_t r = info->max_bits;
    if (r < 2 || r > LR_MAX) return fail(BARK_ERR_ARG, "tree swap supports 2..%d leaves in total (got %lld)", LR_MAX, (long long)r);
    for (int64_t b = 0; b < nc; ++b)
        if (r_old[b] < 1 || r_old[b] >= r) return fail(BARK_ERR_ARG, "chain %lld: r_old = %lld is not inside (0, %lld)", (long long)b, (long long)r_old[b], (long long)r);
    size_t stride = 0;
    if (workspace_bytes < bark_tree_swap_chains_workspace_bytes(N, r, nc, &stride))
        return fail(BARK_ERR_WORKSPACE, "tree-swap workspace too small for %lld chains", (long long)nc);
    return eval_chains(ctx, K_inv, N, nc, packed, info, X, d, r_old, s, y, scalars_out, static_cast<char *>(workspace), stride,
                       static_cast<hipStream_t>(stream_));
}

// One sweep over the trees of nc chains with the Metropolis decision taken on the device — the per-tree loop of
// bark_sampler.py:233-264 without a host round trip per tree.  Step t (t < n_steps) swaps one tree per chain:
//   packed + packed_offsets[t], infos[t]   wire format of the nc pairs [old tree, new tree] of that step (B = nc, m = 2)
//   r_old[t * nc + b]                      leaves of chain b's old tree (HOST)
//   log_q_prior, log_u                     (n_steps, nc) DEVICE: proposal ratio and log of the uniform draw
// For every step: evaluate (as bark_tree_swap_eval_chains_hip), decide (decide_kernel), rewrite K_inv[b] for the
// accepted chains.  state (nc, 2) DEVICE holds y'K^-1 y and log|K| of every chain on entry and exit;
// accept_out (n_steps, nc) DEVICE int32: 1 accepted, 0 rejected, -1 singular system.  Nothing synchronises: the host
// reads accept_out and state once per sweep.  workspace: bark_tree_swap_chains_workspace_bytes(N, r_max, nc) bytes with
// r_max = max_t infos[t].max_bits, plus 16 * nc bytes for the step's scalars.
int bark_tree_sweep_chains_hip(bark_ctx *ctx, double *K_inv, int64_t N, int64_t nc, int64_t n_steps, const void *packed,
                               const int64_t *packed_offsets, const bark_pack_info *infos, const double *X, int64_t d,
                               const int64_t *r_old, const double *s, const double *y, const double *log_q_prior,
                               const double *log_u, double *state, int32_t *accept_out, void *workspace,
                               size_t workspace_bytes, void *stream_) {
    error_buffer()[0] = 0;
    {
        const int crc = check_ctx(ctx);
        if (crc) return crc;
    }
    if (!K_inv || !packed || !packed_offsets || !infos || !X || !r_old || !s || !y || !log_q_prior || !log_u || !state ||
        !accept_out || !workspace || N < 1 || d < 1 || nc < 1 || nc > MAX_CHAINS || n_steps < 1 || N > (1 << 24))
        return fail(BARK_ERR_ARG, "bark_tree_sweep_chains_hip: bad argument (1 <= chains <= %d)", MAX_CHAINS);
    int64_t r_max = 0;
    for (int64_t t = 0; t < n_steps; ++t) {
        const bark_pack_info &in = infos[t];
        if (in.B != nc || in.m != 2) return fail(BARK_ERR_ARG, "step %lld: pack one [old, new] pair per chain (B = chains, m = 2)", (long long)t);
        if (in.max_bits < 2 || in.max_bits > LR_MAX)
            return fail(BARK_ERR_ARG, "step %lld: tree swap supports 2..%d leaves in total (got %lld)", (long long)t, LR_MAX, (long long)in.max_bits);
        for (int64_t b = 0; b < nc; ++b)
            if (r_old[t * nc + b] < 1 || r_old[t * nc + b] >= in.max_bits)
                return fail(BARK_ERR_ARG, "step %lld chain %lld: r_old out of range", (long long)t, (long long)b);
        if (in.max_bits > r_max) r_max = in.max_bits;
    }
    size_t stride = 0;
    const size_t need = bark_tree_swap_chains_workspace_bytes(N, r_max, nc, &stride);
    if (workspace_bytes < need + 16 * (size_t)nc) return fail(BARK_ERR_WORKSPACE, "tree-sweep workspace too small");
    hipStream_t caller = static_cast<hipStream_t>(stream_);
    char *ws = static_cast<char *>(workspace);
    double *scalars = reinterpret_cast<double *>(ws + need);
    const Chain ch{(size_t)N * (size_t)N, stride / sizeof(double)};
    for (int64_t t = 0; t < n_steps; ++t) {
        const bark_pack_info &in = infos[t];
        const int64_t r = in.max_bits;
        // Up to N ~ 1000 a sweep is bound by the host's launch rate (~7 us per launch, whatever N): six launches per tree proposal
        // instead of ten — the flag memset in expand_onehot_kernel, shares + Metropolis decision in small_kernel, M = Y den^-1
        // in rank_update_kernel (50 proposals at N = 128 .. 1000: 3.7 -> 2.3 ms).
        int32_t *acc = accept_out + t * nc;
        const DecideArgs dec{log_q_prior + t * nc, log_u + t * nc, state, t > 0 ? acc - nc : nullptr, acc};
        bool decided = false;
        int rc = eval_chains(ctx, K_inv, N, nc, static_cast<const char *>(packed) + packed_offsets[t], &in, X, d, r_old + t * nc, s,
                             y, scalars, ws, stride, caller, &dec, &decided);
        if (rc) return rc;
        const LowRankWs w = lowrank_ws(ws, N, r);
        if (!decided) {  // general shapes (per-chain streams): the decision as a launch of its own
            hipLaunchKernelGGL(decide_kernel, dim3((unsigned)((nc + 63) / 64)), dim3(64), 0, caller, scalars, log_q_prior + t * nc,
                               log_u + t * nc, w.flag, stride / sizeof(int), (int)nc, state, t > 0 ? acc - nc : nullptr, acc);
            BARK_LAUNCH_CHECK();
        }
        launch_rewrite(caller, K_inv, (int)N, (int)r, w.Y, w.inv, w.M, w.Y, K_inv, ChainInts{}, (int)nc, ch, acc, decided);
        BARK_LAUNCH_CHECK();
    }
    return BARK_OK;
}

// accept[b] != 0: K_inv[b] <- K_inv[b] - Y (C+G)^-1 Y' from chain b's workspace block (host array, nc entries)
int bark_lowrank_swap_apply_chains_hip(double *K_inv, int64_t N, int64_t nc, int64_t r, const int32_t *accept,
                                       void *workspace, size_t workspace_bytes, void *stream_) {
    error_buffer()[0] = 0;
    if (!K_inv || !accept || !workspace || N < 1 || nc < 1 || nc > MAX_CHAINS || r < 1 || r > LR_MAX)
        return fail(BARK_ERR_ARG, "bark_lowrank_swap_apply_chains_hip: bad argument");
    size_t stride = 0;
    if (workspace_bytes < bark_tree_swap_chains_workspace_bytes(N, r, nc, &stride))
        return fail(BARK_ERR_WORKSPACE, "tree-swap workspace too small for %lld chains", (long long)nc);
    hipStream_t caller = static_cast<hipStream_t>(stream_);
    const Chain ch{(size_t)N * (size_t)N, stride / sizeof(double)};
    const LowRankWs w = lowrank_ws(workspace, N, r);
    ChainInts acc = {};
    bool any = false;
    for (int64_t b = 0; b < nc; ++b) {
        acc.v[b] = accept[b] != 0;
        any = any || acc.v[b];
    }
    if (!any) return BARK_OK;
    launch_rewrite(caller, K_inv, (int)N, (int)r, w.Y, w.inv, w.M, w.Y, K_inv, acc, (int)nc, ch);
    BARK_LAUNCH_CHECK();
    return BARK_OK;
}

}  // extern "C"
