/*
 * bark_hip.h — C ABI of libbarkhip.so: the MI355X (gfx950) forest-kernel Gram + GP
 * marginal-log-likelihood / posterior engine behind the `bark.forest`,
 * `bark.tree_kernels` and `bark.fitting` Python API of TobyBoyne/bark.
 *
 * The reference has no FFI layer (it is Python + numba); its boundary for this path is a
 * set of module-level Python functions taking numpy arrays.  Each entry point below names
 * the reference function (path:line under /root/reference/src/bark) it serves.  The
 * binding a maintainer adds on the reference side is a ctypes stub — see INTEGRATION.md.
 *
 * Conventions
 *  - plain C types only; `int` status return, 0 == BARK_OK; message via bark_last_error()
 *    (thread-local).  The library never aborts and never frees/retains caller buffers.
 *  - `*_hip` entry points take DEVICE pointers (e.g. torch `tensor.data_ptr()`) and a
 *    `hipStream_t` passed as `void*` (NULL = default stream).  They only enqueue work:
 *    no allocation, no host synchronisation, so they are hipGraph-capturable (after one
 *    un-captured call with the same context, which creates its helper streams and events).
 *  - entry points that walk trees or fork helper streams take a `bark_ctx*` first: the per-device context that owns
 *    those streams/events, a scratch buffer and the categorical-fault flag.  The library is re-entrant per context:
 *    host threads that use one context (and one stream) each share no mutable state (SURVEY §8b Ownership /
 *    Threading; the reference's callers are single-threaded Python, bark_sampler.py / tree_gps.py).
 *  - `*_pack*` entry points are HOST functions on host pointers (format conversion only).
 *  - all matrices are row-major float64; feature matrices X are (N, d) row-major float64.
 */
#ifndef BARK_HIP_H
#define BARK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BARK_HIP_VERSION 210 /* 0.2.1: host-pointer entry points (bark_dev_alloc, bark_ctx_upload, ..._host_pair) */

enum {
    BARK_OK = 0,
    BARK_ERR_ARG = 1,         /* bad shape / null pointer / unsupported size */
    BARK_ERR_TREE = 2,        /* malformed tree: child index >= node_limit, cycle, feature_idx >= d */
    BARK_ERR_CATEGORICAL = 3, /* categorical threshold is not a bitmask in [0, 2^32) */
    BARK_ERR_HIP = 4,         /* a HIP runtime call failed (launch error etc.) */
    BARK_ERR_WORKSPACE = 5    /* workspace too small */
};

/* MLL conventions (bit flags) */
enum {
    BARK_MLL_INCLUDE_SCALE = 1, /* K_s = scale*K + (1e-6+noise) I   (bark_sampler.py:153-156) */
    BARK_MLL_INCLUDE_2PI = 2,   /* subtract n*log(2*pi)             (examples/mcmc/mcmc_record_mll.py:73) */
    BARK_MLL_RHS_IDENTITY = 4   /* right-hand-side block = I (C must equal N, cand ignored): cov_out receives
                                   K_s^-1 and mu_out receives K_s^-1 y  (bark/optimizer/opt_model.py:54-59,101) */
};

int bark_version(void);
/* Last error message of the calling thread ("" if none).  Pointer valid until the next call. */
const char *bark_last_error(void);
/* Process-wide switch (returns the previous setting; default on, off when $BARK_NO_DEVICE_WAIT is set).  On: sweeps of
 * few resident matrices — bound by the dependent chain diag -> solve -> diag (bark_sampler.py:153-162 with one chain,
 * BASELINE configs c2 / c5) — hand the completion of their row launches over to the caller's stream through a
 * device-side progress counter that the diagonal-block kernel waits for (bounded: 2 s), instead of a cross-stream
 * event wait between two kernels of that stream.  It needs the library's helper streams to run beside the caller's;
 * where they cannot (every stream of the process serialised onto one hardware queue), the wait times out and the call
 * reports info_out[b] = -3: DRAIN THE DEVICE (hipDeviceSynchronize — the entry point has joined its helper streams into
 * the caller's stream, so synchronising that stream is enough), switch the mechanism off and call again; the results and the
 * workspace of the timed-out call are not valid.  A time-out is sticky within a chunk (every later wait gives up at once:
 * the call costs one 2 s bound, not one per block step).  Off by default when $AMD_SERIALIZE_KERNEL / $HIP_LAUNCH_BLOCKING
 * serialise dispatch.  Never used under stream capture. */
int bark_device_wait(int on);
/* Host-side self-check of the workgroup -> (matrix, tile) map the sweep kernels share (XCD-aware placement: speed only, but a
 * map that skipped or doubled a pair would be a wrong result): 0 when the launch grid of `ntiles` tiles x `Bc` matrices
 * reaches every pair exactly once, else the number of pairs missed or reached twice.  No GPU needed. */
int bark_xcd_map_selftest(int ntiles, int Bc);

/* ---------------------------------------------------------------------------------------
 * Context — replaces nothing in the reference (pure functions on numpy arrays, forest.py:58-111); it is where the
 * state a GPU implementation needs between calls lives instead of in process globals.
 * ------------------------------------------------------------------------------------- */
typedef struct bark_ctx bark_ctx;
/* One context per (host thread, device).  Does not change the current device. */
int bark_ctx_create(int device, bark_ctx **out);
/* Frees the helper streams, events, scratch and flag.  Synchronise the streams used with the context first. */
void bark_ctx_destroy(bark_ctx *ctx);
/* Grow-only, 256-byte aligned device scratch owned by the context (the `workspace` arguments below may point into
 * it, or at any caller-owned device buffer).  *ptr_out stays valid until a later call asks for more bytes. */
int bark_ctx_workspace(bark_ctx *ctx, size_t bytes, void **ptr_out);
size_t bark_ctx_workspace_bytes(const bark_ctx *ctx);
/* Reads and clears the categorical-fault flag (synchronises `stream`).  *cat_fault_out != 0: a leaf walk enqueued with
 * this context evaluated a categorical split on a NaN / inf / negative value, where the reference raises inside
 * `1 << int(x)` (forest.py:38).  The MLL / posterior entry points report the same condition as info_out[b] = -1. */
int bark_ctx_status(bark_ctx *ctx, void *stream, int32_t *cat_fault_out);

/* ---------------------------------------------------------------------------------------
 * Host-pointer entry points — the ABI for a caller that has no tensor library: the reference's sampler is numba nopython
 * code (bark_sampler.py:120 `@njit _run_bark_sampler_multichain`, :216 `@njit _step_bark_sampler`), which can call C through
 * ctypes function pointers with integers, floats and array addresses only.  With these it can hold its chain state in HBM
 * (bark_sampler.py:153-162: K_inv by bark_mll_batched_hip + BARK_MLL_RHS_IDENTITY) and evaluate every tree proposal
 * (bark_sampler.py:233-257) from the two host trees it already has.  INTEGRATION.md §4 shows the caller.
 * ------------------------------------------------------------------------------------- */
/* hipMalloc / hipFree on the context's device (bark_dev_free synchronises the device, as hipFree does). */
int bark_dev_alloc(bark_ctx *ctx, size_t bytes, void **ptr_out);
int bark_dev_free(bark_ctx *ctx, void *ptr);
/* Host -> device on `stream`; `src_host` may be reused when the call returns (blocks of up to 64 KiB are staged through the
 * context's pinned page and copied asynchronously, larger ones are copied synchronously).  The three staged entry points of a
 * context (this one, bark_ctx_download, bark_tree_swap_eval_host_pair) share that page and may be given different streams:
 * each waits for the event recorded behind the previous staged copy, on whatever stream it went to, before it touches the
 * page.  One host thread per context, as everywhere. */
int bark_ctx_upload(bark_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes, void *stream);
/* Device -> host; synchronises `stream`. */
int bark_ctx_download(bark_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes, void *stream);
int bark_stream_sync(bark_ctx *ctx, void *stream);
/* One tree proposal of `_step_bark_sampler` (bark_sampler.py:233-257) from the trees as the sampler holds them:
 * pair26 = [forest[tree_idx], new_nodes], 2 x L packed 26-byte records in HOST memory; feat_types HOST int64 (d,);
 * K_inv (N, N), X (N, d), y (N,) DEVICE; s = sqrt(scale / m) (bark_sampler.py:231).  Packs the pair, stages it through the
 * context's pinned page, runs bark_tree_swap_eval_hip and returns on the HOST
 *   scalars_host_out[0] = y'K^-1 y - y'K'^-1 y ,  scalars_host_out[1] = log|K'| - log|K| ,  *r_out = leaves of the pair,
 * so new_mll = 0.5 (-(quad - scalars[0]) - (logdet + scalars[1])) (quick_inverse.py:38).  An accepted proposal is committed by
 * bark_lowrank_swap_apply_hip(K_inv, N, *r_out, workspace, K_inv, stream).  workspace >= bark_tree_swap_workspace_bytes(N, 64)
 * covers every pair this entry point accepts (more than 64 leaves in the pair: BARK_ERR_ARG).  Synchronises `stream`.
 * NaN / inf scalars: ask bark_lowrank_status_hip (singular update, quick_inverse.py:19,31) and bark_ctx_status. */
int bark_tree_swap_eval_host_pair(bark_ctx *ctx, const double *K_inv, int64_t N, const void *pair26, int64_t L,
                                  const int64_t *feat_types, int64_t d, const double *X, double s, const double *y,
                                  double *scalars_host_out, int64_t *r_out, void *workspace, size_t workspace_bytes,
                                  void *stream);

/* ---------------------------------------------------------------------------------------
 * Forest container  (forest.py:8-19 NODE_RECORD_DTYPE: packed 26-byte records)
 *
 * bark_forest_pack converts B*m trees of L packed 26-byte node records (host memory,
 * C-contiguous (B, m, L)) into the device wire format: per tree `stride` 16-byte nodes in
 * depth-first order (root = slot 0), containing only the nodes reachable from the root.
 *   node.w0  internal: feature_idx | (is_categorical << 30);   leaf: 0x80000000 | dense_leaf_id
 *   node.w1  internal: float32 threshold bits, or the uint32 category bitmask;  leaf: original node index
 *   node.w2/w3  internal: compact index of left/right child;  leaf: w2 = position of the leaf's bit in the
 *               forest's one-hot code (trees own consecutive bit fields, one bit per reachable leaf)
 * `feat_types` is the reference's int64 array (0 = Cat, 1 = Int, 2 = Cont; forest.py:22-25).
 * ------------------------------------------------------------------------------------- */
typedef struct {
    int64_t B, m, L;      /* as passed in */
    int64_t stride;       /* nodes per tree in the packed array (max reachable nodes, >= 1) */
    int64_t max_leaves;   /* max number of reachable leaves in any tree (dense ids < max_leaves) */
    int64_t max_depth;    /* longest root-to-leaf walk (edges) */
    int64_t packed_bytes; /* B*m*stride*16 */
    int64_t max_bits;     /* max over forests of sum_t (#reachable leaves of tree t): width of the one-hot code */
} bark_pack_info;

/* Pass 1: validate + measure.  Fills *info. */
int bark_forest_pack_info(const void *nodes26, int64_t B, int64_t m, int64_t L, const int64_t *feat_types,
                          int64_t d, bark_pack_info *info);
/* Pass 2: write info->packed_bytes bytes to `packed` (host). */
int bark_forest_pack(const void *nodes26, const int64_t *feat_types, int64_t d, const bark_pack_info *info,
                     void *packed);

/* ---------------------------------------------------------------------------------------
 * Leaf traversal   — forest.py:28-67 (_pass_one_through_tree / pass_through_forest)
 * ------------------------------------------------------------------------------------- */
/* (N, m) uint32 leaf NODE indices for each of the B forests: out is (B, N, m) uint32, C order —
 * bit-identical to pass_through_forest(nodes[b], X, feat_types). */
int bark_leaf_indices_hip(bark_ctx *ctx, const void *packed, const bark_pack_info *info, const double *X, int64_t N,
                          int64_t d, uint32_t *out, void *stream);

/* Leaf codes consumed by the Gram kernels: out is (B, W, Npad) uint32, W = bark_leaf_words(info),
 * Npad = bark_leaf_npad(N), point index fastest.  Two encodings, chosen from `info` alone
 * (bark_leaf_encoding): 
 *   BARK_LEAF_BITS  one-hot: tree t owns L_t consecutive bits (L_t = its reachable leaves), a point sets the
 *                   bit of the leaf it reaches in every tree; then  #agreeing trees = popcount(z_i & z_j)
 *                   (2 VALU per 32 bits).  Chosen when max_bits < 64 * ceil(m/4), and always for trees with
 *                   more than 256 leaves.
 *   BARK_LEAF_BYTES 4 dense 8-bit leaf ids per dword; a tree pair agrees iff its byte of z_i ^ z_j is zero. */
enum { BARK_LEAF_BYTES = 0, BARK_LEAF_BITS = 1 };
int64_t bark_leaf_npad(int64_t N);
int bark_leaf_encoding(const bark_pack_info *info);
int64_t bark_leaf_words(const bark_pack_info *info);
int bark_leaf_codes_hip(bark_ctx *ctx, const void *packed, const bark_pack_info *info, const double *X, int64_t N,
                        int64_t d, uint32_t *out, void *stream);

/* One-hot leaf vectors — forest.py:70-75 get_leaf_vectors: out[i * ldo + c] = value if leaves[i * ldl] == ids[c] else 0
 * (`np.equal(leaves[:, None], all_leaves[None, :])`, optionally scaled as bark_sampler.py:233-236 does).  leaves: leaf
 * node indices of one tree (element stride ldl, e.g. a column of bark_leaf_indices_hip's output); ids: the r distinct
 * reached leaves in ascending order (np.unique on the host: N * 4 bytes). */
int bark_onehot_match_hip(const uint32_t *leaves, int64_t N, int64_t ldl, const uint32_t *ids, int64_t r, double value,
                          double *out, int64_t ldo, void *stream);

/* ---------------------------------------------------------------------------------------
 * Gram matrix   — forest.py:78-111 (forest_gram_matrix / batched_forest_gram_matrix / _no_null)
 *   out[b][i][j] = (1.0/m) * #{t : leaf_t(x1_i) == leaf_t(x2_j)}          (bit-exact)
 * optionally followed, in this order, by
 *   - shift[b]            (forest.py:111  `sim_mat - num_null_trees / num_trees`)
 *   * scale[b]            (forest.py:111 `* scale`; tree_gps.py:97; bark_sampler.py:153)
 *   + (1e-6 + noise[b])   on the diagonal (tree_gps.py:100, mcmc_record_mll.py:67)
 * each rounded separately, as numpy does.  shift / scale / noise may be NULL (skipped).
 * leaf1/leaf2 come from bark_leaf_codes_hip for x1 / x2 with the same `info` (pass the same pointer when
 * x1 is x2).  out is (B, N, ld) float64, row stride `ld` >= M, batch stride `batch_stride` elements.
 * ------------------------------------------------------------------------------------- */
int bark_gram_from_leaves_hip(const uint32_t *leaf1, int64_t N, const uint32_t *leaf2, int64_t M,
                              const bark_pack_info *info, const double *shift, const double *scale,
                              const double *noise, double *out, int64_t ld, int64_t batch_stride, void *stream);

/* ---------------------------------------------------------------------------------------
 * Batched marginal log-likelihood — mcmc_record_mll.py:57-74, bark_sampler.py:153-162 +
 * quick_inverse.py:37-38.  One evaluation per forest sample b:
 *   K_s = [scale_b *] K_b + (1e-6 + noise_b) I ;  mll_b = 0.5(-y' K_s^-1 y - log|K_s| [- n log 2pi])
 * computed by a blocked fp64 Cholesky (A = U'U, MFMA panels) instead of the reference's LU
 * inv + slogdet; agreement is by tolerance (rtol 1e-9, atol 1e-8), see DESIGN.md.
 *
 * Optional posterior predictive (tree_gps.py:80-113) in the same sweep: pass C > 0 candidates
 * and get mu (B, C) and var (B, C) = scale_b - diag(K_xX K_s^-1 K_Xx)   (scale always applied,
 * as forest_predict does).
 *
 * Full covariance (tree_gps.py:108 with diag=False): pass cov_out (B, C, C) to also get
 *   cov_b = scale_b - K_xX K_s^-1 K_Xx     (every entry, as numpy broadcasting does).
 * With BARK_MLL_RHS_IDENTITY the right-hand-side block is the identity instead of K_Xx: cov_out = K_s^-1
 * (no `scale -`), mu_out = K_s^-1 y, var_out = diag(K_s^-1) — the explicit inverse the acquisition builder
 * consumes.  `shift` (B,) or NULL subtracts n_null/m before scaling (forest.py:102-111 no-null kernel).
 *
 * workspace: device buffer of at least bark_mll_workspace_bytes(N, C, m, Bc) bytes, where Bc
 * (1 <= Bc <= B) is the number of forests resident / factorised concurrently; B is processed in chunks of Bc.
 * info_out (device, B int32): 0, or 1-based index of the first non-positive pivot (not PD), or -3 when a device-side
 *   wait timed out (bark_device_wait), or -1 when a leaf walk
 * of the call met an invalid categorical value (see bark_ctx_status).
 * ------------------------------------------------------------------------------------- */
size_t bark_mll_workspace_bytes(int64_t N, int64_t C, int64_t m, int64_t Bc);

/* Which launch schedule bark_mll_batched_hip takes for a shape — replaces nothing in the reference (its `inv` + `slogdet`,
 * examples/mcmc/mcmc_record_mll.py:63-73, have one schedule); a debug / documentation query: the same function the entry
 * point itself configures its sweep from, so DESIGN.md's table of schedules per BASELINE config can be asserted by a test
 * and a tuning constant cannot silently move a shape onto another schedule.  No GPU needed.
 *   leaf_words: bark_leaf_words(info) of the forests (decides whether the Gram is fused into the row kernels);
 *   timing != 0: as a call with a bark_mll_timing (the one-launch evaluation of N <= 128 is not taken then).
 * dev_wait / dev_gate are reported as the process-wide switch stands (bark_device_wait), outside stream capture. */
enum {
    BARK_SCHED_ONE_BLOCK = 0,        /* N <= 128, MLL only: leaf walk + one launch per chunk */
    BARK_SCHED_PLAIN = 1,            /* diag(j) || rows(j), then solve(j) */
    BARK_SCHED_PAIRED = 2,           /* plain with two block rows per row launch (chunks of a multiple of 256 matrices) */
    BARK_SCHED_PIPELINED = 3,        /* rows(j) over k < j-1 two steps ahead; consumers apply the last block row */
    BARK_SCHED_SPLITK = 4,           /* split-K layout (A materialised, slab scratch), no look-ahead step */
    BARK_SCHED_SPLITK_LOOKAHEAD = 5, /* split-K layout with look-ahead bulk launches */
    BARK_SCHED_TWO_BLOCK = 6,        /* 128 < N <= 256, MLL only: leaf walk + one launch per chunk (two block rows in one kernel) */
    BARK_SCHED_MULTI_BLOCK = 7       /* 256 < N <= 768, MLL only: the same with three to six block rows */
};
typedef struct {
    int32_t n_chunks, chunk, last_chunk;  /* chunks of `chunk` matrices, the last one of `last_chunk` */
    int32_t schedule, last_schedule;      /* BARK_SCHED_* of the full chunks / of the last chunk */
    int32_t splitk_layout, fused_gram;    /* slab scratch + materialised A; A generated inside the row kernels */
    int32_t dev_wait, dev_gate, pre_update; /* device-side hand-over, gate kernels, diag_pre_kernel (full chunks) */
    int32_t lookahead_steps, splitk_steps;  /* block steps with a look-ahead bulk / with any split-K launch (full chunks) */
    int32_t nrb, ncb;                     /* block rows, block columns incl. candidate blocks */
} bark_mll_plan;
int bark_mll_plan_query(int64_t N, int64_t C, int64_t m, int64_t B, int64_t Bc, int leaf_words, int timing, bark_mll_plan *out);

typedef struct {
    float total_ms;     /* the whole call on the caller's stream */
    float gram_ms;      /* leaf traversal + Gram fill + rhs init, summed over chunks */
    float chol_ms;      /* factorisation sequence (diag + panel + solve, finish, reductions) = total - gram */
    float diag_ms;      /* of which: diag_kernel  (128x128 potrf + inverse + z_j)            */
    float panel_ms;     /* of which: row_kernel / split-K kernels (fp64 MFMA trailing-panel update, K = 128 j) */
    float solve_ms;     /* of which: solve_kernel (MFMA triangular solve by the block inverse) */
    int64_t n_diag_launches, n_panel_launches, n_solve_launches;
    double panel_flops; /* fp64 flops executed by the row / split-K launches */
    double solve_flops; /* fp64 flops executed by the solve_kernel launches */
} bark_mll_timing;

int bark_mll_batched_hip(bark_ctx *ctx,
                         const void *packed, const bark_pack_info *info, /* forests (device / host info) */
                         const double *X, int64_t N, int64_t d,         /* training inputs (device) */
                         const double *y,                               /* (N,) targets (device) */
                         const double *noise, const double *scale,      /* (B,) device; scale may be NULL */
                         const double *shift,                           /* (B,) device or NULL */
                         int flags,                                     /* BARK_MLL_* */
                         const double *cand, int64_t C,                 /* (C, d) candidates or NULL/0 */
                         double *mll_out,                               /* (B,) device */
                         double *mu_out, double *var_out,               /* (B, C) device or NULL */
                         double *cov_out,                               /* (B, C, C) device or NULL */
                         int32_t *info_out,                             /* (B,) device */
                         void *workspace, size_t workspace_bytes, int64_t Bc,
                         bark_mll_timing *timing, /* optional (host); when non-NULL the call synchronises */
                         void *stream);

/* ---------------------------------------------------------------------------------------
 * Leaf-space MLL — the same quantity as bark_mll_batched_hip (mcmc_record_mll.py:57-74 / bark_sampler.py:153-162
 * conventions via `flags`), computed without ever forming the N x N matrix: K = (1/m) Z Z' with Z the one-hot
 * leaf matrix (N x R, R = info->max_bits), so
 *   log|K_s| = N log s2 + log|I_R + c Z'Z| ,   y'K_s^-1 y = (y'y - c v'(I_R + c Z'Z)^-1 v) / s2 ,
 *   s2 = 1e-6 + noise, c = scale / (m s2), v = Z'y .
 * O(N R^2 / 64 + R^3) instead of O(N^3): an exact algebraic alternative (agreement with the dense path to
 * ~1e-12 relative at noise 0.1; it loses digits as noise -> 0 through the subtraction y'y - c v'M^-1 v).
 * It is NOT the Gram + Cholesky work the benchmark metric counts and bench.py never times it as `value`.
 * Posterior (tree_gps.py:80-113, diagonal) in the same leaf space, with M = I_R + c Z'Z, w = M^-1 v and L(x) the m
 * leaves a candidate x reaches (Z'K_s^-1 Z = (m/scale)(I - M^-1)):
 *   mu(x) = c * sum_{a in L(x)} w_a ,    var(x) = (scale/m) * sum_{a,b in L(x)} (M^-1)_ab .
 * Pass C candidates (+ mu_out, var_out (B, C), scale, BARK_MLL_INCLUDE_SCALE); at most 64 trees.
 * ------------------------------------------------------------------------------------- */
size_t bark_mll_leafspace_workspace_bytes(int64_t N, int64_t max_bits, int64_t m, int64_t Bc, int64_t C);
int bark_mll_leafspace_hip(bark_ctx *ctx, const void *packed, const bark_pack_info *info, const double *X, int64_t N, int64_t d,
                           const double *y, const double *noise, const double *scale, int flags, const double *cand,
                           int64_t C, double *mll_out, double *mu_out, double *var_out, int32_t *info_out,
                           void *workspace, size_t workspace_bytes, int64_t Bc, void *stream);

/* Explicit inverse in the same leaf space — what the sampler rebuilds after every noise/scale proposal
 * (bark_sampler.py:267-272: inv + slogdet of scale K + (1e-6 + noise) I) and the acquisition builder reads
 * (opt_model.py:54-59) — without factorising the N x N matrix:
 *   K_s^-1 = (I - c Z M^-1 Z') / s2 ,   K_s^-1 y = (y - c Z w) / s2 ,   log|K_s| = -2 mll - y'K_s^-1 y
 * (mll in the convention selected by `flags`, as above).  kinv_out: (B, N, N); kinv_y_out: (B, N) or NULL.
 * Cost: the R x R sweep with an identity right-hand side + N R m + N^2 m gathered adds. */
size_t bark_kernel_inverse_leafspace_workspace_bytes(int64_t N, int64_t max_bits, int64_t m, int64_t Bc);
int bark_kernel_inverse_leafspace_hip(bark_ctx *ctx, const void *packed, const bark_pack_info *info, const double *X, int64_t N,
                                      int64_t d, const double *y, const double *noise, const double *scale, int flags,
                                      double *mll_out, double *kinv_out, double *kinv_y_out, int32_t *info_out,
                                      void *workspace, size_t workspace_bytes, int64_t Bc, void *stream);

/* ---------------------------------------------------------------------------------------
 * Woodbury / determinant-lemma updates — quick_inverse.py:13-33 (the per-tree step of the sampler,
 * bark_sampler.py:233-257).  With mul = -1 if `subtract` else +1:
 *   K_out         = K_inv - K_inv U (mul I + U' K_inv U)^-1 U' K_inv          (low_rank_inv_update)
 *   logabsdet_out = log|det(I + mul U' K_inv U)|                              (low_rank_det_update adds K_logdet)
 * K_inv, K_out: (N, N) device (K_out may alias K_inv or be NULL); U: (N, r) device, r <= 64;
 * logabsdet_out: device scalar or NULL.  `symmetric != 0` promises K_inv == K_inv' and saves one pass.
 * ------------------------------------------------------------------------------------- */
size_t bark_lowrank_workspace_bytes(int64_t N, int64_t r);
int bark_lowrank_update_hip(const double *K_inv, int64_t N, const double *U, int64_t r, int subtract, int symmetric,
                            double *K_out, double *logabsdet_out, void *workspace, size_t workspace_bytes,
                            void *stream);
/* *singular_out (HOST) = 1-based column of the first exactly-zero pivot of (mul I + U' K_inv U) met by the last
 * bark_lowrank_update_hip / swap evaluation that used `workspace` with this (N, r), or 0.  Synchronises `stream`.
 * The reference's np.linalg.solve / slogdet raise LinAlgError there (quick_inverse.py:19,31). */
int bark_lowrank_status_hip(void *workspace, int64_t N, int64_t r, int32_t *singular_out, void *stream);

/* Fused tree swap for the sampler's per-tree step (bark_sampler.py:233-257): the reference chains
 * subtract(U_old) -> add(U_new) -> mll on K_inv (about nine passes over the N x N matrix) before it can
 * accept or reject.  With U = [U_old U_new] (N x (r_old + r_new), r_old + r_new <= 64), C = diag(-I, +I):
 *   eval : ONE pass, Y = K_inv U; scalars_out[0] = v'(C+G)^-1 v  (v = Y'y, G = U'Y)  so that
 *          y'K'^-1 y = y'K^-1 y - scalars_out[0];   scalars_out[1] = log|det(C+G)| = log|K'| - log|K|
 *   apply: K_out = K_inv - Y (C+G)^-1 Y'   from the workspace left by eval (accepted proposals only).
 * K_inv symmetric; workspace >= bark_lowrank_workspace_bytes(N, r_old + r_new). */
int bark_lowrank_swap_eval_hip(const double *K_inv, int64_t N, const double *U, int64_t r_old, int64_t r_new,
                               const double *y, double *scalars_out, void *workspace, size_t workspace_bytes,
                               void *stream);
int bark_lowrank_swap_apply_hip(const double *K_inv, int64_t N, int64_t r, void *workspace, double *K_out,
                                void *stream);

/* The same proposal evaluated straight from the two trees (bark_sampler.py:233-257 incl. get_leaf_vectors,
 * forest.py:70-75): `packed`/`info` = wire format of ONE forest made of the pair [old tree, new tree]
 * (bark_forest_pack, B = 1, m = 2).  The pair's one-hot leaf code is [U_old U_new] / s with one column per leaf
 * of each tree (unreached leaves give zero columns, which change neither scalar).  r_old = leaves of the old tree
 * (max_bits of bark_forest_pack_info on that tree alone), s = sqrt(scale / m).  scalars_out as above.  The
 * workspace (>= bark_tree_swap_workspace_bytes(N, info->max_bits)) begins with the swap_eval layout:
 * bark_lowrank_swap_apply_hip(K_inv, N, info->max_bits, workspace, K_out, stream) commits the proposal. */
size_t bark_tree_swap_workspace_bytes(int64_t N, int64_t r);
int bark_tree_swap_eval_hip(bark_ctx *ctx, const double *K_inv, int64_t N, const void *packed, const bark_pack_info *info,
                            const double *X, int64_t d, int64_t r_old, double s, const double *y, double *scalars_out,
                            void *workspace, size_t workspace_bytes, void *stream);

/* The proposals of several independent chains (bark_sampler.py:147) in one call.  K_inv: (nc, N, N) device;
 * packed/info: nc forests of two trees [old, new] (B = nc, m = 2); r_old, s: HOST arrays (nc); scalars_out: device
 * (nc, 2) as in bark_lowrank_swap_eval_hip; workspace: bark_tree_swap_chains_workspace_bytes(N, info->max_bits, nc)
 * bytes (nc per-chain blocks of *chain_stride_bytes, then the leaf codes of all chains).  1 <= nc <= 64.
 * With N even and at most 16 leaves per pair every kernel takes the chain index from its grid (one launch sequence
 * for all chains); otherwise each chain's single-chain sequence runs on its own stream forked from `stream`.
 * ..._apply_chains: K_inv[b] is rewritten in place for every b with accept[b] != 0 (HOST int32 array). */
size_t bark_tree_swap_chains_workspace_bytes(int64_t N, int64_t r, int64_t nc, size_t *chain_stride_bytes);
int bark_tree_swap_eval_chains_hip(bark_ctx *ctx, const double *K_inv, int64_t N, int64_t nc, const void *packed,
                                   const bark_pack_info *info, const double *X, int64_t d, const int64_t *r_old,
                                   const double *s, const double *y, double *scalars_out, void *workspace,
                                   size_t workspace_bytes, void *stream);
int bark_lowrank_swap_apply_chains_hip(double *K_inv, int64_t N, int64_t nc, int64_t r, const int32_t *accept,
                                       void *workspace, size_t workspace_bytes, void *stream);

/* One sweep over the trees of nc chains with the Metropolis decision taken on the DEVICE — the per-tree loop of
 * bark_sampler.py:233-264 without a host round trip per tree.  Step t < n_steps swaps one tree per chain:
 *   packed + packed_offsets[t] (bytes, HOST array), infos[t] (HOST array): wire format of the step's nc pairs
 *                                          [old tree, new tree]  (bark_forest_pack with B = nc, m = 2)
 *   r_old[t * nc + b] (HOST)               leaves of chain b's old tree;  s[b] (HOST) = sqrt(scale_b / m)
 *   log_q_prior, log_u (DEVICE, (n_steps, nc))   proposal ratio and log of the uniform draw of bark_sampler.py:258
 * Per step: evaluate (as bark_tree_swap_eval_chains_hip), accept iff log_u <= min(log_q_prior + 0.5 (dquad - dlogdet), 0),
 * rewrite K_inv[b] of the accepted chains.  state (DEVICE, (nc, 2)): y'K^-1 y and log|K| per chain, updated in place.
 * accept_out (DEVICE int32, (n_steps, nc)): 1 accepted, 0 rejected, -1 singular r x r system (the reference raises
 * LinAlgError).  Nothing synchronises: the host reads accept_out and state once per sweep.
 * workspace >= bark_tree_swap_chains_workspace_bytes(N, r_max, nc, NULL) + 16 * nc bytes, r_max = max_t infos[t].max_bits. */
int bark_tree_sweep_chains_hip(bark_ctx *ctx, double *K_inv, int64_t N, int64_t nc, int64_t n_steps, const void *packed,
                               const int64_t *packed_offsets, const bark_pack_info *infos, const double *X, int64_t d,
                               const int64_t *r_old, const double *s, const double *y, const double *log_q_prior,
                               const double *log_u, double *state, int32_t *accept_out, void *workspace,
                               size_t workspace_bytes, void *stream);

/* quick_inverse.py:37-38  mll(K_inv, K_logdet, y) = 0.5 * (-y' K_inv y - K_logdet), on device. */
int bark_quadform_hip(const double *K_inv, const double *y, int64_t N, double *out, void *stream);
/* out[b] = alpha * sum_i A[b * lda + i] * y[i] + beta * c[b] for b < B (c may be NULL) — e.g. log|K_s| =
 * -(y' K_s^-1 y) - 2 mll from the rows K_s^-1 y of the inverse export (opt_model.py:101) and the MLL vector. */
int bark_rowdot_hip(const double *A, int64_t B, int64_t N, int64_t lda, const double *y, double alpha, const double *c,
                    double beta, double *out, void *stream);

/* ---------------------------------------------------------------------------------------
 * Mixture of Gaussians over forest samples — tree_gps.py:116-131 (mixture_of_gaussians_as_normal):
 *   E[Y] = mean_b mu_b ;  Var[Y] = mean_b (var_b + mu_b^2) - E[Y]^2
 * in two steps so that shards of samples on several GPUs combine with one all-reduce(sum) of `partial`:
 *   partial (2, C): [sum_b mu_b, sum_b (var_b + mu_b^2)] over the B local samples; finish divides by the global count.
 * ------------------------------------------------------------------------------------- */
int bark_mixture_partial_hip(const double *mu, const double *var, int64_t B, int64_t C, double *partial, void *stream);
int bark_mixture_finish_hip(const double *partial, double total, int64_t C, double *mean, double *var, void *stream);

/* Strided device-to-device copy of a rows x cols float64 block (hipMemcpy2DAsync): assembling [U_old U_new] from two
 * leaf-vector matrices (bark_sampler.py:233-236) without host arithmetic on device data. */
int bark_copy2d_hip(double *dst, int64_t ldd, const double *src, int64_t lds, int64_t rows, int64_t cols, void *stream);

/* ---------------------------------------------------------------------------------------
 * Multi-GPU exchanges over RCCL — replaces nothing in the reference, which evaluates the forest samples
 * of `batched_forest_gram_matrix` (forest.py:92-98) one after the other in one process.  One process per GPU holds a
 * contiguous block of samples; the ONLY data that crosses GPUs is the (B,) vector of log-likelihoods
 * (examples/mcmc/mcmc_record_mll.py:72-74) and, for the posterior, the 2 C partial sums of the mixture moments
 * (tree_gps.py:116-131).  librccl is loaded at run time; every function fails with BARK_ERR_HIP when it is absent.
 *   bark_comm_unique_id   128 bytes, generated by ONE rank and handed to the others by the caller (socket, file, MPI ...)
 *   bark_comm_create      collective over `world` ranks: communicator of this rank on `device`
 *   bark_allgather_mll    out[r * n_local + i] = local_r[i] for every rank r (equal block sizes; device pointers; enqueued
 *                         on `stream`)
 *   bark_allreduce_f64    in place over all ranks: sum (op_max == 0) or maximum (op_max != 0) */
int bark_comm_unique_id(void *id_out);
int bark_comm_create(const void *id, int rank, int world, int device, void **comm_out);
void bark_comm_destroy(void *comm);
int bark_allgather_mll(void *comm, const double *local, int64_t n_local, double *out, void *stream);
int bark_allreduce_f64(void *comm, double *buf, int64_t n, int op_max, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* BARK_HIP_H */
