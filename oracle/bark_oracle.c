/*
 * bark_oracle.c — CPU restatement of the reference's forest-kernel hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load this library; the product path
 * (bark_amd/ -> libbarkhip.so) never links, imports or calls it.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit against golden
 * vectors produced by running the reference's own Python (tests/golden/make_golden.py,
 * fixtures tests/golden/g*.npz) in tests/test_oracle_golden.py.
 *
 * The reference is pure Python + numba (no native code), so this is a restatement of
 * its semantics in plain C, function by function, citing /root/reference/src/bark/forest.py.
 * It is deliberately single-threaded and unoptimised: the reference's loops are
 * `@njit(parallel=False)` (forest.py:50,58,92).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NODE_BYTES 26 /* packed NODE_RECORD_DTYPE, forest.py:8-19 */

enum { FEAT_CAT = 0, FEAT_INT = 1, FEAT_CONT = 2 }; /* FeatureTypeEnum, forest.py:22-25 */

/* error codes (mirrored by oracle.py) */
enum {
    ORC_OK = 0,
    ORC_BAD_CHILD = 1,   /* child index outside [0, L) or walk longer than L steps (cycle) */
    ORC_BAD_FEATURE = 2, /* feature_idx >= d */
    ORC_BAD_CAT = 3      /* categorical value negative / NaN / inf: Python raises there */
};

typedef struct {
    uint8_t is_leaf;
    uint32_t feature_idx;
    float threshold;
    uint32_t left, right;
} node_t;

/* field offsets 0,1,5,9,13,17,21,25 (unaligned) — read with memcpy, never by cast */
static node_t load_node(const uint8_t *rec) {
    node_t n;
    n.is_leaf = rec[0];
    memcpy(&n.feature_idx, rec + 1, 4);
    memcpy(&n.threshold, rec + 5, 4);
    memcpy(&n.left, rec + 9, 4);
    memcpy(&n.right, rec + 13, 4);
    return n;
}

/* forest.py:28-47 `_pass_one_through_tree`: returns the NODE INDEX of the reached leaf. */
static int pass_one(const uint8_t *tree, int64_t L, const double *x, int64_t d,
                    const int64_t *feat_types, uint32_t *out) {
    uint32_t idx = 0;
    for (int64_t step = 0; step <= L; ++step) {
        node_t n = load_node(tree + (size_t)idx * NODE_BYTES);
        if (n.is_leaf) { /* forest.py:34-35 */
            *out = idx;
            return ORC_OK;
        }
        if ((int64_t)n.feature_idx >= d) return ORC_BAD_FEATURE;
        int cond;
        if (feat_types[n.feature_idx] == FEAT_CAT) {
            /* forest.py:37-39: bit = 1 << int(X[f]); cond = bit & int(threshold).
             * int() truncates toward zero; Python raises for NaN/inf/negative shift. */
            double xv = x[n.feature_idx];
            if (!(xv == xv) || isinf(xv) || xv <= -1.0) return ORC_BAD_CAT;
            double xt = trunc(xv);
            int64_t mask = (int64_t)n.threshold; /* bitmask stored as float32, < 2^24 by construction */
            cond = (xt < 63.0) ? (int)((mask >> (int64_t)xt) & 1) : 0;
        } else {
            /* forest.py:41: float64 x vs float32 threshold widened to float64; NaN -> False */
            cond = x[n.feature_idx] <= (double)n.threshold;
        }
        idx = cond ? n.left : n.right; /* forest.py:43-46 */
        if ((int64_t)idx >= L) return ORC_BAD_CHILD;
    }
    return ORC_BAD_CHILD;
}

/* forest.py:58-67 `pass_through_forest`: out is (N, m) uint32, C order. */
int orc_pass_through_forest(const uint8_t *nodes, int64_t m, int64_t L, const double *X, int64_t N,
                            int64_t d, const int64_t *feat_types, uint32_t *out) {
    for (int64_t t = 0; t < m; ++t) {
        const uint8_t *tree = nodes + (size_t)t * L * NODE_BYTES;
        for (int64_t i = 0; i < N; ++i) { /* forest.py:50-55 `pass_through_tree` */
            int rc = pass_one(tree, L, X + (size_t)i * d, d, feat_types, &out[(size_t)i * m + t]);
            if (rc) return rc;
        }
    }
    return ORC_OK;
}

/* forest.py:78-89 `forest_gram_matrix`: K = (1/m) * sum_t [leaf_t(x1_i) == leaf_t(x2_j)].
 * The reference evaluates `1 / m * np.sum(bool, axis=-1)` => fl(fl(1/m) * count). */
int orc_forest_gram_matrix(const uint8_t *nodes, int64_t m, int64_t L, const double *x1, int64_t N,
                           const double *x2, int64_t M, int64_t d, const int64_t *feat_types,
                           double *out) {
    uint32_t *l1 = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(N * m + 1));
    uint32_t *l2 = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(M * m + 1));
    int rc = orc_pass_through_forest(nodes, m, L, x1, N, d, feat_types, l1);
    if (!rc) rc = orc_pass_through_forest(nodes, m, L, x2, M, d, feat_types, l2);
    if (!rc) {
        const double inv_m = 1.0 / (double)m;
        for (int64_t i = 0; i < N; ++i) {
            const uint32_t *a = l1 + (size_t)i * m;
            for (int64_t j = 0; j < M; ++j) {
                const uint32_t *b = l2 + (size_t)j * m;
                int64_t count = 0;
                for (int64_t t = 0; t < m; ++t) count += (a[t] == b[t]); /* forest.py:87 */
                out[(size_t)i * M + j] = inv_m * (double)count;          /* forest.py:88 */
            }
        }
    }
    free(l1);
    free(l2);
    return rc;
}

/* forest.py:92-98 `batched_forest_gram_matrix`: serial loop over the B forests. */
int orc_batched_forest_gram_matrix(const uint8_t *nodes, int64_t B, int64_t m, int64_t L,
                                   const double *x1, int64_t N, const double *x2, int64_t M,
                                   int64_t d, const int64_t *feat_types, double *out) {
    for (int64_t b = 0; b < B; ++b) {
        int rc = orc_forest_gram_matrix(nodes + (size_t)b * m * L * NODE_BYTES, m, L, x1, N, x2, M, d,
                                        feat_types, out + (size_t)b * N * M);
        if (rc) return rc;
    }
    return ORC_OK;
}

/* forest.py:102-111 `batched_forest_gram_matrix_no_null`:
 * (K - n_null/m) * (m / max(m - n_null, 1)),  n_null = sum_t nodes[b,t,0].is_leaf */
int orc_batched_forest_gram_matrix_no_null(const uint8_t *nodes, int64_t B, int64_t m, int64_t L,
                                           const double *x1, int64_t N, const double *x2, int64_t M,
                                           int64_t d, const int64_t *feat_types, double *out) {
    int rc = orc_batched_forest_gram_matrix(nodes, B, m, L, x1, N, x2, M, d, feat_types, out);
    if (rc) return rc;
    for (int64_t b = 0; b < B; ++b) {
        int64_t n_null = 0;
        for (int64_t t = 0; t < m; ++t) n_null += nodes[((size_t)b * m + t) * L * NODE_BYTES];
        int64_t non_null = m - n_null;
        double scale = (double)m / (double)(non_null > 1 ? non_null : 1); /* forest.py:110 */
        double shift = (double)n_null / (double)m;                        /* forest.py:111 */
        double *K = out + (size_t)b * N * M;
        for (int64_t e = 0; e < N * M; ++e) K[e] = (K[e] - shift) * scale;
    }
    return ORC_OK;
}

static int cmp_u32(const void *a, const void *b) {
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return (x > y) - (x < y);
}

/* forest.py:70-75 `get_leaf_vectors`: one-hot (N, r), columns = np.unique (ascending) of the
 * reached leaf node indices.  `out` must hold N*L doubles; returns r via *r_out. */
int orc_get_leaf_vectors(const uint8_t *tree, int64_t L, const double *X, int64_t N, int64_t d,
                         const int64_t *feat_types, double *out, int64_t *r_out) {
    uint32_t *leaves = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(2 * N + 1));
    uint32_t *uniq = leaves + N;
    int rc = orc_pass_through_forest(tree, 1, L, X, N, d, feat_types, leaves);
    if (!rc) {
        memcpy(uniq, leaves, sizeof(uint32_t) * (size_t)N);
        qsort(uniq, (size_t)N, sizeof(uint32_t), cmp_u32);
        int64_t r = 0;
        for (int64_t i = 0; i < N; ++i)
            if (i == 0 || uniq[i] != uniq[r - 1]) uniq[r++] = uniq[i];
        for (int64_t i = 0; i < N; ++i)
            for (int64_t c = 0; c < r; ++c) out[(size_t)i * r + c] = (leaves[i] == uniq[c]) ? 1.0 : 0.0;
        *r_out = r;
    }
    free(leaves);
    return rc;
}
