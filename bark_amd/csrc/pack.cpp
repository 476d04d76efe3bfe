// Host side of libbarkhip.so: error plumbing and the forest wire-format packer.
//
// The reference keeps a forest as a numpy array of packed 26-byte records
// (src/bark/forest.py:8-19) in which children sit at arbitrary slots of an L-slot container
// (first two inactive slots, src/bark/fitting/tree_proposals.py:46-58,147-165) and pruned
// sub-trees stay behind as garbage.  The device wants the few live nodes of each tree,
// aligned, with validated child links (a GPU walk must never leave the container or spin).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

namespace bark {

char *error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

namespace {

struct Node {
    uint8_t is_leaf;
    uint32_t feature_idx;
    float threshold;
    uint32_t left, right;
};

// offsets 0,1,5,9,13 of the packed record; unaligned => memcpy
inline Node load_node(const uint8_t *rec) {
    Node n;
    n.is_leaf = rec[0];
    std::memcpy(&n.feature_idx, rec + 1, 4);
    std::memcpy(&n.threshold, rec + 5, 4);
    std::memcpy(&n.left, rec + 9, 4);
    std::memcpy(&n.right, rec + 13, 4);
    return n;
}

struct Scratch {
    std::vector<int32_t> cidx;   // original slot -> compact index (-1 = unseen)
    std::vector<uint8_t> colour; // 0 white, 1 on the DFS stack, 2 finished
    std::vector<uint32_t> stack_node;
    std::vector<uint8_t> stack_phase;
    std::vector<int32_t> height;  // edges to the farthest leaf below a finished node
};

struct TreeStats {
    int64_t nodes = 0, leaves = 0, depth = 0;
};

// Depth-first walk from the root.  `out` (may be null) receives 4 uint32 per compact node.
// `cap` (with `out`): number of 16-byte slots `out` holds; a tree with more live nodes is an error, never an overrun.
int pack_tree(const uint8_t *tree, int64_t L, const int64_t *feat_types, int64_t d, Scratch &s, uint32_t *out,
              TreeStats *stats, int64_t b, int64_t t, uint32_t bit_base = 0, int64_t cap = 0) {
    s.cidx.assign((size_t)L, -1);
    s.colour.assign((size_t)L, 0);
    s.stack_node.clear();
    s.stack_phase.clear();
    s.height.assign((size_t)L, 0);
    int64_t n_nodes = 0, n_leaves = 0;

    auto discover = [&](uint32_t orig) -> int {
        if (out && n_nodes >= cap) return 1;
        s.cidx[orig] = (int32_t)n_nodes++;
        s.colour[orig] = 1;
        s.stack_node.push_back(orig);
        s.stack_phase.push_back(0);
        return 0;
    };
    if (discover(0)) return fail(BARK_ERR_ARG, "bark_forest_pack: info does not match forest");

    while (!s.stack_node.empty()) {
        uint32_t orig = s.stack_node.back();
        uint8_t phase = s.stack_phase.back();
        Node n = load_node(tree + (size_t)orig * NODE_BYTES);
        uint32_t *slot = out ? out + (size_t)s.cidx[orig] * 4 : nullptr;
        if (n.is_leaf) {  // forest.py:34-35: any non-zero is_leaf ends the walk
            if (slot) {
                slot[0] = LEAF_FLAG | (uint32_t)n_leaves;
                slot[1] = orig;
                slot[2] = bit_base + (uint32_t)n_leaves;  // this leaf's bit in the forest's one-hot code
                slot[3] = (uint32_t)s.cidx[orig];
            }
            ++n_leaves;
            s.colour[orig] = 2;
            s.stack_node.pop_back();
            s.stack_phase.pop_back();
            continue;
        }
        if (phase == 0) {
            if ((int64_t)n.feature_idx >= d)
                return fail(BARK_ERR_TREE, "forest %lld tree %lld node %u: feature_idx %u >= d=%lld", (long long)b,
                            (long long)t, orig, n.feature_idx, (long long)d);
            if ((int64_t)n.left >= L || (int64_t)n.right >= L)
                return fail(BARK_ERR_TREE, "forest %lld tree %lld node %u: child index (%u,%u) outside container L=%lld",
                            (long long)b, (long long)t, orig, n.left, n.right, (long long)L);
            if (slot) {
                if (feat_types[n.feature_idx] == 0) {  // FeatureTypeEnum.Cat, forest.py:37-39
                    float thr = n.threshold;
                    if (!(thr >= 0.0f) || !(thr < 4294967296.0f))
                        return fail(BARK_ERR_CATEGORICAL,
                                    "forest %lld tree %lld node %u: categorical threshold %g is not a bitmask in [0, 2^32)",
                                    (long long)b, (long long)t, orig, (double)thr);
                    slot[0] = n.feature_idx | CAT_FLAG;
                    slot[1] = (uint32_t)(int64_t)thr;  // int(threshold): truncation toward zero
                } else {
                    slot[0] = n.feature_idx;
                    std::memcpy(&slot[1], &n.threshold, 4);
                }
            } else if (feat_types[n.feature_idx] == 0) {
                float thr = n.threshold;
                if (!(thr >= 0.0f) || !(thr < 4294967296.0f))
                    return fail(BARK_ERR_CATEGORICAL,
                                "forest %lld tree %lld node %u: categorical threshold %g is not a bitmask in [0, 2^32)",
                                (long long)b, (long long)t, orig, (double)thr);
            }
            if (n.feature_idx > FEAT_MASK)
                return fail(BARK_ERR_ARG, "feature_idx %u too large", n.feature_idx);
        }
        if (phase < 2) {
            uint32_t child = phase == 0 ? n.left : n.right;
            s.stack_phase.back() = phase + 1;
            if (s.colour[child] == 1)
                return fail(BARK_ERR_TREE, "forest %lld tree %lld: cycle through node %u", (long long)b, (long long)t,
                            child);
            if (s.colour[child] == 0 && discover(child))
                return fail(BARK_ERR_ARG, "bark_forest_pack: info does not match forest");
            if (slot) slot[2 + phase] = (uint32_t)s.cidx[child];
            continue;
        }
        // The walk bound is the LONGEST root-to-leaf path.  The reference just follows child pointers, so a node
        // that two parents share (not a tree, but walkable) may sit deeper on one path than where the DFS first met
        // it: heights are taken bottom-up from the finished children, not from the discovery depth.
        const int32_t hl = s.height[n.left], hr = s.height[n.right];
        s.height[orig] = 1 + (hl > hr ? hl : hr);
        s.colour[orig] = 2;
        s.stack_node.pop_back();
        s.stack_phase.pop_back();
    }
    stats->nodes = n_nodes;
    stats->leaves = n_leaves;
    stats->depth = s.height[0];
    return BARK_OK;
}

// Forests are independent: run `fn(b_begin, b_end)` over contiguous ranges on a few host threads (the sampler
// hands over hundreds of forests; one thread packs ~1 us per tree).  The error of the lowest failing range wins
// and is copied into the CALLER's thread-local message buffer.
template <class Fn>
int for_forest_ranges(int64_t B, Fn fn) {
    const unsigned hw = std::thread::hardware_concurrency();
    const int64_t want = std::min<int64_t>({(int64_t)(hw ? hw : 1), 8, B / 16});
    if (want <= 1) return fn(0, B);
    const int64_t T = want, per = (B + T - 1) / T;
    std::vector<int> rc((size_t)T, BARK_OK);
    std::vector<std::string> msg((size_t)T);
    std::vector<std::thread> pool;
    for (int64_t k = 0; k < T; ++k)
        pool.emplace_back([&, k] {
            const int64_t b0 = k * per, b1 = std::min(B, b0 + per);
            if (b0 >= b1) return;
            rc[(size_t)k] = fn(b0, b1);
            if (rc[(size_t)k]) msg[(size_t)k] = error_buffer();  // this worker's thread-local buffer
        });
    for (auto &t : pool) t.join();
    for (int64_t k = 0; k < T; ++k)
        if (rc[(size_t)k]) return fail(rc[(size_t)k], "%s", msg[(size_t)k].c_str());
    return BARK_OK;
}

}  // namespace
}  // namespace bark

using namespace bark;

extern "C" {

int bark_version(void) { return BARK_HIP_VERSION; }

const char *bark_last_error(void) { return error_buffer(); }

int bark_forest_pack_info(const void *nodes26, int64_t B, int64_t m, int64_t L, const int64_t *feat_types, int64_t d,
                          bark_pack_info *info) {
    error_buffer()[0] = 0;
    if (!nodes26 || !feat_types || !info || B < 1 || m < 1 || L < 1 || d < 1)
        return fail(BARK_ERR_ARG, "bark_forest_pack_info: bad argument (B=%lld m=%lld L=%lld d=%lld)", (long long)B,
                    (long long)m, (long long)L, (long long)d);
    const uint8_t *base = static_cast<const uint8_t *>(nodes26);
    struct Extent {
        int64_t stride = 1, max_leaves = 1, max_depth = 0, max_bits = 1;
    };
    std::vector<Extent> per_forest((size_t)B);
    int rc_all = for_forest_ranges(B, [&](int64_t b0, int64_t b1) -> int {
        Scratch s;
        for (int64_t b = b0; b < b1; ++b) {
            Extent &e = per_forest[(size_t)b];
            int64_t bits = 0;
            for (int64_t t = 0; t < m; ++t) {
                TreeStats st;
                int rc = pack_tree(base + ((size_t)b * m + t) * L * NODE_BYTES, L, feat_types, d, s, nullptr, &st, b, t);
                if (rc) return rc;
                e.stride = std::max(e.stride, st.nodes);
                e.max_leaves = std::max(e.max_leaves, st.leaves);
                e.max_depth = std::max(e.max_depth, st.depth);
                bits += st.leaves;
            }
            e.max_bits = std::max(e.max_bits, bits);
        }
        return BARK_OK;
    });
    if (rc_all) return rc_all;
    int64_t stride = 1, max_leaves = 1, max_depth = 0, max_bits = 1;
    for (const Extent &e : per_forest) {
        stride = std::max(stride, e.stride);
        max_leaves = std::max(max_leaves, e.max_leaves);
        max_depth = std::max(max_depth, e.max_depth);
        max_bits = std::max(max_bits, e.max_bits);
    }
    info->B = B;
    info->m = m;
    info->L = L;
    info->stride = stride;
    info->max_leaves = max_leaves;
    info->max_depth = max_depth;
    info->packed_bytes = B * m * stride * 16;
    info->max_bits = max_bits;
    return BARK_OK;
}

int bark_forest_pack(const void *nodes26, const int64_t *feat_types, int64_t d, const bark_pack_info *info,
                     void *packed) {
    error_buffer()[0] = 0;
    if (!nodes26 || !feat_types || !info || !packed) return fail(BARK_ERR_ARG, "bark_forest_pack: null argument");
    const uint8_t *base = static_cast<const uint8_t *>(nodes26);
    uint32_t *out = static_cast<uint32_t *>(packed);
    const int64_t B = info->B, m = info->m, L = info->L, stride = info->stride;
    if (B < 1 || m < 1 || L < 1 || stride < 1) return fail(BARK_ERR_ARG, "bark_forest_pack: bad info");
    return for_forest_ranges(B, [&](int64_t b0, int64_t b1) -> int {
        Scratch s;
        for (int64_t b = b0; b < b1; ++b) {
            uint32_t bit_base = 0;  // trees of a forest own consecutive bit fields
            for (int64_t t = 0; t < m; ++t) {
                uint32_t *dst = out + ((size_t)b * m + t) * stride * 4;
                // unused tail slots: self-looping leaves (never reached, but harmless if they were)
                for (int64_t k = 0; k < stride; ++k) {
                    dst[k * 4 + 0] = LEAF_FLAG;
                    dst[k * 4 + 1] = 0;
                    dst[k * 4 + 2] = dst[k * 4 + 3] = (uint32_t)k;
                }
                TreeStats st;
                int rc = pack_tree(base + ((size_t)b * m + t) * L * NODE_BYTES, L, feat_types, d, s, dst, &st, b, t, bit_base,
                                   stride);
                if (rc) return rc;
                bit_base += (uint32_t)st.leaves;
            }
            if ((int64_t)bit_base > info->max_bits) return fail(BARK_ERR_ARG, "bark_forest_pack: info does not match forest");
        }
        return BARK_OK;
    });
}

}  // extern "C"
