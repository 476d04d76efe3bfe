"""Synthetic inputs for benchmarks and tests (host logic; no reference code is executed).

* `sample_prior_forests` restates the BART depth prior the reference samples forests from
  (src/bark/fitting/bark_prior_sampler.py:15-65 with tree_proposals.py:46-58,81-101,147-165 and
  tree_traversal.py:52-86): a node at depth d splits with probability alpha * (1 + d)^-beta, the
  rule is drawn uniformly inside the node's sub-domain, children take the first two inactive slots.
  It is a generator of *valid forests with the reference's statistics*, not a reproduction of the
  reference's RNG stream.
* `tree_function` restates the config-1 data path of
  src/bofire_mixed/benchmarks/tree_function.py:19-57 (TreeFunction benchmark).
* `full_binary_forest` is the stress variant of SURVEY §8d (depth-5 complete trees).
"""

from __future__ import annotations

import numpy as np

from .forest import NODE_RECORD_DTYPE, FeatureTypeEnum, create_empty_forest

CAT, INT, CONT = FeatureTypeEnum.Cat.value, FeatureTypeEnum.Int.value, FeatureTypeEnum.Cont.value


def _first_two_inactive(tree):
    idx = np.flatnonzero(tree["active"] == 0)
    if idx.size < 2:
        raise OverflowError("The tree container is not large enough")  # tree_proposals.py:58
    return int(idx[0]), int(idx[1])


def _grow(tree, node_idx, feature_idx, threshold):
    """tree_proposals.py:147-165."""
    left, right = _first_two_inactive(tree)
    depth = int(tree[node_idx]["depth"])
    for child in (left, right):
        tree[child] = (1, 0, 0, 0, 0, node_idx, depth + 1, 1)
    parent = tree[node_idx]["parent"]
    tree[node_idx] = (0, feature_idx, threshold, left, right, parent, depth, 1)
    return left, right


def _subspace(tree, node_idx, bounds, feat_types):
    """tree_traversal.py:52-86: the part of the domain that reaches `node_idx`."""
    sub = bounds.astype(np.float64).copy()
    while node_idx != 0:
        parent_idx = int(tree[node_idx]["parent"])
        parent = tree[parent_idx]
        f = int(parent["feature_idx"])
        thr = float(parent["threshold"])
        is_left = node_idx == int(parent["left"])
        if feat_types[f] == CAT:
            avail = int(sub[f, 1])
            if is_left:
                sub[f, 1] = int(thr) & avail
            else:
                full = 1
                while avail >= full:
                    full <<= 1
                sub[f, 1] = int(full - 1 - thr) & avail
        elif is_left:
            sub[f, 1] = min(thr, sub[f, 1])
        else:
            sub[f, 0] = max(thr + (1 if feat_types[f] == INT else 0), sub[f, 0])
        node_idx = parent_idx
    return sub


def _sample_rule(sub, feat_types, rng):
    """tree_proposals.py:81-101 (+ the validity checks of bark_prior_sampler.py:44-58)."""
    f = int(rng.integers(0, sub.shape[0]))
    if feat_types[f] == CAT:
        avail = int(sub[f, 1])
        bits = [i for i in range(avail.bit_length()) if avail >> i & 1]
        if len(bits) < 2:
            return None
        pick = int(rng.integers(1, (1 << len(bits)) - 1))  # proper non-empty subset
        thr = 0
        for k, i in enumerate(bits):
            thr |= (pick >> k & 1) << i
        return f, float(thr)
    if feat_types[f] == INT:
        lo, hi = int(sub[f, 0]), int(sub[f, 1])
        if lo >= hi:
            return None
        return f, float(rng.integers(lo, hi))
    return f, float(rng.uniform(sub[f, 0], sub[f, 1]))


def sample_prior_forest(m, bounds, feat_types, rng, alpha=0.95, beta=2.0, node_limit=100):
    bounds = np.asarray(bounds, dtype=np.float64)
    feat_types = np.asarray(feat_types)
    forest = create_empty_forest(m, node_limit)
    for t in range(m):
        tree = forest[t]
        stack = [0]
        while stack:
            node_idx = stack.pop()
            depth = int(tree[node_idx]["depth"])
            if rng.uniform() > alpha * (1 + depth) ** (-beta):
                continue
            rule = _sample_rule(_subspace(tree, node_idx, bounds, feat_types), feat_types, rng)
            if rule is None:
                continue
            left, right = _grow(tree, node_idx, *rule)
            stack.extend((left, right))
    return forest


def sample_prior_forests(B, m, bounds, feat_types, seed, alpha=0.95, beta=2.0, node_limit=100):
    """(B, m, node_limit) forests; forest b uses default_rng(seed + b)."""
    out = np.zeros((B, m, node_limit), dtype=NODE_RECORD_DTYPE)
    for b in range(B):
        out[b] = sample_prior_forest(m, bounds, feat_types, np.random.default_rng(seed + b), alpha, beta, node_limit)
    return out


def full_binary_forest(m, d, depth, rng, node_limit=100):
    """Complete binary trees with random continuous splits in (0,1): 2^(depth+1)-1 nodes each."""
    n_nodes = 2 ** (depth + 1) - 1
    assert n_nodes <= node_limit
    forest = np.zeros((m, node_limit), dtype=NODE_RECORD_DTYPE)
    for t in range(m):
        for i in range(n_nodes):
            dep = int(np.log2(i + 1))
            parent = 0xFFFFFFFF if i == 0 else (i - 1) // 2
            if dep == depth:
                forest[t, i] = (1, 0, 0, 0, 0, parent, dep, 1)
            else:
                forest[t, i] = (0, rng.integers(0, d), rng.uniform(0.05, 0.95), 2 * i + 1, 2 * i + 2, parent, dep, 1)
    return forest


def full_binary_forests(B, m, d, depth, rng, node_limit=100):
    """B forests of complete binary trees (the vectorised form of `full_binary_forest`, for benchmark-sized batches);
    SURVEY §8d's stress variant of c2/c3 uses depth 5: 63 nodes, 32 leaves per tree."""
    n_nodes = 2 ** (depth + 1) - 1
    assert n_nodes <= node_limit
    forest = np.zeros((B, m, node_limit), dtype=NODE_RECORD_DTYPE)
    i = np.arange(n_nodes)
    dep = np.floor(np.log2(i + 1)).astype(np.uint32)
    leaf = dep == depth
    f = forest[:, :, :n_nodes]
    f["is_leaf"] = leaf
    f["feature_idx"] = np.where(leaf, 0, rng.integers(0, d, size=(B, m, n_nodes)))
    f["threshold"] = np.where(leaf, 0.0, rng.uniform(0.05, 0.95, size=(B, m, n_nodes)))
    f["left"] = np.where(leaf, 0, 2 * i + 1)
    f["right"] = np.where(leaf, 0, 2 * i + 2)
    f["parent"] = np.where(i == 0, 0xFFFFFFFF, (i - 1) // 2).astype(np.uint32)
    f["depth"] = dep
    f["active"] = 1
    return forest


def unit_cube_problem(N, d, seed):
    """SURVEY §8d c2/c3 inputs: X ~ U[0,1)^(N x d), y = standardised N(0,1), all-continuous domain."""
    rng = np.random.default_rng(seed)
    X = rng.uniform(size=(N, d))
    y = rng.standard_normal((N, 1))
    y = (y - y.mean()) / y.std()
    bounds = np.tile(np.array([[0.0, 1.0]]), (d, 1))
    feat_types = np.full(d, CONT, dtype=np.int64)
    return X, y, bounds, feat_types


def mixed_problem(N, seed, d_cont=8, n_int=2, n_cat=2, cats=5):
    """SURVEY §8d c5 inputs: continuous U[0,1) + integer in [0,10] + categorical (ordinal codes)."""
    rng = np.random.default_rng(seed)
    cols, bounds, ft = [], [], []
    for _ in range(d_cont):
        cols.append(rng.uniform(size=N)); bounds.append((0.0, 1.0)); ft.append(CONT)
    for _ in range(n_int):
        cols.append(rng.integers(0, 11, size=N).astype(np.float64)); bounds.append((0.0, 10.0)); ft.append(INT)
    for _ in range(n_cat):
        cols.append(rng.integers(0, cats, size=N).astype(np.float64))
        bounds.append((0.0, float((1 << cats) - 1))); ft.append(CAT)
    X = np.stack(cols, axis=1)
    y = rng.standard_normal((N, 1))
    spread = y.std()
    y = (y - y.mean()) / (spread if spread > 0 else 1.0)  # N = 1: a single centred target
    return X, y, np.array(bounds), np.array(ft, dtype=np.int64)


def tree_function(dim=5, m=50, function_seed=1, node_limit=100):
    """tree_function.py:36-57,19-33 (TreeFunction defaults, continuous inputs only).

    Returns (forest, leaf_values, f) with f(X) = sum_t leaf_values[t, leaf_t(x)] evaluated through
    `bark_amd.forest.pass_through_forest` (GPU)."""
    from .forest import pass_through_forest

    rng = np.random.default_rng(function_seed)
    forest = create_empty_forest(m, node_limit)
    for tree in forest:
        new_nodes = [0]
        while new_nodes:
            node_idx = new_nodes.pop()
            depth = int(tree[node_idx]["depth"])
            if rng.uniform() > 0.95 * (1 + depth) ** (-2.0):
                continue
            feature_idx = rng.integers(dim)
            threshold = rng.uniform(0, 1)
            left, right = _grow(tree, node_idx, feature_idx, threshold)
            new_nodes.extend([left, right])
    leaf_values = rng.standard_normal(forest.shape)
    feat_types = np.full(dim, CONT, dtype=np.int64)

    def f(X):
        leaves = pass_through_forest(forest, X, feat_types)
        return leaf_values[np.arange(m), leaves].sum(axis=1)

    return forest, leaf_values, f
