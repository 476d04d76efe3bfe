"""Parity of the HIP path (through the Python API -> ctypes C ABI -> libbarkhip.so) against
  (1) golden vectors produced by the reference itself (tests/golden/*.npz), and
  (2) the CPU oracle (oracle/) on seeded inputs, and
  (3) size-independent properties at BASELINE.json's full sizes.

Bars: leaf indices and Gram matrices BIT-EXACT; MLL / posterior within the stated fp64 tolerance
rtol=1e-9, atol=1e-8 (reference: LU inv+slogdet; here: blocked Cholesky)."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

MLL_RTOL, MLL_ATOL = 1e-9, 1e-8


@pytest.fixture(scope="module")
def B():
    import torch

    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import bark_amd.fitting as fit
    import bark_amd.forest as bf
    import bark_amd.tree_kernels as tk
    from bark_amd import synthetic
    from oracle import oracle as orc

    class NS:
        pass

    ns = NS()
    ns.bf, ns.fit, ns.tk, ns.syn, ns.orc, ns.torch = bf, fit, tk, synthetic, orc, torch
    return ns


def raw(B, a):
    return B.orc.nodes_from_raw(a)


# ------------------------------------------------------------------ golden vectors (reference) ----
def test_g1_kat_tree(B):
    g = load_golden("g1_kat_tree")
    nodes = raw(B, g["nodes"])
    leaves = B.bf.pass_through_forest(nodes, g["x"], g["feat_types"])
    assert leaves.dtype == np.uint32 and leaves.shape == (20, 1) and np.array_equal(leaves, g["leaves"])
    assert np.array_equal(B.bf.pass_through_tree(nodes[0], g["x"], g["feat_types"]), g["leaves"][:, 0])
    for i in (0, 7, 19):  # the single-point walk of forest.py:28-47
        assert B.bf._pass_one_through_tree(nodes[0], g["x"][i], g["feat_types"]) == int(g["leaves"][i, 0])
    K = B.bf.forest_gram_matrix(nodes, g["x"], g["x"], g["feat_types"])
    assert K.dtype == np.float64 and np.array_equal(K, g["K"])
    assert np.array_equal(B.bf.get_leaf_vectors(nodes[0], g["x"], g["feat_types"]), g["leaf_vectors"])


def test_g2_two_tree_kat(B):
    g = load_golden("g2_two_tree_kat")
    forest, x, ft = raw(B, g["forest"]), g["x"], g["feat_types"]
    assert np.array_equal(B.bf.forest_gram_matrix(forest, x, x, ft), g["K"])
    assert np.array_equal(B.bf.get_leaf_vectors(forest[0], x, ft) * np.sqrt(0.25), g["cur_leaf_vectors"])
    # quick_inverse.mll on the reference's K_inv / logdet (test_quick_inverse.py:55-101 setting)
    y = np.linspace(-1, 1, 20).reshape(-1, 1)
    want = B.orc.mll(g["K_inv"], float(g["K_logdet"]), y)
    got = B.fit.quick_inverse.mll(g["K_inv"], float(g["K_logdet"]), y)
    assert np.isclose(got, want, rtol=1e-12)
    # the same through the Cholesky sweep: K_s = 0.5 K + (1e-6 + noise) I with noise = 0.1 - 1e-6
    got2 = B.fit.batched_mll(forest[None], [0.1 - 1e-6], [0.5], x, y, ft, include_scale=True, include_2pi=False)
    assert np.allclose(got2, want, rtol=MLL_RTOL, atol=MLL_ATOL)
    swapped = forest.copy()
    swapped[0] = raw(B, g["new_nodes"])
    assert np.array_equal(B.bf.forest_gram_matrix(swapped, x, x, ft), g["K_swapped"])


@pytest.mark.parametrize("N", [64, 257])
def test_g3_prior_mixed(B, N):
    g = load_golden(f"g3_prior_mixed_n{N}")
    forest, X, y, ft = raw(B, g["forest"]), g["X"], g["y"], g["feat_types"]
    for b in range(forest.shape[0]):
        assert np.array_equal(B.bf.pass_through_forest(forest[b], X, ft), g["leaves"][b])
    assert np.array_equal(B.bf.batched_forest_gram_matrix(forest, X, X, ft), g["K"])
    assert np.array_equal(B.bf.batched_forest_gram_matrix_no_null(forest, X, X, ft), g["K_no_null"])
    ex = B.fit.batched_mll(forest, g["noise"], None, X, y, ft, include_scale=False, include_2pi=True)
    sa = B.fit.batched_mll(forest, g["noise"], g["scale"], X, y, ft, include_scale=True, include_2pi=False)
    assert np.allclose(ex, g["mll_example"], rtol=MLL_RTOL, atol=MLL_ATOL)
    assert np.allclose(sa, g["mll_sampler"], rtol=MLL_RTOL, atol=MLL_ATOL)
    # the example script's own signature (mcmc_record_mll.py:57): mll(model, data, domain)
    ex2 = B.fit.mll((forest, g["noise"], g["scale"]), (X, y), ft)
    assert np.array_equal(ex2, ex)


def test_g4_all_null_forest(B):
    g = load_golden("g4_all_null")
    forest, X, ft = raw(B, g["forest"]), g["X"], g["feat_types"]
    assert np.array_equal(B.bf.pass_through_forest(forest[0], X, ft), g["leaves"])
    K = B.bf.batched_forest_gram_matrix(forest, X, X, ft)
    assert np.array_equal(K, g["K"]) and np.all(K == 1.0)
    assert np.array_equal(B.bf.batched_forest_gram_matrix_no_null(forest, X, X, ft), g["K_no_null"])
    ex = B.fit.batched_mll(forest, g["noise"], None, X, g["y"], ft, include_scale=False, include_2pi=True)
    assert np.allclose(ex, g["mll_example"], rtol=MLL_RTOL, atol=MLL_ATOL)


def test_g5_boundary_semantics(B):
    g = load_golden("g5_boundaries")
    forest, X, ft = raw(B, g["forest"]), g["X"], g["feat_types"]
    assert np.array_equal(B.bf.pass_through_forest(forest, X, ft), g["leaves"])
    assert np.array_equal(B.bf.forest_gram_matrix(forest, X, X, ft), g["K"])
    for bad in (-1.0, np.nan, np.inf):  # the reference raises inside `1 << int(x)`
        Xb = X.copy()
        Xb[0, 3] = bad
        with pytest.raises(ValueError):
            B.bf.pass_through_forest(forest, Xb, ft)


def test_g6_forest_predict(B):
    g = load_golden("g6_predict")
    forest = raw(B, g["forest"])  # (2, 2, m, L): chains x samples, flattened by forest_predict
    model = (forest, g["noise"], g["scale"])
    mu, var = B.tk.forest_predict(model, (g["X"], g["y"]), g["cand"], g["feat_types"], diag=True)
    assert mu.shape == g["mu"].shape and var.shape == g["var"].shape
    assert np.allclose(mu, g["mu"], rtol=1e-9, atol=1e-9)
    assert np.allclose(var, g["var"], rtol=1e-9, atol=1e-9)
    mu2, full = B.tk.forest_predict(model, (g["X"], g["y"]), g["cand"], g["feat_types"], diag=False)
    assert full.shape == g["var_full"].shape and np.array_equal(mu2, mu)
    assert np.allclose(full, g["var_full"], rtol=1e-9, atol=1e-9)
    K_xX = B.bf.batched_forest_gram_matrix(forest.reshape(-1, *forest.shape[-2:]), g["cand"], g["X"], g["feat_types"])
    assert np.array_equal(K_xX, g["K_xX"])
    mix_mu, mix_var = B.tk.mixture_of_gaussians_as_normal(mu, var)
    assert np.allclose(mix_mu, g["mix_mu"], rtol=1e-9, atol=1e-9) and np.allclose(mix_var, g["mix_var"], rtol=1e-8, atol=1e-9)


def test_g7_tree_function_config1(B):
    g = load_golden("g7_tree_function")
    forest, leaf_values, f = B.syn.tree_function(dim=5, m=50, function_seed=1)
    assert np.array_equal(forest, raw(B, g["forest"])) and np.array_equal(leaf_values, g["leaf_values"])
    assert np.array_equal(f(g["X"]), g["y"])  # bit-exact: integer gather + the same summation order


def test_g8_batched_mll(B):
    g = load_golden("g8_batched_mll")
    forest, X, y, ft = raw(B, g["forest"]), g["X"], g["y"], g["feat_types"]
    assert np.array_equal(B.bf.batched_forest_gram_matrix(forest, X, X, ft), g["K"])
    ex = B.fit.batched_mll(forest, g["noise"], None, X, y, ft, include_scale=False, include_2pi=True)
    sa = B.fit.batched_mll(forest, g["noise"], g["scale"], X, y, ft, include_scale=True, include_2pi=False)
    assert np.allclose(ex, g["mll_example"], rtol=MLL_RTOL, atol=MLL_ATOL)
    assert np.allclose(sa, g["mll_sampler"], rtol=MLL_RTOL, atol=MLL_ATOL)
    # chunking must not change a single bit
    one = B.fit.batched_mll(forest, g["noise"], g["scale"], X, y, ft, include_scale=True, include_2pi=False, chunk=1)
    three = B.fit.batched_mll(forest, g["noise"], g["scale"], X, y, ft, include_scale=True, include_2pi=False, chunk=3)
    assert np.array_equal(one, sa) and np.array_equal(three, sa)


def test_g10_forests_from_the_reference_mcmc_sampler(B):
    """Posterior forests produced by the reference's own sampler (bark_sampler.py:121-284)."""
    g = load_golden("g10_mcmc_posterior_forests")
    forest = raw(B, g["forest"])  # (chains, samples, m, L)
    flat = forest.reshape(-1, *forest.shape[-2:])
    X, y, ft = g["X"], g["y"], g["feat_types"]
    for b in range(flat.shape[0]):
        assert np.array_equal(B.bf.pass_through_forest(flat[b], X, ft), g["leaves"][b])
    assert np.array_equal(B.bf.batched_forest_gram_matrix(flat, X, X, ft), g["K"])
    assert np.array_equal(B.bf.batched_forest_gram_matrix_no_null(flat, X, X, ft), g["K_no_null"])
    ex = B.fit.mll((forest, g["noise"], g["scale"]), (X, y), ft)
    sa = B.fit.batched_mll(forest, g["noise"], g["scale"], X, y, ft, include_scale=True, include_2pi=False)
    assert np.allclose(ex, g["mll_example"], rtol=MLL_RTOL, atol=MLL_ATOL)
    assert np.allclose(sa, g["mll_sampler"], rtol=MLL_RTOL, atol=MLL_ATOL)
    # continue one chain on the device: swap tree 0 of the last sample for tree 0 of the first sample
    state = B.fit.ChainState.from_forest(flat[-1], float(g["noise"].reshape(-1)[-1]), float(g["scale"].reshape(-1)[-1]),
                                         X, y, ft)
    assert np.isclose(state.mll, g["mll_sampler"][-1], rtol=1e-9)
    swapped = flat[-1].copy()
    swapped[0] = flat[0][0]
    new_mll = state.propose_tree(flat[-1][0], flat[0][0], X, ft, float(g["scale"].reshape(-1)[-1]), flat.shape[1])
    want = B.orc.batched_mll(swapped[None], g["noise"].reshape(-1)[-1:], g["scale"].reshape(-1)[-1:], X, y, ft,
                             include_scale=True, include_2pi=False)[0]
    assert np.isclose(new_mll, want, rtol=1e-9)


# ------------------------------------------------------------------ oracle on seeded inputs -------
def test_c2_n1024_against_oracle(B):
    X, y, bounds, ft = B.syn.unit_cube_problem(1024, 8, seed=1024)
    F = B.syn.sample_prior_forests(2, 50, bounds, ft, seed=1024)
    for b in range(2):
        assert np.array_equal(B.bf.pass_through_forest(F[b], X, ft), B.orc.pass_through_forest(F[b], X, ft))
    K = B.bf.batched_forest_gram_matrix(F, X, X, ft)
    assert np.array_equal(K, B.orc.batched_forest_gram_matrix(F, X, X, ft))
    noise, scale = np.array([0.1, 0.07]), np.array([1.0, 1.3])
    got = B.fit.batched_mll(F, noise, scale, X, y, ft, include_scale=True, include_2pi=True)
    want = B.orc.batched_mll(F, noise, scale, X, y, ft, include_scale=True, include_2pi=True)
    assert np.allclose(got, want, rtol=MLL_RTOL, atol=MLL_ATOL), (got, want)


def test_mixed_types_and_ragged_sizes_against_oracle(B):
    for N, M in ((1, 1), (3, 130), (127, 129), (200, 65), (513, 64)):
        X, y, bounds, ft = B.syn.mixed_problem(N, seed=N)
        if N == 1:
            y = np.array([[0.3]])  # a single point cannot be standardised
        X2, _, _, _ = B.syn.mixed_problem(M, seed=1000 + M)
        F = B.syn.sample_prior_forests(2, 13, bounds, ft, seed=N)  # m = 13: one padded byte lane
        assert np.array_equal(B.bf.batched_forest_gram_matrix(F, X, X2, ft),
                              B.orc.batched_forest_gram_matrix(F, X, X2, ft))
        assert np.array_equal(B.bf.pass_through_forest(F[1], X2, ft), B.orc.pass_through_forest(F[1], X2, ft))
        noise = np.array([0.05, 0.2])
        got = B.fit.batched_mll(F, noise, None, X, y, ft, include_scale=False, include_2pi=True)
        want = B.orc.batched_mll(F, noise, None, X, y, ft, include_scale=False, include_2pi=True)
        assert np.allclose(got, want, rtol=MLL_RTOL, atol=MLL_ATOL), (N, got, want)


def test_gram_with_odd_row_lengths(B):
    """An N x M output with odd M has every other row starting 8 bytes off a 16-byte boundary (and, with odd N M, every other
    forest too): gram_kernel writes those rows as shifted pairs (gram.hip).  Bit-exact against the oracle, sizes on both sides
    of the 64- and 128-column tile edges."""
    for N, M in ((5, 257), (9, 127), (64, 255), (33, 1), (2, 3), (257, 385), (130, 129)):
        X, _, bounds, ft = B.syn.mixed_problem(N, seed=7 * N + M)
        X2 = B.syn.mixed_problem(M, seed=11 * M + N)[0]
        F = B.syn.sample_prior_forests(3, 21, bounds, ft, seed=N + M)
        assert np.array_equal(B.bf.batched_forest_gram_matrix(F, X, X2, ft), B.orc.batched_forest_gram_matrix(F, X, X2, ft)), (N, M)
        assert np.array_equal(B.bf.forest_gram_matrix(F[2], X2, X2, ft), B.orc.forest_gram_matrix(F[2], X2, X2, ft)), M


def test_deep_trees_and_many_leaves(B):
    rng = np.random.default_rng(5)
    X = rng.uniform(size=(300, 6))
    ft = np.full(6, 2)
    deep = B.syn.full_binary_forest(9, 6, 5, rng)  # 32 leaves / tree
    assert np.array_equal(B.bf.pass_through_forest(deep, X, ft), B.orc.pass_through_forest(deep, X, ft))
    assert np.array_equal(B.bf.forest_gram_matrix(deep, X, X, ft), B.orc.forest_gram_matrix(deep, X, X, ft))
    wide = B.syn.full_binary_forest(5, 6, 8, rng, node_limit=512)  # 256 leaves / tree: 8-bit ids, general compare
    assert np.array_equal(B.bf.pass_through_forest(wide, X, ft), B.orc.pass_through_forest(wide, X, ft))
    assert np.array_equal(B.bf.forest_gram_matrix(wide, X, X, ft), B.orc.forest_gram_matrix(wide, X, X, ft))
    # 512 leaves per tree do not fit a byte: the one-hot code takes over (2 trees x 512 bits = 32 words)
    huge = B.syn.full_binary_forest(2, 6, 9, rng, node_limit=1024)
    assert np.array_equal(B.bf.pass_through_forest(huge, X, ft), B.orc.pass_through_forest(huge, X, ft))
    assert np.array_equal(B.bf.forest_gram_matrix(huge, X, X, ft), B.orc.forest_gram_matrix(huge, X, X, ft))
    # byte code with ids >= 128 (50 trees x 256 leaves would need 400 words as bits)
    wide50 = B.syn.full_binary_forest(50, 6, 8, rng, node_limit=512)
    assert np.array_equal(B.bf.forest_gram_matrix(wide50, X, X, ft), B.orc.forest_gram_matrix(wide50, X, X, ft))
    deep50 = B.syn.full_binary_forest(50, 6, 5, rng)  # 32 leaves: byte code with ids < 128
    y = rng.standard_normal((300, 1))
    for forest in (deep50, wide50):
        got = B.fit.batched_mll(forest[None], [0.1], [1.0], X, y, ft, include_scale=True, include_2pi=True)
        want = B.orc.batched_mll(forest[None], [0.1], [1.0], X, y, ft, include_scale=True, include_2pi=True)
        assert np.allclose(got, want, rtol=MLL_RTOL, atol=MLL_ATOL)


def _scrambled_forest(rng, m, L, ft, max_leaves):
    """Random tree topologies placed in arbitrary container slots (as tree_proposals.py:46-58 leaves them after
    grow/prune cycles), unused slots filled with garbage records, thresholds of every feature type."""
    from bark_amd.forest import NODE_RECORD_DTYPE

    d = len(ft)
    forest = np.zeros((m, L), dtype=NODE_RECORD_DTYPE)
    for t in range(m):
        tree = forest[t]
        # garbage everywhere first: never reachable from the root, must be ignored
        tree["is_leaf"] = rng.integers(0, 2, L)
        tree["feature_idx"] = rng.integers(0, 2**31, L)
        tree["threshold"] = rng.standard_normal(L).astype(np.float32)
        tree["left"] = rng.integers(0, 2**31, L)
        tree["right"] = rng.integers(0, 2**31, L)
        tree["active"] = 0
        free = list(rng.permutation(np.arange(1, L)))
        tree[0] = (1, 0, 0, 0, 0, 0xFFFFFFFF, 0, 1)
        leaves = [0]
        for _ in range(int(rng.integers(0, max_leaves))):
            if len(free) < 2:
                break
            node = leaves.pop(int(rng.integers(len(leaves))))
            f = int(rng.integers(d))
            if ft[f] == 0:      # categorical: bitmask of categories as float32 (domain.py:31-34)
                thr = float(rng.integers(0, 32))
            elif ft[f] == 1:    # integer feature: x <= thr
                thr = float(rng.integers(0, 10))
            else:
                thr = float(np.float32(rng.uniform()))
            left, right = int(free.pop()), int(free.pop())
            depth = int(tree[node]["depth"])
            tree[node] = (0, f, thr, left, right, tree[node]["parent"], depth, 1)
            tree[left] = (1, 0, 0, 0, 0, node, depth + 1, 1)
            tree[right] = (1, 0, 0, 0, 0, node, depth + 1, 1)
            leaves += [left, right]
    return forest


def test_fuzz_scrambled_containers_and_boundary_points(B):
    """Seeded structural fuzz: leaves, Gram and no-null Gram bit-identical to the oracle for random topologies in
    scrambled containers, with points planted exactly on thresholds and at +-inf / NaN / -0.0."""
    rng = np.random.default_rng(2024)
    ft = np.array([2, 2, 1, 0, 2, 0, 1, 2])
    for case in range(12):
        m = int(rng.integers(1, 23))
        L = int(rng.choice([7, 16, 100, 255]))
        forest = np.stack([_scrambled_forest(rng, m, L, ft, max_leaves=1 + case * 2) for _ in range(3)])
        N, M = int(rng.integers(1, 300)), int(rng.integers(1, 200))

        def points(n):
            X = np.empty((n, 8))
            for f, kind in enumerate(ft):
                if kind == 0:
                    X[:, f] = rng.integers(0, 5, n)
                elif kind == 1:
                    X[:, f] = rng.integers(0, 11, n)
                else:
                    X[:, f] = rng.uniform(size=n)
            # plant continuous / integer values exactly on thresholds the forests use, and non-finite values
            thr = forest["threshold"][forest["active"] == 1].astype(np.float64)
            for f in (0, 1, 4, 7, 2, 6):
                hit = rng.random(n) < 0.15
                X[hit, f] = rng.choice(thr, hit.sum())
                X[rng.random(n) < 0.02, f] = rng.choice([np.inf, -np.inf, np.nan, -0.0, np.nextafter(0.5, 1)])
            return X

        X1, X2 = points(N), points(M)
        for b in range(3):
            assert np.array_equal(B.bf.pass_through_forest(forest[b], X1, ft), B.orc.pass_through_forest(forest[b], X1, ft))
        assert np.array_equal(B.bf.batched_forest_gram_matrix(forest, X1, X2, ft),
                              B.orc.batched_forest_gram_matrix(forest, X1, X2, ft)), case
        assert np.array_equal(B.bf.batched_forest_gram_matrix_no_null(forest, X1, X1, ft),
                              B.orc.batched_forest_gram_matrix_no_null(forest, X1, X1, ft)), case


def test_wide_feature_matrix_walks_from_global_memory(B):
    """d = 40 > 31: the point rows no longer fit the walk kernel's LDS tile (leaf_walk_kernel<., false>)."""
    rng = np.random.default_rng(40)
    d, N = 40, 333
    X = rng.uniform(size=(N, d))
    ft = np.full(d, 2)
    ft[[3, 17]] = 0
    X[:, [3, 17]] = rng.integers(0, 5, size=(N, 2))
    bounds = np.tile([[0.0, 1.0]], (d, 1))
    bounds[[3, 17]] = [0.0, 31.0]
    F = B.syn.sample_prior_forests(2, 20, bounds, ft, seed=40, alpha=0.95, beta=1.0)
    for b in range(2):
        assert np.array_equal(B.bf.pass_through_forest(F[b], X, ft), B.orc.pass_through_forest(F[b], X, ft))
    assert np.array_equal(B.bf.batched_forest_gram_matrix(F, X, X, ft), B.orc.batched_forest_gram_matrix(F, X, X, ft))


def test_input_layout_and_degenerate_shapes(B):
    """What `DataFrame.to_numpy()` and sampler views hand over: Fortran order, float32, strided forests,
    (N,) targets; and the smallest containers (one tree, one node slot)."""
    X, y, bounds, ft = B.syn.mixed_problem(150, seed=12)
    F = B.syn.sample_prior_forests(4, 9, bounds, ft, seed=12)
    want_K = B.orc.batched_forest_gram_matrix(F, X, X, ft)
    assert np.array_equal(B.bf.batched_forest_gram_matrix(F, np.asfortranarray(X), np.asfortranarray(X), ft), want_K)
    X32 = X.astype(np.float32)
    assert np.array_equal(B.bf.batched_forest_gram_matrix(F, X32, X32, ft),
                          B.orc.batched_forest_gram_matrix(F, X32.astype(np.float64), X32.astype(np.float64), ft))
    chains = np.stack([F, F[::-1]])[:, ::2]          # (2, 2, m, L) non-contiguous view, like forest[:, -1]
    noise = np.array([[0.1, 0.2], [0.3, 0.15]])
    got = B.fit.mll((chains, noise, None), (X, y[:, 0]), ft)   # y as (N,)
    want = B.orc.batched_mll(np.ascontiguousarray(chains), noise, None, X, y, ft, include_scale=False, include_2pi=True)
    assert np.allclose(got, want, rtol=MLL_RTOL, atol=MLL_ATOL)
    one = F[:1, :1]                                     # a single tree
    assert np.array_equal(B.bf.forest_gram_matrix(one[0], X, X, ft), B.orc.forest_gram_matrix(one[0], X, X, ft))
    root_only = B.bf.create_empty_forest(3, node_limit=1)  # L = 1: every tree is just its root
    K = B.bf.forest_gram_matrix(root_only, X, X, ft)
    assert np.all(K == 1.0) and np.array_equal(B.bf.pass_through_forest(root_only, X, ft), np.zeros((150, 3), np.uint32))
    with pytest.raises(ValueError):
        B.bf.forest_gram_matrix(F[0], X[:, :3], X[:, :3], ft)   # wrong feature count
    with pytest.raises(TypeError):
        B.bf.forest_gram_matrix(np.zeros((2, 3)), X, X, ft)     # not a node record array


def test_posterior_against_oracle_ragged(B):
    X, y, bounds, ft = B.syn.mixed_problem(300, seed=9)
    cand, _, _, _ = B.syn.mixed_problem(257, seed=10)
    F = B.syn.sample_prior_forests(3, 50, bounds, ft, seed=9)
    noise, scale = np.array([0.1, 0.05, 0.2]), np.array([1.0, 0.6, 1.4])
    mu, var = B.tk.forest_predict((F, noise, scale), (X, y), cand, ft)
    mu0, var0 = B.orc.forest_predict((F, noise, scale), (X, y), cand, ft)
    assert np.allclose(mu, mu0, rtol=1e-9, atol=1e-9) and np.allclose(var, var0, rtol=1e-9, atol=1e-9)


def test_beyond_4096_ragged_with_candidates(B):
    """N = 5003 (40 block rows, ragged), mixed cat/int/cont, C = 300 candidates in the same sweep;
    reference arithmetic (LU inv) via the oracle."""
    X, y, bounds, ft = B.syn.mixed_problem(5003, seed=55)
    cand, _, _, _ = B.syn.mixed_problem(300, seed=56)
    F = B.syn.sample_prior_forests(2, 50, bounds, ft, seed=55)
    noise, scale = np.array([0.1, 0.06]), np.array([1.0, 1.2])
    mu, var = B.tk.forest_predict((F, noise, scale), (X, y), cand, ft)
    mu0, var0 = B.orc.forest_predict((F, noise, scale), (X, y), cand, ft)
    assert np.allclose(mu, mu0, rtol=1e-9, atol=1e-9) and np.allclose(var, var0, rtol=1e-9, atol=1e-9)
    got = B.fit.batched_mll(F, noise, scale, X, y, ft, include_scale=True, include_2pi=False)
    want = B.orc.batched_mll(F, noise, scale, X, y, ft, include_scale=True, include_2pi=False)
    assert np.allclose(got, want, rtol=MLL_RTOL, atol=MLL_ATOL), (got, want)


def test_kernel_inverse_for_acquisition_builder(B):
    """opt_model.py:54-59,101: inv(scale * K_no_null + (1e-6+noise) I) and scale-free K_inv @ y, ragged N."""
    g = load_golden("g3_prior_mixed_n257")
    forest, X, y, ft = raw(B, g["forest"]), g["X"], g["y"], g["feat_types"]
    noise, scale = g["noise"], g["scale"]
    for no_null, Kref in ((True, g["K_no_null"]), (False, g["K"])):
        K_s = scale[:, None, None] * Kref + (1e-6 + noise[:, None, None]) * np.eye(257)
        want = np.linalg.inv(K_s)
        K_inv, K_inv_y, logdet = B.fit.batched_kernel_inverse(forest, noise, scale, X, y, ft, no_null=no_null)
        assert K_inv.shape == (3, 257, 257)
        assert np.allclose(K_inv, want, rtol=1e-8, atol=1e-9)
        assert np.allclose(K_inv_y, (want @ y)[..., 0], rtol=1e-8, atol=1e-9)
        assert np.allclose(logdet, np.linalg.slogdet(K_s)[1], rtol=1e-10)
        assert np.allclose(K_inv @ K_s, np.eye(257)[None], atol=1e-8)
        if not no_null:  # the same three results from the R x R leaf-space system
            L_inv, L_inv_y, L_logdet = B.fit.batched_kernel_inverse(forest, noise, scale, X, y, ft, no_null=False,
                                                                    method="leafspace")
            assert np.allclose(L_inv, want, rtol=1e-8, atol=1e-9)
            assert np.allclose(L_inv_y, (want @ y)[..., 0], rtol=1e-8, atol=1e-9)
            assert np.allclose(L_logdet, np.linalg.slogdet(K_s)[1], rtol=1e-10)
            one = B.fit.batched_kernel_inverse(forest, noise, scale, X, y, ft, no_null=False, method="leafspace", chunk=1)
            assert np.array_equal(one[0], L_inv)  # chunking does not change the arithmetic
    with pytest.raises(ValueError):
        B.fit.batched_kernel_inverse(forest, noise, scale, X, y, ft, no_null=True, method="leafspace")


def test_g9_woodbury_updates(B):
    """quick_inverse.py:13-33 on device vs the reference's outputs (tests/bark_fitting/test_quick_inverse.py:13-52)."""
    g = load_golden("g9_woodbury")
    qi = B.fit.quick_inverse
    for i in range(3):
        A, U, A_inv, logdet = g[f"A{i}"], g[f"U{i}"], g[f"Ainv{i}"], float(g[f"logdet{i}"])
        assert np.allclose(qi.low_rank_inv_update(A_inv, U), g[f"inv_add{i}"], rtol=1e-9, atol=1e-11)
        assert np.allclose(qi.low_rank_inv_update(A_inv, U, subtract=True), g[f"inv_sub{i}"], rtol=1e-9, atol=1e-11)
        assert np.isclose(qi.low_rank_det_update(A_inv, U, logdet), g[f"det_add{i}"], rtol=1e-11)
        assert np.isclose(qi.low_rank_det_update(A_inv, U, logdet, subtract=True), g[f"det_sub{i}"], rtol=1e-11)
        # the reference's own assertions
        assert np.isclose(qi.low_rank_inv_update(A_inv, U), np.linalg.inv(A + U @ U.T)).all()
        assert np.isclose(qi.low_rank_inv_update(A_inv, U, subtract=True), np.linalg.inv(A - U @ U.T)).all()


def test_g2_sampler_tree_swap_chain_on_device(B):
    """The per-tree step of the sampler (bark_sampler.py:233-257) with K_inv resident on the GPU:
    subtract the old tree's leaf vectors, add the new tree's, compare with the exact recomputation
    (tests/bark_fitting/test_quick_inverse.py:55-101)."""
    torch = B.torch
    g = load_golden("g2_two_tree_kat")
    qi = B.fit.quick_inverse
    forest, new_nodes, x, ft = raw(B, g["forest"]), raw(B, g["new_nodes"]), g["x"], g["feat_types"]
    s = np.sqrt(0.5 / 2)
    cur = s * B.bf.get_leaf_vectors(forest[0], x, ft)
    new = s * B.bf.get_leaf_vectors(new_nodes, x, ft)
    K_inv = torch.from_numpy(g["K_inv"]).cuda()
    logdet = float(g["K_logdet"])
    inv1 = qi.low_rank_inv_update(K_inv, torch.from_numpy(cur).cuda(), subtract=True)
    det1 = qi.low_rank_det_update(K_inv, torch.from_numpy(cur).cuda(), logdet, subtract=True)
    inv2 = qi.low_rank_inv_update(inv1, torch.from_numpy(new).cuda())
    det2 = qi.low_rank_det_update(inv1, torch.from_numpy(new).cuda(), det1)
    assert inv2.is_cuda
    assert np.allclose(inv1.cpu().numpy(), g["inv_after_subtract"]) and np.isclose(float(det1), g["det_after_subtract"])
    assert np.isclose(float(det2), g["K_swapped_logdet"]) and np.isclose(inv2.cpu().numpy(), g["K_swapped_inv"]).all()
    y = torch.linspace(-1, 1, 20, dtype=torch.float64).reshape(-1, 1).cuda()
    want = B.orc.mll(g["K_swapped_inv"], float(g["K_swapped_logdet"]), y.cpu().numpy())
    assert np.isclose(float(qi.mll(inv2, det2, y)), want, rtol=1e-10)


def test_fused_tree_swap_matches_reference_chain(B):
    """ChainState.propose/accept vs the reference's subtract -> add -> mll chain (bark_sampler.py:233-264),
    evaluated by the oracle, over a sequence of tree swaps; K_inv stays resident on the GPU."""
    rng = np.random.default_rng(8)
    X, y, bounds, ft = B.syn.mixed_problem(400, seed=8)
    m, scale, noise = 20, 1.3, 0.1
    forest = B.syn.sample_prior_forests(1, m, bounds, ft, seed=8)[0]
    fresh = B.syn.sample_prior_forests(1, m, bounds, ft, seed=9, alpha=0.95, beta=1.0)[0]
    state = B.fit.ChainState.from_forest(forest, noise, scale, X, y, ft)
    # oracle state, exactly as bark_sampler.py:153-162
    K = scale * B.orc.forest_gram_matrix(forest, X, X, ft) + (1e-6 + noise) * np.eye(400)
    K_inv = np.linalg.inv(K)
    logdet = np.linalg.slogdet(K)[1]
    assert np.isclose(state.mll, B.orc.mll(K_inv, logdet, y), rtol=1e-10)
    s = np.sqrt(scale / m)
    for t_idx in range(6):
        new_nodes = fresh[t_idx]
        cur_lv = s * B.orc.get_leaf_vectors(forest[t_idx], X, ft)
        new_lv = s * B.orc.get_leaf_vectors(new_nodes, X, ft)
        inv1 = B.orc.low_rank_inv_update(K_inv, cur_lv, subtract=True)
        det1 = B.orc.low_rank_det_update(K_inv, cur_lv, logdet, subtract=True)
        inv2 = B.orc.low_rank_inv_update(inv1, new_lv)
        det2 = B.orc.low_rank_det_update(inv1, new_lv, det1)
        want = B.orc.mll(inv2, det2, y)
        got = state.propose_tree(forest[t_idx], new_nodes, X, ft, scale, m)
        assert np.isclose(got, want, rtol=1e-9, atol=1e-9), (t_idx, got, want)
        if rng.uniform() < 0.7:  # accept
            state.accept()
            K_inv, logdet = inv2, det2
            forest = forest.copy()
            forest[t_idx] = new_nodes
        assert np.isclose(state.mll, B.orc.mll(K_inv, logdet, y), rtol=1e-9)
    # after the swaps the resident inverse is the exact inverse of the current forest's kernel
    K = scale * B.orc.forest_gram_matrix(forest, X, X, ft) + (1e-6 + noise) * np.eye(400)
    assert np.allclose(state.K_inv.cpu().numpy(), np.linalg.inv(K), rtol=1e-7, atol=1e-8)
    assert np.isclose(state.logdet, np.linalg.slogdet(K)[1], rtol=1e-10)
    with pytest.raises(RuntimeError):
        state.accept()
    # second half of the sampler step (bark_sampler.py:266-272): new (noise, scale), inverse rebuilt on accept
    new_noise, new_scale = 0.07, 0.9
    K2 = new_scale * B.orc.forest_gram_matrix(forest, X, X, ft) + (1e-6 + new_noise) * np.eye(400)
    K2_inv, K2_logdet = np.linalg.inv(K2), np.linalg.slogdet(K2)[1]
    before = state.mll
    got = state.propose_noise_scale(forest, new_noise, new_scale, X, ft)
    assert np.isclose(got, B.orc.mll(K2_inv, K2_logdet, y), rtol=1e-9)
    assert state.mll == before  # a proposal does not modify the state
    state.accept()
    assert np.isclose(state.mll, got, rtol=1e-10) and np.isclose(state.logdet, K2_logdet, rtol=1e-10)
    assert np.allclose(state.K_inv.cpu().numpy(), K2_inv, rtol=1e-7, atol=1e-8)
    # and tree proposals continue from the rebuilt state
    cur_lv = np.sqrt(new_scale / m) * B.orc.get_leaf_vectors(forest[7], X, ft)
    new_lv = np.sqrt(new_scale / m) * B.orc.get_leaf_vectors(fresh[7], X, ft)
    inv1 = B.orc.low_rank_inv_update(K2_inv, cur_lv, subtract=True)
    det1 = B.orc.low_rank_det_update(K2_inv, cur_lv, K2_logdet, subtract=True)
    want = B.orc.mll(B.orc.low_rank_inv_update(inv1, new_lv), B.orc.low_rank_det_update(inv1, new_lv, det1), y)
    assert np.isclose(state.propose_tree(forest[7], fresh[7], X, ft, new_scale, m), want, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("N", [300, 301])
def test_chain_batch_matches_single_chains(B, N):
    """Several chains in one call give exactly the single-chain results, chain by chain, including a partial
    accept.  N = 300: the chain index is a grid dimension of every kernel; N = 301 (odd): per-chain sequences on
    their own streams."""
    X, y, bounds, ft = B.syn.mixed_problem(N, seed=14)
    m, nc = 12, 3
    forests = B.syn.sample_prior_forests(nc, m, bounds, ft, seed=14)
    fresh = B.syn.sample_prior_forests(nc, m, bounds, ft, seed=15, alpha=0.95, beta=1.0)
    noise, scale = np.array([0.1, 0.05, 0.2]), np.array([1.0, 1.4, 0.7])
    batch = B.fit.ChainBatch.from_forests(forests, noise, scale, X, y, ft)
    singles = [B.fit.ChainState.from_forest(forests[b], noise[b], scale[b], X, y, ft) for b in range(nc)]
    assert np.allclose(batch.mll, [s.mll for s in singles], rtol=1e-12)
    for t_idx in range(4):
        got = batch.propose_trees(forests[:, t_idx], fresh[:, t_idx], X, ft, scale, m)
        want = [singles[b].propose_tree(forests[b, t_idx], fresh[b, t_idx], X, ft, scale[b], m) for b in range(nc)]
        assert np.allclose(got, want, rtol=1e-12, atol=1e-12), (t_idx, got, want)
        mask = np.array([True, t_idx % 2 == 0, False])
        batch.accept(mask)
        for b in range(nc):
            if mask[b]:
                singles[b].accept()
                forests[b, t_idx] = fresh[b, t_idx]
        assert np.allclose(batch.mll, [s.mll for s in singles], rtol=1e-12)
    for b in range(nc):
        assert np.array_equal(batch.K_inv[b].cpu().numpy(), singles[b].K_inv.cpu().numpy())
    # and the states are the exact inverses of the chains' current kernels
    for b in range(nc):
        K = scale[b] * B.orc.forest_gram_matrix(forests[b], X, X, ft) + (1e-6 + noise[b]) * np.eye(N)
        assert np.allclose(batch.K_inv[b].cpu().numpy(), np.linalg.inv(K), rtol=1e-7, atol=1e-8)
    with pytest.raises(RuntimeError):
        batch.accept(True)
    with pytest.raises(ValueError):
        batch.propose_trees(forests[:2, 0], fresh[:2, 0], X, ft, scale, m)


def test_woodbury_large_against_oracle(B):
    """N = 1500, r = 7 / 33 / 64: one-hot style and dense U, symmetric SPD K_inv."""
    rng = np.random.default_rng(3)
    N = 1500
    A = rng.standard_normal((N, N)) / np.sqrt(N)
    K_inv = np.linalg.inv(A @ A.T + np.eye(N))
    _, logdet = np.linalg.slogdet(A @ A.T + np.eye(N))
    qi = B.fit.quick_inverse
    for r in (7, 33, 64):
        U = rng.standard_normal((N, r)) * 0.05
        for sub in (False, True):
            got = qi.low_rank_inv_update(K_inv, U, subtract=sub)
            want = B.orc.low_rank_inv_update(K_inv, U, subtract=sub)
            assert np.allclose(got, want, rtol=1e-9, atol=1e-11)
            fast = qi.low_rank_inv_update(K_inv, U, subtract=sub, assume_symmetric=True)
            assert np.allclose(fast, want, rtol=1e-9, atol=1e-11)
            assert np.isclose(qi.low_rank_det_update(K_inv, U, logdet, subtract=sub),
                              B.orc.low_rank_det_update(K_inv, U, logdet, subtract=sub), rtol=1e-12)
    with pytest.raises(ValueError):
        qi.low_rank_inv_update(K_inv, rng.standard_normal((N, 65)))
    # odd N (rows not 16-byte aligned -> scalar loads) that is not a multiple of the 8-row / 128-column tiling
    N = 333
    A = rng.standard_normal((N, N)) / np.sqrt(N)
    K_inv = np.linalg.inv(A @ A.T + np.eye(N))
    _, logdet = np.linalg.slogdet(A @ A.T + np.eye(N))
    for r in (1, 5, 16):
        U = rng.standard_normal((N, r)) * 0.05
        for sub in (False, True):
            want = B.orc.low_rank_inv_update(K_inv, U, subtract=sub)
            assert np.allclose(qi.low_rank_inv_update(K_inv, U, subtract=sub), want, rtol=1e-9, atol=1e-11)
            assert np.allclose(qi.low_rank_inv_update(K_inv, U, subtract=sub, assume_symmetric=True), want, rtol=1e-9,
                               atol=1e-11)
            assert np.isclose(qi.low_rank_det_update(K_inv, U, logdet, subtract=sub),
                              B.orc.low_rank_det_update(K_inv, U, logdet, subtract=sub), rtol=1e-12)


def test_single_large_matrix_split_k_path(B):
    """B = 1 and B = 3 at N = 2100 (17 block rows): under-filled steps take the split-K panel path
    (panel_split_kernel + panel_reduce_kernel) and tiles are interleaved over the XCDs."""
    X, y, bounds, ft = B.syn.mixed_problem(2100, seed=21)
    cand, _, _, _ = B.syn.mixed_problem(150, seed=22)
    F = B.syn.sample_prior_forests(3, 50, bounds, ft, seed=21)
    noise, scale = np.array([0.1, 0.05, 0.2]), np.array([1.0, 0.8, 1.3])
    want = B.orc.batched_mll(F, noise, scale, X, y, ft, include_scale=True, include_2pi=True)
    for nb in (1, 3):
        got = B.fit.batched_mll(F[:nb], noise[:nb], scale[:nb], X, y, ft, include_scale=True, include_2pi=True)
        assert np.allclose(got, want[:nb], rtol=MLL_RTOL, atol=MLL_ATOL), (nb, got, want)
    mu, var = B.tk.forest_predict((F[:1], noise[:1], scale[:1]), (X, y), cand, ft)
    mu0, var0 = B.orc.forest_predict((F[:1], noise[:1], scale[:1]), (X, y), cand, ft)
    assert np.allclose(mu, mu0, rtol=1e-9, atol=1e-9) and np.allclose(var, var0, rtol=1e-9, atol=1e-9)
    # same call twice: bit-identical (slabs are reduced in a fixed order, no atomics)
    again = B.fit.batched_mll(F[:1], noise[:1], scale[:1], X, y, ft, include_scale=True, include_2pi=True)
    first = B.fit.batched_mll(F[:1], noise[:1], scale[:1], X, y, ft, include_scale=True, include_2pi=True)
    assert np.array_equal(again, first)


def test_many_small_matrices_and_chunk_invariance(B):
    """B = 288 forests at N = 700 (6 block rows): a filled chip of small matrices against the oracle's LU route, and
    the same forests factorised 192 at a time.  Which tiles of a step take the split-K route (ragged last round of
    workgroups) depends on the chunk size, so results agree to rounding (1e-12 relative), not bit for bit; for a
    given chunk size they are reproducible exactly.  Also with candidates (materialised A, candidate columns carry
    no right-hand side)."""
    nb, N = 288, 700
    X, y, bounds, ft = B.syn.mixed_problem(N, seed=31)
    F = B.syn.sample_prior_forests(nb, 50, bounds, ft, seed=310)
    rng = np.random.default_rng(31)
    noise, scale = rng.uniform(0.05, 0.2, nb), rng.uniform(0.7, 1.4, nb)
    sub = np.r_[0:24, nb - 8:nb]  # the oracle on a sample; the rest through chunk invariance below
    want = B.orc.batched_mll(F[sub], noise[sub], scale[sub], X, y, ft, include_scale=True, include_2pi=True)
    fused = B.fit.batched_mll(F, noise, scale, X, y, ft, include_scale=True, include_2pi=True)
    assert np.allclose(fused[sub], want, rtol=MLL_RTOL, atol=MLL_ATOL), np.abs(fused[sub] - want).max()
    small = B.fit.batched_mll(F, noise, scale, X, y, ft, include_scale=True, include_2pi=True, chunk=192)
    assert np.allclose(fused, small, rtol=1e-12, atol=0.0)
    assert np.array_equal(small, B.fit.batched_mll(F, noise, scale, X, y, ft, include_scale=True, include_2pi=True, chunk=192))
    # ragged last chunk, other convention
    part = B.fit.batched_mll(F, noise, None, X, y, ft, include_scale=False, include_2pi=True, chunk=200)
    want2 = B.orc.batched_mll(F[sub], noise[sub], None, X, y, ft, include_scale=False, include_2pi=True)
    assert np.allclose(part[sub], want2, rtol=MLL_RTOL, atol=MLL_ATOL)
    # posterior: 200 candidates appended as extra block columns
    cand, _, _, _ = B.syn.mixed_problem(200, seed=32)
    mu, var = B.tk.forest_predict((F, noise, scale), (X, y), cand, ft)
    mu0, var0 = B.orc.forest_predict((F[:24], noise[:24], scale[:24]), (X, y), cand, ft)
    assert np.allclose(mu[:24], mu0, rtol=1e-9, atol=1e-9) and np.allclose(var[:24], var0, rtol=1e-9, atol=1e-9)
    from bark_amd.fitting.mll import _run
    import bark_amd._lib as L
    _, mu8, var8 = _run(F, noise, scale, X, y, ft, L.MLL_INCLUDE_SCALE, cand=cand, chunk=200)
    assert np.allclose(mu8.cpu().numpy(), mu, rtol=1e-10, atol=1e-11) and np.allclose(var8.cpu().numpy(), var, rtol=1e-10, atol=1e-11)


def test_leafspace_mll_equals_dense_and_reference(B):
    """method="leafspace" (R x R system over the leaves) against the golden MLLs, the oracle and the dense path."""
    for name in ("g3_prior_mixed_n257", "g8_batched_mll", "g10_mcmc_posterior_forests", "g4_all_null"):
        g = load_golden(name)
        forest, X, y, ft = raw(B, g["forest"]), g["X"], g["y"], g["feat_types"]
        ex = B.fit.batched_mll(forest, g["noise"], None, X, y, ft, include_scale=False, include_2pi=True, method="leafspace")
        assert np.allclose(ex, g["mll_example"], rtol=MLL_RTOL, atol=MLL_ATOL), name
        if "mll_sampler" in g:
            sa = B.fit.batched_mll(forest, g["noise"], g["scale"], X, y, ft, include_scale=True, include_2pi=False,
                                   method="leafspace")
            assert np.allclose(sa, g["mll_sampler"], rtol=MLL_RTOL, atol=MLL_ATOL), name
    # larger: N = 3000 (ragged), 40 forests in chunks of 16; bushy forest with 32 leaves per tree (R = 1600)
    X, y, bounds, ft = B.syn.mixed_problem(3000, seed=31)
    F = B.syn.sample_prior_forests(40, 50, bounds, ft, seed=31)
    noise, scale = np.linspace(0.05, 0.3, 40), np.linspace(0.7, 1.4, 40)
    dense = B.fit.batched_mll(F, noise, scale, X, y, ft, include_scale=True, include_2pi=True)
    leaf = B.fit.batched_mll(F, noise, scale, X, y, ft, include_scale=True, include_2pi=True, method="leafspace", chunk=16)
    assert np.allclose(leaf, dense, rtol=1e-10, atol=1e-8), np.abs(leaf - dense).max()
    want = B.orc.batched_mll(F[:2], noise[:2], scale[:2], X, y, ft, include_scale=True, include_2pi=True)
    assert np.allclose(leaf[:2], want, rtol=MLL_RTOL, atol=MLL_ATOL)
    rng = np.random.default_rng(2)
    Xc = rng.uniform(size=(700, 6))
    yc = rng.standard_normal((700, 1))
    bushy = B.syn.full_binary_forest(50, 6, 5, rng)[None]
    got = B.fit.batched_mll(bushy, [0.1], [1.0], Xc, yc, np.full(6, 2), include_scale=True, include_2pi=True,
                            method="leafspace")
    want = B.orc.batched_mll(bushy, [0.1], [1.0], Xc, yc, np.full(6, 2), include_scale=True, include_2pi=True)
    assert np.allclose(got, want, rtol=MLL_RTOL, atol=MLL_ATOL)
    # 40 bushy forests at once: 13 block rows x 40 matrices in leaf space — the pipelined schedule of the sweep
    many = np.concatenate([B.syn.full_binary_forest(50, 6, 5, rng)[None] for _ in range(40)])
    nz, sc = np.linspace(0.05, 0.3, 40), np.linspace(0.7, 1.4, 40)
    leaf40 = B.fit.batched_mll(many, nz, sc, Xc, yc, np.full(6, 2), include_scale=True, include_2pi=True, method="leafspace")
    dense40 = B.fit.batched_mll(many, nz, sc, Xc, yc, np.full(6, 2), include_scale=True, include_2pi=True)
    assert np.allclose(leaf40, dense40, rtol=1e-10, atol=1e-8), np.abs(leaf40 - dense40).max()
    with pytest.raises(ValueError):
        B.fit.batched_mll(bushy, [0.1], [1.0], Xc, yc, np.full(6, 2), include_scale=True, include_2pi=True, method="lu")
    # posterior in leaf space: golden (reference arithmetic), then a larger mixed problem against the dense path
    g = load_golden("g6_predict")
    model = (raw(B, g["forest"]), g["noise"], g["scale"])
    mu, var = B.tk.forest_predict(model, (g["X"], g["y"]), g["cand"], g["feat_types"], method="leafspace")
    assert np.allclose(mu, g["mu"], rtol=1e-9, atol=1e-9) and np.allclose(var, g["var"], rtol=1e-9, atol=1e-9)
    cand, _, _, _ = B.syn.mixed_problem(1000, seed=32)
    mu_d, var_d = B.tk.forest_predict((F[:5], noise[:5], scale[:5]), (X, y), cand, ft)
    mu_l, var_l = B.tk.forest_predict((F[:5], noise[:5], scale[:5]), (X, y), cand, ft, method="leafspace")
    assert np.allclose(mu_l, mu_d, rtol=1e-9, atol=1e-10) and np.allclose(var_l, var_d, rtol=1e-8, atol=1e-10)


def test_not_positive_definite_raises(B):
    g = load_golden("g8_batched_mll")
    forest, X, y, ft = raw(B, g["forest"]), g["X"], g["y"], g["feat_types"]
    with pytest.raises(np.linalg.LinAlgError, match="not positive definite"):
        B.fit.batched_mll(forest, np.full(4, -0.5), None, X, y, ft, include_scale=False, include_2pi=True)


@pytest.mark.parametrize("N,dup,m,eps", [(90, 40, 50, 1e-3), (300, 150, 50, 1e-3), (300, 150, 200, 1e-3), (700, 600, 200, 1e-3),
                                         (700, 600, 400, 1e-4),
                                         # 128 < N <= 256: two_block_kernel — the pivot in its first and in its second block
                                         (200, 60, 50, 1e-3), (250, 200, 200, 1e-3), (256, 250, 400, 1e-4)])
def test_not_positive_definite_reports_the_first_bad_pivot(B, N, dup, m, eps):
    """The index in the error is LAPACK potrf's `info`: the 1-based position of the first pivot that is not positive when the
    matrix is eliminated in order — also when it lies in a later 4-pivot block of factor16, a later 16 x 16 sub-block of
    diag_kernel or a later 128-row block step of the sweep (badbits, base_index, j * 128).  K - eps I with one duplicated point:
    the elimination fails at the duplicate (pivot ~ -2 eps) or earlier, where the forest's rank runs out — positions 40, 56, 150,
    162 and 390 for these inputs.  The expected index comes from an unblocked numpy elimination of the same matrix and must be a
    clear failure (|pivot| > 1e-4, six orders above what rounding moves), so it cannot depend on the order of operations."""
    import re

    X, y, bounds, ft = B.syn.mixed_problem(N, seed=N)
    X = X.copy()
    X[dup - 1] = X[0]
    F = B.syn.sample_prior_forests(1, m, bounds, ft, seed=N)
    noise = -1e-6 - eps
    A, want = B.orc.forest_gram_matrix(F[0], X, X, ft) + (1e-6 + noise) * np.eye(N), 0
    for k in range(N):
        if not A[k, k] > 0.0:
            want = k + 1
            assert A[k, k] < -1e-4, A[k, k]
            break
        A[k + 1:, k + 1:] -= np.outer(A[k + 1:, k], A[k, k + 1:]) / A[k, k]
    assert want > 0
    with pytest.raises(np.linalg.LinAlgError, match="not positive definite") as exc:
        B.fit.batched_mll(F, [noise], None, X, y, ft, include_scale=False, include_2pi=True)
    assert int(re.search(r"pivot (\d+)", str(exc.value)).group(1)) == want, (str(exc.value), want)
    if 128 < N <= 256:  # ... two_block_kernel above (one matrix: eight waves); 16 copies of the forest as well
        from bark_amd.fitting import schedule_plan

        assert schedule_plan(N, 16, m=m)["schedule"] == "two_block"
        with pytest.raises(np.linalg.LinAlgError, match="not positive definite") as exc:
            B.fit.batched_mll(np.repeat(F, 16, axis=0), np.full(16, noise), None, X, y, ft, include_scale=False, include_2pi=True)
        assert int(re.search(r"pivot (\d+)", str(exc.value)).group(1)) == want, (str(exc.value), want)


def test_argument_validation_across_the_api(B):
    """Bad shapes / unsupported options raise Python exceptions (SURVEY §8b error behaviour); nothing aborts."""
    X, y, bounds, ft = B.syn.mixed_problem(90, seed=3)
    F = B.syn.sample_prior_forests(3, 7, bounds, ft, seed=3)
    noise, scale = np.full(3, 0.1), np.ones(3)
    with pytest.raises(ValueError, match="noise"):
        B.fit.batched_mll(F, noise[:2], scale, X, y, ft, include_scale=True, include_2pi=False)
    with pytest.raises(ValueError, match="scale"):
        B.fit.batched_mll(F, noise, scale[:1], X, y, ft, include_scale=True, include_2pi=False)
    with pytest.raises(ValueError, match="rows"):
        B.fit.batched_mll(F, noise, scale, X, y[:-1], ft, include_scale=True, include_2pi=False)
    with pytest.raises(ValueError):
        B.fit.batched_mll(F, noise, scale, X, y[:-1], ft, include_scale=True, include_2pi=False, method="leafspace")
    model, cand = (F, noise, scale), X[:11]
    with pytest.raises(ValueError, match="diagonal"):
        B.tk.forest_predict(model, (X, y), cand, ft, diag=False, method="leafspace")
    with pytest.raises(ValueError, match="unknown method"):
        B.tk.forest_predict(model, (X, y), cand, ft, method="qr")
    with pytest.raises(ValueError, match="features"):
        B.tk.forest_predict(model, (X, y), cand[:, :4], ft)
    many = B.syn.sample_prior_forests(1, 70, bounds, ft, seed=4)  # the leaf-space posterior gathers <= 64 trees
    with pytest.raises(ValueError, match="64 trees"):
        B.tk.forest_predict((many, noise[:1], scale[:1]), (X, y), cand, ft, method="leafspace")
    mu, var = B.tk.forest_predict((many, noise[:1], scale[:1]), (X, y), cand, ft)  # the dense path has no such limit
    mu0, var0 = B.orc.forest_predict((many, noise[:1], scale[:1]), (X, y), cand, ft)
    assert np.allclose(mu, mu0, rtol=1e-9, atol=1e-9) and np.allclose(var, var0, rtol=1e-9, atol=1e-9)
    # sampler state: row counts and the rank limit of a tree swap
    state = B.fit.ChainState.from_forest(F[0], 0.1, 1.0, X, y, ft)
    with pytest.raises(ValueError, match="rows"):
        state.propose_tree(F[0][0], F[1][0], X[:50], ft, 1.0, 7)
    with pytest.raises(ValueError, match="N rows"):
        state.propose(np.ones((50, 2)), np.ones((90, 2)))
    with pytest.raises(RuntimeError):
        state.accept()


def test_tree_swap_with_more_than_64_leaves(B):
    """Tree pairs beyond the 64-column limit of one fused update (the default container allows 50 leaves per tree,
    tree_proposals.py:46-58): (a) structurally > 64 leaves but <= 64 reached -> reached-leaf vectors in one update;
    (b) > 64 reached -> the reference's own subtract-then-add chain (bark_sampler.py:242-255).  Both against a full
    recomputation by the oracle, for ChainState and ChainBatch."""
    rng = np.random.default_rng(0)
    ft = np.full(12, 2)
    bushy = B.syn.full_binary_forest(3, 12, 6, rng, node_limit=127)  # 64 leaves per tree
    for N in (30, 400):  # (a) at most 30 leaves reached per tree; (b) nearly all 64 + 64
        Xc = rng.uniform(size=(N, 12))
        y = rng.standard_normal((N, 1))
        cur = bushy[:2].copy()  # the chain's forest: trees 0, 1; proposal: tree 0 -> tree 2
        st = B.fit.ChainState.from_forest(cur, 0.1, 1.3, Xc, y, ft)
        got = st.propose_tree(cur[0], bushy[2], Xc, ft, 1.3, 2)
        new_forest = np.stack([bushy[2], bushy[1]])
        want = B.orc.batched_mll(new_forest[None], [0.1], [1.3], Xc, y, ft, include_scale=True, include_2pi=False)[0]
        assert np.isclose(got, want, rtol=1e-9, atol=1e-8), (N, got, want)
        st.accept()
        K = 1.3 * B.orc.forest_gram_matrix(new_forest, Xc, Xc, ft) + (1e-6 + 0.1) * np.eye(N)
        assert np.allclose(st.K_inv.cpu().numpy(), np.linalg.inv(K), rtol=1e-7, atol=1e-9)
        assert np.isclose(st.mll, want, rtol=1e-9, atol=1e-8)
        # two chains, one of them with the bushy pair
        cb = B.fit.ChainBatch.from_forests(np.stack([cur, cur]), [0.1, 0.2], [1.3, 0.9], Xc, y, ft)
        vals = cb.propose_trees(np.stack([cur[0], cur[1]]), np.stack([bushy[2], bushy[2]]), Xc, ft, [1.3, 0.9], 2)
        f1 = np.stack([cur[0], bushy[2]])
        want2 = [want, B.orc.batched_mll(f1[None], [0.2], [0.9], Xc, y, ft, include_scale=True, include_2pi=False)[0]]
        assert np.allclose(vals, want2, rtol=1e-9, atol=1e-8)
        cb.accept([True, False])
        assert np.allclose(cb.mll, [want, B.orc.batched_mll(cur[None], [0.2], [0.9], Xc, y, ft, include_scale=True,
                                                            include_2pi=False)[0]], rtol=1e-9, atol=1e-8)
        assert np.allclose(cb.K_inv[0].cpu().numpy(), np.linalg.inv(K), rtol=1e-7, atol=1e-9)


def test_torch_tensors_stay_on_device(B):
    torch = B.torch
    g = load_golden("g3_prior_mixed_n64")
    forest, ft = raw(B, g["forest"]), g["feat_types"]
    Xd = torch.from_numpy(g["X"]).cuda()
    K = B.bf.batched_forest_gram_matrix(forest, Xd, Xd, ft)
    assert K.is_cuda and np.array_equal(K.cpu().numpy(), g["K"])
    from bark_amd.tree_kernels.tree_model_kernel import TreeAgreementKernel

    kern = TreeAgreementKernel(forest[0], ft)
    assert np.array_equal(kern.forward(Xd, Xd).cpu().numpy(), g["K"][0])
    Xc = torch.from_numpy(g["X"])
    assert np.array_equal(kern.forward(Xc, Xc).numpy(), g["K"][0])
    assert torch.equal(kern.forward(Xc, Xc, diag=True), torch.ones(64))
    dg = kern.forward(Xd, Xd, diag=True)  # device inputs: the diagonal of the device Gram, on the device, in its dtype
    full = kern.forward(Xd, Xd)
    assert dg.is_cuda and dg.dtype == full.dtype and torch.equal(dg, full.diagonal())


# ------------------------------------------------------------------ full-size properties ----------
def test_c3_full_size_properties(B):
    """N = 4096, m = 50 (BASELINE configs[2]) — properties that need no CPU reference."""
    N, m, nb = 4096, 50, 6
    X, y, bounds, ft = B.syn.unit_cube_problem(N, 8, seed=4096)
    F = B.syn.sample_prior_forests(nb, m, bounds, ft, seed=4096)
    Xd = B.torch.from_numpy(X).cuda()
    K = B.bf.batched_forest_gram_matrix(F, Xd, Xd, ft)
    assert bool((K == K.transpose(1, 2)).all())                      # symmetric, bit for bit
    assert bool((K.diagonal(dim1=1, dim2=2) == 1.0).all())           # a point always shares its own leaf
    counts = K * m
    assert bool((counts.round() - counts).abs().max() < 1e-9)        # entries are multiples of 1/m
    perm = np.random.default_rng(0).permutation(m)                   # tree order is irrelevant
    assert bool((B.bf.batched_forest_gram_matrix(F[:, perm], Xd, Xd, ft) == K).all())
    # leaf indices of one forest against the oracle, all 4096 x 50 of them
    assert np.array_equal(B.bf.pass_through_forest(F[0], X, ft), B.orc.pass_through_forest(F[0], X, ft))
    noise = np.linspace(0.05, 0.15, nb)
    mll = B.fit.batched_mll(F, noise, None, Xd, y, ft, include_scale=False, include_2pi=True)
    # one forest against the oracle's LU route (the reference's own arithmetic)
    want = B.orc.batched_mll(F[:1], noise[:1], None, X, y, ft, include_scale=False, include_2pi=True)
    assert np.allclose(mll[:1], want, rtol=MLL_RTOL, atol=MLL_ATOL), (mll[0], want)
    # batch-position independence: same forest, same noise -> identical bits wherever it sits
    again = B.fit.batched_mll(F[::-1].copy(), noise[::-1].copy(), None, Xd, y, ft, include_scale=False, include_2pi=True)
    assert np.array_equal(again[::-1], mll)
    # scale/noise identity: MLL(scale=s, noise) with y  ==  MLL(scale=1, (noise+1e-6)/s - 1e-6) with y/sqrt(s) - N/2 log s
    s = 1.7
    lhs = B.fit.batched_mll(F[:2], noise[:2], np.full(2, s), Xd, y, ft, include_scale=True, include_2pi=False)
    rhs = B.fit.batched_mll(F[:2], (noise[:2] + 1e-6) / s - 1e-6, np.ones(2), Xd, y / np.sqrt(s), ft,
                            include_scale=True, include_2pi=False) - 0.5 * N * np.log(s)
    assert np.allclose(lhs, rhs, rtol=1e-9, atol=1e-6)
