"""Pins the CPU oracle (oracle/) against golden vectors produced by the reference itself.

Integer / Gram outputs must be BIT-EXACT; LAPACK-routed values (inv/slogdet) use the
numpy-default closeness the reference's own tests use (tests/bark_fitting/test_quick_inverse.py).
"""
import numpy as np
import pytest

from oracle import oracle as orc

from conftest import load_golden

RAW = orc.nodes_from_raw


def test_node_record_layout():
    # forest.py:8-19: packed offsets 0,1,5,9,13,17,21,25
    offs = [orc.NODE_RECORD_DTYPE.fields[n][1] for n in orc.NODE_RECORD_DTYPE.names]
    assert offs == [0, 1, 5, 9, 13, 17, 21, 25]


def test_g1_kat_tree():
    g = load_golden("g1_kat_tree")
    nodes = RAW(g["nodes"])
    leaves = orc.pass_through_forest(nodes, g["x"], g["feat_types"])
    assert leaves.dtype == np.uint32 and np.array_equal(leaves, g["leaves"])
    assert list(leaves.ravel()) == [3] * 5 + [4] * 5 + [2] * 10  # SURVEY §4 KAT
    K = orc.forest_gram_matrix(nodes, g["x"], g["x"], g["feat_types"])
    assert np.array_equal(K, g["K"])
    assert np.array_equal(orc.get_leaf_vectors(nodes[0], g["x"], g["feat_types"]), g["leaf_vectors"])
    assert np.array_equal(orc.pass_through_forest_py(nodes, g["x"], g["feat_types"]), g["leaves"])


def test_g2_two_tree_kat_woodbury_chain():
    g = load_golden("g2_two_tree_kat")
    forest, new_nodes, x, ft = RAW(g["forest"]), RAW(g["new_nodes"]), g["x"], g["feat_types"]
    K = orc.forest_gram_matrix(forest, x, x, ft)
    assert np.array_equal(K, g["K"])
    Ks = 0.5 * K + 0.1 * np.eye(20)
    _, logdet = np.linalg.slogdet(Ks)
    assert np.isclose(logdet, -36.24970731147612, rtol=1e-13)
    K_inv = np.linalg.inv(Ks)
    s = np.sqrt(0.5 / 2)
    cur = s * orc.get_leaf_vectors(forest[0], x, ft)
    new = s * orc.get_leaf_vectors(new_nodes, x, ft)
    assert np.array_equal(cur, g["cur_leaf_vectors"]) and np.array_equal(new, g["new_leaf_vectors"])
    inv1 = orc.low_rank_inv_update(K_inv, cur, subtract=True)
    det1 = orc.low_rank_det_update(K_inv, cur, logdet, subtract=True)
    inv2 = orc.low_rank_inv_update(inv1, new)
    det2 = orc.low_rank_det_update(inv1, new, det1)
    assert np.allclose(inv1, g["inv_after_subtract"]) and np.isclose(det1, g["det_after_subtract"])
    assert np.allclose(inv2, g["inv_after_add"]) and np.isclose(det2, g["det_after_add"])
    assert np.isclose(det2, g["K_swapped_logdet"]) and np.isclose(inv2, g["K_swapped_inv"]).all()


@pytest.mark.parametrize("N", [64, 257])
def test_g3_prior_mixed(N):
    g = load_golden(f"g3_prior_mixed_n{N}")
    forest, X, y, ft = RAW(g["forest"]), g["X"], g["y"], g["feat_types"]
    for b in range(forest.shape[0]):
        assert np.array_equal(orc.pass_through_forest(forest[b], X, ft), g["leaves"][b])
    assert np.array_equal(orc.batched_forest_gram_matrix(forest, X, X, ft), g["K"])
    assert np.array_equal(orc.batched_forest_gram_matrix_no_null(forest, X, X, ft), g["K_no_null"])
    ex = orc.batched_mll(forest, g["noise"], None, X, y, ft, include_scale=False, include_2pi=True)
    sa = orc.batched_mll(forest, g["noise"], g["scale"], X, y, ft, include_scale=True, include_2pi=False)
    assert np.allclose(ex, g["mll_example"], rtol=1e-12, atol=0)
    assert np.allclose(sa, g["mll_sampler"], rtol=1e-12, atol=0)
    ch = orc.batched_mll(forest, g["noise"], g["scale"], X, y, ft, include_scale=True, include_2pi=False,
                         cholesky=True)
    assert np.allclose(ch, g["mll_sampler"], rtol=1e-9, atol=1e-8)  # the stated GPU tolerance


def test_g3_small_python_walk():
    g = load_golden("g3_prior_mixed_n64")
    forest = RAW(g["forest"])
    assert np.array_equal(orc.pass_through_forest_py(forest[2], g["X"], g["feat_types"]), g["leaves"][2])


def test_g4_all_null():
    g = load_golden("g4_all_null")
    forest, X, ft = RAW(g["forest"]), g["X"], g["feat_types"]
    assert np.array_equal(forest[0], orc.create_empty_forest(forest.shape[1], forest.shape[2]))
    assert np.array_equal(orc.pass_through_forest(forest[0], X, ft), g["leaves"])
    K = orc.batched_forest_gram_matrix(forest, X, X, ft)
    assert np.array_equal(K, g["K"]) and np.all(K == 1.0)
    assert np.array_equal(orc.batched_forest_gram_matrix_no_null(forest, X, X, ft), g["K_no_null"])
    ex = orc.batched_mll(forest, g["noise"], None, X, g["y"], ft, include_scale=False, include_2pi=True)
    assert np.allclose(ex, g["mll_example"], rtol=1e-11)


def test_g5_boundaries():
    g = load_golden("g5_boundaries")
    forest, X, ft = RAW(g["forest"]), g["X"], g["feat_types"]
    assert np.array_equal(orc.pass_through_forest(forest, X, ft), g["leaves"])
    assert np.array_equal(orc.pass_through_forest_py(forest, X, ft), g["leaves"])
    assert np.array_equal(orc.forest_gram_matrix(forest, X, X, ft), g["K"])


def test_g5_categorical_errors():
    g = load_golden("g5_boundaries")
    forest, X, ft = RAW(g["forest"]), g["X"].copy(), g["feat_types"]
    for bad in (-1.0, np.nan, np.inf):
        Xb = X.copy()
        Xb[0, 3] = bad
        with pytest.raises(ValueError):
            orc.pass_through_forest(forest, Xb, ft)
        with pytest.raises((ValueError, OverflowError)):
            orc.pass_through_forest_py(forest, Xb, ft)


def test_g6_predict():
    g = load_golden("g6_predict")
    forest = RAW(g["forest"])
    model = (forest, g["noise"], g["scale"])
    mu, var = orc.forest_predict(model, (g["X"], g["y"]), g["cand"], g["feat_types"], diag=True)
    assert np.allclose(mu, g["mu"], rtol=1e-12, atol=1e-14) and np.allclose(var, g["var"], rtol=1e-12, atol=1e-14)
    _, full = orc.forest_predict(model, (g["X"], g["y"]), g["cand"], g["feat_types"], diag=False)
    assert np.allclose(full, g["var_full"], rtol=1e-12, atol=1e-14)
    K_xX = orc.batched_forest_gram_matrix(forest.reshape(-1, *forest.shape[-2:]), g["cand"], g["X"], g["feat_types"])
    assert np.array_equal(K_xX, g["K_xX"])
    mix_mu, mix_var = orc.mixture_of_gaussians_as_normal(mu, var)
    assert np.allclose(mix_mu, g["mix_mu"]) and np.allclose(mix_var, g["mix_var"])


def test_g7_tree_function_c1():
    g = load_golden("g7_tree_function")
    forest = RAW(g["forest"])
    leaves = orc.pass_through_forest(forest, g["X"], g["feat_types"])
    assert np.array_equal(leaves, g["leaves"])
    y = g["leaf_values"][np.arange(forest.shape[0]), leaves].sum(axis=1)  # tree_function.py:27-31
    assert np.array_equal(y, g["y"])


def test_g8_batched_mll():
    g = load_golden("g8_batched_mll")
    forest, X, y, ft = RAW(g["forest"]), g["X"], g["y"], g["feat_types"]
    assert np.array_equal(orc.batched_forest_gram_matrix(forest, X, X, ft), g["K"])
    ex = orc.batched_mll(forest, g["noise"], None, X, y, ft, include_scale=False, include_2pi=True)
    sa = orc.batched_mll(forest, g["noise"], g["scale"], X, y, ft, include_scale=True, include_2pi=False)
    assert np.allclose(ex, g["mll_example"], rtol=1e-12) and np.allclose(sa, g["mll_sampler"], rtol=1e-12)


def test_g9_woodbury():
    g = load_golden("g9_woodbury")
    for i in range(3):
        A_inv, U, logdet = g[f"Ainv{i}"], g[f"U{i}"], float(g[f"logdet{i}"])
        assert np.allclose(orc.low_rank_inv_update(A_inv, U), g[f"inv_add{i}"])
        assert np.allclose(orc.low_rank_inv_update(A_inv, U, subtract=True), g[f"inv_sub{i}"])
        assert np.isclose(orc.low_rank_det_update(A_inv, U, logdet), g[f"det_add{i}"])
        assert np.isclose(orc.low_rank_det_update(A_inv, U, logdet, subtract=True), g[f"det_sub{i}"])
        # the reference's own assertions (test_quick_inverse.py:29-52)
        A = g[f"A{i}"]
        assert np.isclose(g[f"inv_add{i}"], np.linalg.inv(A + U @ U.T)).all()
        assert np.isclose(g[f"inv_sub{i}"], np.linalg.inv(A - U @ U.T)).all()


def test_g10_forests_from_the_reference_mcmc_sampler():
    """Posterior forests produced by the reference's own sampler (bark_sampler.py:121-284): node containers
    with pruned garbage and reused slots."""
    g = load_golden("g10_mcmc_posterior_forests")
    forest = RAW(g["forest"])  # (chains, samples, m, L)
    flat = forest.reshape(-1, *forest.shape[-2:])
    X, y, ft = g["X"], g["y"], g["feat_types"]
    for b in range(flat.shape[0]):
        assert np.array_equal(orc.pass_through_forest(flat[b], X, ft), g["leaves"][b])
        assert np.array_equal(orc.pass_through_forest_py(flat[b], X, ft), g["leaves"][b])
    assert np.array_equal(orc.batched_forest_gram_matrix(flat, X, X, ft), g["K"])
    assert np.array_equal(orc.batched_forest_gram_matrix_no_null(flat, X, X, ft), g["K_no_null"])
    ex = orc.batched_mll(forest, g["noise"], None, X, y, ft, include_scale=False, include_2pi=True)
    sa = orc.batched_mll(forest, g["noise"], g["scale"], X, y, ft, include_scale=True, include_2pi=False)
    assert np.allclose(ex, g["mll_example"], rtol=1e-12) and np.allclose(sa, g["mll_sampler"], rtol=1e-12)


def test_g11_step_trajectory_of_the_reference_sampler():
    """`_step_bark_sampler` (bark_sampler.py:217-284) recorded proposal by proposal: the oracle's restated Woodbury
    chain (subtract old leaf vectors, add new ones, quick_inverse.mll) reproduces every new_mll, every Metropolis
    decision and the running cur_mll; the noise/scale half by the full LU rebuild, as the reference does."""
    g = load_golden("g11_sampler_steps")
    X, y, ft = g["X"], g["y"], g["feat_types"]
    N = X.shape[0]
    chains, steps, m = g["accept"].shape
    for c in range(chains):
        forest = RAW(g["start_forest"][c]).copy()
        noise, scale = float(g["start_noise"][c]), float(g["start_scale"][c])
        K_s = scale * orc.forest_gram_matrix(forest, X, X, ft) + (1e-6 + noise) * np.eye(N)
        K_inv, logdet = np.linalg.inv(K_s), np.linalg.slogdet(K_s)[1]
        cur = orc.mll(K_inv, logdet, y)
        assert np.isclose(cur, g["start_mll"][c], rtol=1e-10)
        for s in range(steps):
            old, new = RAW(g["old"][c, s]), RAW(g["new"][c, s])
            s_sqrtm = np.sqrt(scale / m)
            for t in range(m):
                assert np.array_equal(old[t], forest[t])  # the proposal replaces the chain's current tree t
                U_old = s_sqrtm * orc.get_leaf_vectors(old[t], X, ft)
                U_new = s_sqrtm * orc.get_leaf_vectors(new[t], X, ft)
                K1 = orc.low_rank_inv_update(K_inv, U_old, subtract=True)
                d1 = orc.low_rank_det_update(K_inv, U_old, logdet, subtract=True)
                K2 = orc.low_rank_inv_update(K1, U_new)
                d2 = orc.low_rank_det_update(K1, U_new, d1)
                new_mll = orc.mll(K2, d2, y)
                assert np.isclose(new_mll, g["new_mll"][c, s, t], rtol=1e-9, atol=1e-9)
                acc = bool(np.log(g["u"][c, s, t]) <= min(g["log_q"][c, s, t] + new_mll - cur, 0))
                assert acc == bool(g["accept"][c, s, t])
                if acc:
                    K_inv, logdet, cur = K2, d2, new_mll
                    forest[t] = new[t]
                assert np.isclose(cur, g["cur_mll"][c, s, t], rtol=1e-9, atol=1e-9)
            nn, nsc = g["ns_prop"][c, s]
            K_s = nsc * orc.forest_gram_matrix(forest, X, X, ft) + (1e-6 + nn) * np.eye(N)
            Kn, dn = np.linalg.inv(K_s), np.linalg.slogdet(K_s)[1]
            ns_mll = orc.mll(Kn, dn, y)
            assert np.isclose(ns_mll, g["ns_new_mll"][c, s], rtol=1e-9, atol=1e-9)
            acc = bool(np.log(g["ns_u"][c, s]) <= min(g["ns_log_q"][c, s] + ns_mll - cur, 0))
            assert acc == bool(g["ns_accept"][c, s])
            if acc:
                K_inv, logdet, cur, noise, scale = Kn, dn, ns_mll, float(nn), float(nsc)
            assert noise == g["noise_after"][c, s] and scale == g["scale_after"][c, s]
            assert np.array_equal(forest, RAW(g["forest_after"][c, s]))
            assert np.isclose(cur, g["mll_after"][c, s], rtol=1e-9, atol=1e-9)
