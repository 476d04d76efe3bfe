#!/usr/bin/env python3
"""Generate the committed golden vectors by RUNNING THE REFERENCE (build container only).

    python tests/golden/make_golden.py

Every expected output in `tests/golden/*.npz` is produced by the reference's own
functions (`bark.forest.*`, `bark.fitting.quick_inverse.*`,
`bark.fitting.bark_prior_sampler._sample_single_forest`, and
`bofire_mixed/benchmarks/tree_function.py`) imported from `/root/reference/src`
through `_ref_shim.py` (identity-njit == the reference's NUMBA_DISABLE_JIT mode),
plus `numpy.linalg.inv/slogdet` exactly where the reference calls them
(`examples/mcmc/mcmc_record_mll.py:57-74`, `bark/tree_kernels/tree_gps.py:80-113`,
`bark/fitting/bark_sampler.py:153-162`).  Since round 4 `forest_predict` /
`mixture_of_gaussians_as_normal` (`bark/tree_kernels/tree_gps.py:80-131`) and the example's
`mll` (`examples/mcmc/mcmc_record_mll.py:57-74`) are the reference's OWN functions too: the
modules import under placeholder `gpytorch` / `beartype` / `bofire` modules (they only need
base classes and type names at import) and `get_feature_types_array` is the reference's
(`bofire_mixed/domain.py:55-65`) on a minimal domain object.  The transcriptions that
rounds 1-3 used (`mll_example`, `predict_lines` below) are kept only as a cross-check that
must agree bit for bit; every fixture's `meta.src` names the imported function.

Forests are stored as raw bytes of the packed 26-byte NODE_RECORD_DTYPE
(`forest.py:8-19`) so the fixtures do not depend on numpy dtype pickling.
"""

import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_shim  # noqa: E402

ref = _ref_shim.reference_modules()
F = ref.forest
QI = ref.quick_inverse
DT = F.NODE_RECORD_DTYPE
assert DT.itemsize == 26

CAT, INT, CONT = 0, 1, 2


def raw(nodes):
    nodes = np.ascontiguousarray(nodes)
    return nodes.view(np.uint8).reshape(*nodes.shape, DT.itemsize).copy()


def save(name, meta, **arrays):
    path = os.path.join(HERE, name + ".npz")
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}  ({os.path.getsize(path)} B)")


def prior_forests(B, m, bounds, feat_types, seed, alpha=0.95, beta=2.0):
    """B forests from the reference prior sampler (bark_prior_sampler.py:15-65)."""
    out = []
    for b in range(B):
        np.random.seed(seed + b)  # sample_splitting_rule uses the global RNG
        rng = np.random.default_rng(seed + b)
        out.append(ref.prior._sample_single_forest(m, bounds, feat_types, alpha, beta, rng))
    return np.array(out)


def mll_example(forest, noise, X, y, feat_types):
    """examples/mcmc/mcmc_record_mll.py:57-74 (no scale, with n*log(2pi))."""
    n = X.shape[0]
    forest = forest.reshape(-1, *forest.shape[-2:])
    noise = noise.reshape(-1)
    K = F.batched_forest_gram_matrix(forest, X, X, feat_types)
    Ks = K + (1e-6 + noise[:, None, None]) * np.eye(n)
    K_inv = np.linalg.inv(Ks)
    _, K_logdet = np.linalg.slogdet(Ks)
    yy = y[None, ...]
    data_fit = (yy.transpose((0, 2, 1)) @ K_inv @ yy).squeeze()
    return 0.5 * (-data_fit - K_logdet - n * np.log(2 * np.pi))


def mll_sampler(forest, noise, scale, X, y, feat_types):
    """bark_sampler.py:153-162 per forest (with scale, quick_inverse.mll)."""
    out = []
    for b in range(forest.shape[0]):
        K = scale[b] * F.forest_gram_matrix(forest[b], X, X, feat_types)
        Ks = K + (1e-6 + noise[b]) * np.eye(K.shape[0])
        K_inv = np.linalg.inv(Ks)
        _, K_logdet = np.linalg.slogdet(Ks)
        out.append(QI.mll(K_inv, K_logdet, y))
    return np.array(out)


def predict_lines(forest, noise, scale, X, y, cand, feat_types):
    """tree_gps.py:87-112 evaluated on the reference's batched Gram."""
    forest = forest.reshape(-1, *forest.shape[-2:])
    noise = noise.reshape(-1)
    scale = scale.reshape(-1)
    S, C = scale.shape[0], cand.shape[0]
    K_XX = scale[:, None, None] * F.batched_forest_gram_matrix(forest, X, X, feat_types)
    K_XX_s = K_XX + (1e-6 + noise[:, None, None]) * np.eye(X.shape[0])
    K_inv = np.linalg.inv(K_XX_s)
    K_xX = scale[:, None, None] * F.batched_forest_gram_matrix(forest, cand, X, feat_types)
    mu = K_xX @ K_inv @ y
    var = scale[:, None, None] - K_xX @ K_inv @ K_xX.transpose((0, 2, 1))
    mu = mu.reshape(S, C)
    return mu, np.diagonal(var, axis1=1, axis2=2).copy(), var


# --------------------------------------------------------------------------------------
def g1_kat_tree():
    # tests/tree_models/test_forest.py:6-20 restated with the 8th field (`active`)
    nodes = np.array(
        [
            (0, 0, 0.5, 1, 2, 0, 0, 1),
            (0, 0, 0.25, 3, 4, 0, 1, 1),
            (1, 0, 1.0, 0, 0, 0, 1, 1),
            (1, 0, 1.0, 0, 0, 1, 2, 1),
            (1, 0, 1.0, 0, 0, 1, 2, 1),
        ],
        dtype=DT,
    ).reshape(1, -1)
    x = np.linspace(0, 1, 20).reshape(-1, 1)
    ft = np.array([CONT])
    leaves = F.pass_through_forest(nodes, x, ft)
    K = F.forest_gram_matrix(nodes, x, x, ft)
    lv = F.get_leaf_vectors(nodes[0], x, ft)
    save("g1_kat_tree", {"src": "tests/tree_models/test_forest.py:6-20 (8-field restatement)"},
         nodes=raw(nodes), x=x, feat_types=ft, leaves=leaves, K=K, leaf_vectors=lv)


def g2_two_tree_kat():
    # tests/bark_fitting/test_quick_inverse.py:55-101 restated with 8 fields
    forest = np.zeros((2, 5), dtype=DT)
    forest[0, 0] = (1, 0, 0, 0, 0, 0, 0, 1)
    forest[1, 0] = (0, 0, 0.5, 1, 2, 0, 0, 1)
    forest[1, 1] = (0, 0, 0.25, 3, 4, 0, 1, 1)
    forest[1, 2] = (1, 0, 0, 0, 0, 0, 1, 1)
    forest[1, 3] = (1, 0, 0, 0, 0, 1, 2, 1)
    forest[1, 4] = (1, 0, 0, 0, 0, 1, 2, 1)
    new_nodes = forest[0].copy()
    new_nodes[0] = (0, 0, 0.75, 1, 2, 0, 0, 1)
    new_nodes[1] = (1, 0, 0, 0, 0, 0, 1, 1)
    new_nodes[2] = (1, 0, 0, 0, 0, 0, 1, 1)

    x = np.linspace(0, 1, 20).reshape(-1, 1)
    ft = np.array([CONT])
    scale, noise = 0.5, 0.1
    K = F.forest_gram_matrix(forest, x, x, ft)
    K_XX_s = scale * K + noise * np.eye(20)
    K_inv = np.linalg.inv(K_XX_s)
    _, K_logdet = np.linalg.slogdet(K_XX_s)

    s_sqrtm = np.sqrt(scale / forest.shape[0])
    cur_lv = s_sqrtm * F.get_leaf_vectors(forest[0], x, ft)
    new_lv = s_sqrtm * F.get_leaf_vectors(new_nodes, x, ft)
    inv1 = QI.low_rank_inv_update(K_inv, cur_lv, subtract=True)
    det1 = QI.low_rank_det_update(K_inv, cur_lv, K_logdet, subtract=True)
    inv2 = QI.low_rank_inv_update(inv1, new_lv)
    det2 = QI.low_rank_det_update(inv1, new_lv, det1)

    forest2 = forest.copy()
    forest2[0] = new_nodes
    K2 = F.forest_gram_matrix(forest2, x, x, ft)
    K2_s = scale * K2 + noise * np.eye(20)
    K2_inv = np.linalg.inv(K2_s)
    _, K2_logdet = np.linalg.slogdet(K2_s)
    assert np.isclose(K2_logdet, det2) and np.isclose(K2_inv, inv2).all()

    save("g2_two_tree_kat",
         {"src": "tests/bark_fitting/test_quick_inverse.py:55-101 (8-field restatement)",
          "scale": scale, "noise": noise},
         forest=raw(forest), new_nodes=raw(new_nodes), x=x, feat_types=ft,
         K=K, K_inv=K_inv, K_logdet=np.float64(K_logdet),
         cur_leaf_vectors=cur_lv, new_leaf_vectors=new_lv,
         inv_after_subtract=inv1, det_after_subtract=np.float64(det1),
         inv_after_add=inv2, det_after_add=np.float64(det2),
         K_swapped=K2, K_swapped_inv=K2_inv, K_swapped_logdet=np.float64(K2_logdet))


def mixed_problem(N, seed, d_cont=6, n_cat=2, n_int=0, cats=5):
    rng = np.random.default_rng(seed)
    cols, bounds, ft = [], [], []
    for _ in range(d_cont):
        cols.append(rng.uniform(size=N))
        bounds.append((0.0, 1.0))
        ft.append(CONT)
    for _ in range(n_int):
        cols.append(rng.integers(0, 11, size=N).astype(np.float64))
        bounds.append((0.0, 10.0))
        ft.append(INT)
    for _ in range(n_cat):
        cols.append(rng.integers(0, cats, size=N).astype(np.float64))
        bounds.append((0.0, float((1 << cats) - 1)))
        ft.append(CAT)
    X = np.stack(cols, axis=1)
    y = rng.standard_normal((N, 1))
    y = (y - y.mean()) / y.std()
    return X, y, np.array(bounds), np.array(ft)


def g3_prior_mixed():
    for N in (64, 257):
        X, y, bounds, ft = mixed_problem(N, seed=N)
        B, m = 3, 50
        forest = prior_forests(B, m, bounds, ft, seed=1000 + N)
        # make one forest contain some null trees and one forest deeper trees (alpha high)
        for attempt in range(50):
            np.random.seed(77 + N + attempt)
            try:
                deep = ref.prior._sample_single_forest(
                    m, bounds, ft, 0.95, 0.9, np.random.default_rng(77 + N + attempt))
                break
            except OverflowError:  # tree_proposals.py:58 — container (L=100) exhausted, redraw
                continue
        forest[2] = deep
        noise = np.array([0.1, 0.05, 0.2])
        scale = np.array([1.0, 0.7, 1.3])
        leaves = np.stack([F.pass_through_forest(forest[b], X, ft) for b in range(B)])
        K = F.batched_forest_gram_matrix(forest, X, X, ft)
        K_nn = F.batched_forest_gram_matrix_no_null(forest, X, X, ft)
        save(f"g3_prior_mixed_n{N}",
             {"src": "bark_prior_sampler.py:15-65 + forest.py:58-111 + mcmc_record_mll.py:57-74 "
                     "+ bark_sampler.py:153-162", "mll_example_src": MLL_EXAMPLE_SRC, "N": N, "m": m, "B": B},
             forest=raw(forest), X=X, y=y, bounds=bounds, feat_types=ft,
             noise=noise, scale=scale, leaves=leaves, K=K, K_no_null=K_nn,
             mll_example=mll_example_ref(forest, noise, X, y, ft),
             mll_sampler=mll_sampler(forest, noise, scale, X, y, ft))


def g4_all_null():
    m = 7
    forest = ref.empty_forest(m)[None]
    X, y, _, ft = mixed_problem(33, seed=4)
    leaves = F.pass_through_forest(forest[0], X, ft)
    K = F.batched_forest_gram_matrix(forest, X, X, ft)
    K_nn = F.batched_forest_gram_matrix_no_null(forest, X, X, ft)
    noise = np.array([0.3])
    save("g4_all_null", {"src": "forest.py:114-117 empty forest; K==1 everywhere", "mll_example_src": MLL_EXAMPLE_SRC},
         forest=raw(forest), X=X, y=y, feat_types=ft, leaves=leaves, K=K, K_no_null=K_nn,
         noise=noise, mll_example=np.atleast_1d(mll_example_ref(forest, noise, X, y, ft)))


def g5_boundaries():
    # one tree per case family; thresholds chosen so float32 rounding matters
    thr32 = np.float32(0.1)  # 0.100000001490116...
    t64 = float(thr32)
    forest = np.zeros((3, 9), dtype=DT)
    # tree 0: continuous split on f0 at float32(0.1), then right child splits f1 at 0.5
    forest[0, 0] = (0, 0, thr32, 5, 7, 0xFFFFFFFF, 0, 1)
    forest[0, 5] = (1, 0, 0, 0, 0, 0, 1, 1)
    forest[0, 7] = (0, 1, 0.5, 2, 8, 0, 1, 1)
    forest[0, 2] = (1, 0, 0, 0, 0, 7, 2, 1)
    forest[0, 8] = (1, 0, 0, 0, 0, 7, 2, 1)
    # tree 1: integer feature f2 split at 3 (x<=3 left), left child splits at 0
    forest[1, 0] = (0, 2, 3.0, 1, 2, 0xFFFFFFFF, 0, 1)
    forest[1, 1] = (0, 2, 0.0, 3, 4, 0, 1, 1)
    forest[1, 2] = (1, 0, 0, 0, 0, 0, 1, 1)
    forest[1, 3] = (1, 0, 0, 0, 0, 1, 2, 1)
    forest[1, 4] = (1, 0, 0, 0, 0, 1, 2, 1)
    # tree 2: categorical f3, mask 0b0110 ; right child: mask 0b1000 on same feature
    forest[2, 0] = (0, 3, 6.0, 1, 2, 0xFFFFFFFF, 0, 1)
    forest[2, 1] = (1, 0, 0, 0, 0, 0, 1, 1)
    forest[2, 2] = (0, 3, 8.0, 3, 4, 0, 1, 1)
    forest[2, 3] = (1, 0, 0, 0, 0, 2, 2, 1)
    forest[2, 4] = (1, 0, 0, 0, 0, 2, 2, 1)
    ft = np.array([CONT, CONT, INT, CAT])
    f0 = [t64, np.nextafter(t64, 1.0), np.nextafter(t64, 0.0), 0.1, 0.0, -0.0, 1.0,
          np.nan, np.inf, -np.inf, 0.09999999, 0.10000001]
    rows = []
    for i, a in enumerate(f0):
        rows.append([a, [0.5, np.nextafter(0.5, 1), np.nan, 0.25][i % 4],
                     float([0, 3, 4, -1, 10, 2][i % 6]), float(i % 5)])
    # categorical values 0..4 plus a large (out of mask) category 30 and fractional 2.9 (truncates to 2)
    rows.append([0.5, 0.5, 3.0, 30.0])
    rows.append([0.5, 0.5, 3.0, 2.9])
    rows.append([0.5, 0.5, 3.5, 1.0])
    X = np.array(rows, dtype=np.float64)
    leaves = F.pass_through_forest(forest, X, ft)
    K = F.forest_gram_matrix(forest, X, X, ft)
    save("g5_boundaries", {"src": "forest.py:28-47 edge semantics: x==thr(f32), nextafter, NaN, inf, "
                                  "-0.0, int split, categorical bitmask, truncation toward zero"},
         forest=raw(forest), X=X, feat_types=ft, leaves=leaves, K=K)


def g6_predict():
    N, C = 64, 33
    X, y, bounds, ft = mixed_problem(N, seed=66)
    cand, _, _, _ = mixed_problem(C, seed=67)
    B, m = 4, 50
    forest = prior_forests(B, m, bounds, ft, seed=660)
    noise = np.array([0.1, 0.07, 0.15, 0.12])
    scale = np.array([1.0, 0.8, 1.2, 0.95])
    # shape (chains=2, samples=2, ...) to exercise the flatten at tree_gps.py:88-90
    tg, _, make_domain = reference_predict_modules()
    model = tg.BARKModel(forest.reshape(2, 2, m, -1), noise.reshape(2, 2), scale.reshape(2, 2))
    dom = make_domain(ft)
    mu, var_diag = tg.forest_predict(model, (X, y), cand, dom, diag=True)
    mu_f, var_full = tg.forest_predict(model, (X, y), cand, dom, diag=False)
    mu_y, var_y = tg.mixture_of_gaussians_as_normal(mu, var_diag)
    # cross-check: the transcription of rounds 1-3 must be the same numbers, bit for bit
    t_mu, t_diag, t_full = predict_lines(forest, noise, scale, X, y, cand, ft)
    assert np.array_equal(mu, t_mu) and np.array_equal(mu_f, t_mu) and np.array_equal(var_diag, t_diag) and np.array_equal(var_full, t_full)
    assert np.array_equal(mu_y, np.mean(mu, axis=0)) and np.array_equal(var_y, np.mean(var_diag + mu**2, axis=0) - mu_y**2)
    K_xX = F.batched_forest_gram_matrix(forest, cand, X, ft)
    save("g6_predict",
         {"src": "bark.tree_kernels.tree_gps.forest_predict (tree_gps.py:80-113; diag=True -> mu, var; diag=False -> var_full) "
                 "and mixture_of_gaussians_as_normal (tree_gps.py:116-131), imported from /root/reference/src under placeholder "
                 "gpytorch / beartype / bofire modules; feat_types from bofire_mixed.domain.get_feature_types_array",
          "N": N, "C": C, "B": B},
         forest=raw(forest.reshape(2, 2, m, -1)), noise=noise.reshape(2, 2), scale=scale.reshape(2, 2),
         X=X, y=y, cand=cand, feat_types=ft, K_xX=K_xX, mu=mu, var=np.ascontiguousarray(var_diag), var_full=var_full,
         mix_mu=mu_y, mix_var=var_y)


def g7_tree_function():
    """Config c1: TreeFunction(dim=5, m=50, function_seed=1) at 64 seeded points.

    `tree_function.py` imports bofire/matplotlib at module top; they are absent, so empty
    placeholder modules satisfy the imports and the two *functions* that define the
    benchmark (`sample_tree_structure_from_prior` :36-57, `sample_tree_function_from_structure`
    :19-33) are called with a minimal domain object (5 continuous inputs), exactly what
    `TreeFunction.__init__` (:65-88) passes them for the default arguments.
    """
    class _Cont:  # placeholder for bofire ContinuousInput
        pass

    class _Cat:
        pass

    class _Disc:
        pass

    def _mod(name, **attrs):
        mod = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(mod, k, v)
        sys.modules[name] = mod
        return mod

    obj = type("_P", (), {})
    _mod("bofire")
    _mod("bofire.benchmarks")
    _mod("bofire.benchmarks.api", Benchmark=object)
    _mod("bofire.data_models")
    _mod("bofire.data_models.domain")
    _mod("bofire.data_models.domain.api", Domain=obj, Inputs=obj, Outputs=obj, Features=obj)
    _mod("bofire.data_models.enum", CategoricalEncodingEnum=obj)
    _mod("bofire.data_models.features")
    _mod("bofire.data_models.features.api", CategoricalInput=_Cat, ContinuousInput=_Cont,
         ContinuousOutput=obj, DiscreteInput=_Disc, AnyFeature=obj)
    _mod("bofire.data_models.objectives")
    _mod("bofire.data_models.objectives.api", MinimizeObjective=obj)
    if "matplotlib" not in sys.modules:
        try:
            import matplotlib.pyplot  # noqa: F401
        except Exception:
            _mod("matplotlib")
            _mod("matplotlib.pyplot")

    # bofire_mixed/__init__ imports the whole plugin; load the two files by path instead
    import importlib.util

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod

    _mod("bofire_mixed")
    load("bofire_mixed.domain", "/root/reference/src/bofire_mixed/domain.py")
    tf = load("ref_tree_function", "/root/reference/src/bofire_mixed/benchmarks/tree_function.py")
    tf.create_empty_forest = ref.empty_forest  # numpy-2-safe ctor, same record

    class _Inputs(list):
        def get(self):
            return list(self)

    dim, m, seed = 5, 50, 1
    domain = types.SimpleNamespace(inputs=_Inputs(_Cont() for _ in range(dim)))
    rng = np.random.default_rng(seed)  # tree_function.py:86
    forest = tf.sample_tree_structure_from_prior(m, domain, rng)
    # sample_tree_function_from_structure draws leaf_values from the same rng (:23)
    state_before = rng.bit_generator.state
    f = tf.sample_tree_function_from_structure(forest, domain, rng)
    leaf_values = np.random.default_rng(0)
    leaf_values.bit_generator.state = state_before
    leaf_values = leaf_values.standard_normal(forest.shape)
    X = np.random.default_rng(64).uniform(size=(64, dim))
    yv = f(X)
    ft = np.full(dim, CONT)
    leaves = F.pass_through_forest(forest, X, ft)
    assert np.allclose(leaf_values[np.arange(m), leaves].sum(axis=1), yv)
    save("g7_tree_function",
         {"src": "bofire_mixed/benchmarks/tree_function.py:19-57,65-88 defaults dim=5 m=50 seed=1; "
                 "X=default_rng(64).uniform((64,5))"},
         forest=raw(forest), leaf_values=leaf_values, X=X, y=yv, feat_types=ft, leaves=leaves)


def g8_batched_mll():
    N = 96
    X, y, bounds, ft = mixed_problem(N, seed=88, d_cont=5, n_cat=1, n_int=2)
    B, m = 4, 50
    forest = prior_forests(B, m, bounds, ft, seed=880)
    noise = np.array([0.05, 0.1, 0.15, 0.08])
    scale = np.array([1.0, 1.1, 0.9, 1.05])
    K = F.batched_forest_gram_matrix(forest, X, X, ft)
    save("g8_batched_mll",
         {"src": "mcmc_record_mll.py:57-74 and bark_sampler.py:153-162; d = 5 cont + 2 int + 1 cat", "mll_example_src": MLL_EXAMPLE_SRC,
          "N": N, "B": B},
         forest=raw(forest), X=X, y=y, bounds=bounds, feat_types=ft, noise=noise, scale=scale, K=K,
         mll_example=mll_example_ref(forest, noise, X, y, ft),
         mll_sampler=mll_sampler(forest, noise, scale, X, y, ft))


def g9_woodbury():
    # tests/bark_fitting/test_quick_inverse.py:13-52 inputs, outputs from the reference functions
    out = {}
    for i, (N, B, seed) in enumerate([(5, 2, 42), (4, 2, 43), (6, 3, 44)]):
        rng = np.random.default_rng(seed)
        A = rng.standard_normal((N, N))
        U = rng.standard_normal((N, B)) * 0.1
        _, logdet = np.linalg.slogdet(A)
        A_inv = np.linalg.inv(A)
        out[f"A{i}"], out[f"U{i}"], out[f"Ainv{i}"], out[f"logdet{i}"] = A, U, A_inv, np.float64(logdet)
        out[f"inv_add{i}"] = QI.low_rank_inv_update(A_inv, U, subtract=False)
        out[f"inv_sub{i}"] = QI.low_rank_inv_update(A_inv, U, subtract=True)
        out[f"det_add{i}"] = np.float64(QI.low_rank_det_update(A_inv, U, logdet))
        out[f"det_sub{i}"] = np.float64(QI.low_rank_det_update(A_inv, U, logdet, subtract=True))
    save("g9_woodbury", {"src": "tests/bark_fitting/test_quick_inverse.py:13-52; quick_inverse.py:13-38"}, **out)


def _bofire_stubs():
    """Empty placeholder modules so that modules importing bofire at top level can be loaded."""
    class _Cont:
        pass

    class _Cat:
        pass

    class _Disc:
        pass

    def _mod(name, **attrs):
        mod = sys.modules.get(name) or types.ModuleType(name)
        for k, v in attrs.items():
            setattr(mod, k, v)
        sys.modules[name] = mod
        return mod

    obj = type("_P", (), {})
    _mod("bofire")
    _mod("bofire.data_models")
    _mod("bofire.data_models.domain")
    _mod("bofire.data_models.domain.api", Domain=obj, Inputs=obj, Outputs=obj, Features=obj)
    _mod("bofire.data_models.features")
    _mod("bofire.data_models.features.api", CategoricalInput=_Cat, ContinuousInput=_Cont, ContinuousOutput=obj,
         DiscreteInput=_Disc, AnyFeature=obj)
    if "bofire_mixed.domain" not in sys.modules:
        import importlib.util

        _mod("bofire_mixed")
        spec = importlib.util.spec_from_file_location("bofire_mixed.domain", "/root/reference/src/bofire_mixed/domain.py")
        mod = importlib.util.module_from_spec(spec)
        sys.modules["bofire_mixed.domain"] = mod
        spec.loader.exec_module(mod)


_REF_PREDICT = None


def reference_predict_modules():
    """-> (bark.tree_kernels.tree_gps, examples/mcmc/mcmc_record_mll.py as a module, make_domain).  The reference's modules,
    imported unmodified: `tree_gps.py` needs gpytorch only as the base class of LeafGP / TreeAgreementKernel (never
    instantiated here), beartype for `Optional`, bofire for the `Domain` annotation; the example script needs a handful of
    bofire / bofire_mixed names at import that its `mll` function never touches.  `make_domain(feat_types)` builds the
    object `get_feature_types_array` (the reference's own, bofire_mixed/domain.py:55-65) is applied to."""
    global _REF_PREDICT
    if _REF_PREDICT is not None:
        return _REF_PREDICT
    import importlib.util
    import typing

    _bofire_stubs()

    def _mod(name, **attrs):
        mod = sys.modules.get(name) or types.ModuleType(name)
        for k, v in attrs.items():
            setattr(mod, k, v)
        sys.modules[name] = mod
        return mod

    base = type("_Base", (), {"__init__": lambda self, *a, **k: None})
    _mod("gpytorch", models=types.SimpleNamespace(ExactGP=base), kernels=types.SimpleNamespace(Kernel=base, ScaleKernel=base),
         means=types.SimpleNamespace(ZeroMean=base, ConstantMean=base), distributions=types.SimpleNamespace(MultivariateNormal=base),
         likelihoods=types.SimpleNamespace(GaussianLikelihood=base))
    _mod("beartype")
    _mod("beartype.typing", Optional=typing.Optional)
    import bark.tree_kernels.tree_gps as tg

    obj = type("_P", (), {})
    _mod("bofire.data_models.strategies")
    _mod("bofire.data_models.strategies.api", RandomStrategy=obj)
    _mod("bofire_mixed.benchmarks", DatasetBenchmark=obj, map_benchmark=lambda *a, **k: None)
    _mod("bofire_mixed.data_models")
    _mod("bofire_mixed.data_models.strategies")
    _mod("bofire_mixed.data_models.strategies.mapper", strategy_map=lambda *a, **k: None)
    _mod("bofire_mixed.data_models.surrogates")
    _mod("bofire_mixed.data_models.surrogates.api", BARKSurrogate=obj)
    _mod("bofire_mixed.data_models.surrogates.mapper", surrogate_map=lambda *a, **k: None)
    import bark.fitting.bark_sampler  # noqa: F401  (DataT / ModelT of the example's annotations)

    spec = importlib.util.spec_from_file_location("ref_mcmc_record_mll", "/root/reference/examples/mcmc/mcmc_record_mll.py")
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)  # defines functions only; its __main__ block does not run

    # the feature classes the reference's get_feature_types_array tests with isinstance — those of the bofire_mixed.domain
    # module object tree_gps and the example imported (g7 loads that file a second time with classes of its own)
    assert ex.get_feature_types_array is tg.get_feature_types_array
    gl = tg.get_feature_types_array.__globals__
    by_type = {CAT: gl["CategoricalInput"], INT: gl["DiscreteInput"], CONT: gl["ContinuousInput"]}

    def make_domain(feat_types):
        inputs = types.SimpleNamespace(get=lambda: [by_type[int(t)]() for t in feat_types])
        dom = types.SimpleNamespace(inputs=inputs)
        assert np.array_equal(tg.get_feature_types_array(dom), feat_types)
        return dom

    _REF_PREDICT = (tg, ex, make_domain)
    return _REF_PREDICT


MLL_EXAMPLE_SRC = ("mll() of /root/reference/examples/mcmc/mcmc_record_mll.py:57-74, the module loaded by path under placeholder "
                   "bofire modules; feat_types through bofire_mixed.domain.get_feature_types_array")


def mll_example_ref(forest, noise, X, y, feat_types):
    """examples/mcmc/mcmc_record_mll.py:57-74 — the reference's function itself; the transcription must agree bit for bit."""
    _, ex, make_domain = reference_predict_modules()
    scale = np.ones_like(np.asarray(noise, dtype=np.float64))  # the example's mll unpacks but never uses `scale`
    out = ex.mll((forest, noise, scale), (X, y), make_domain(feat_types))
    assert np.array_equal(np.atleast_1d(out), np.atleast_1d(mll_example(forest, noise, X, y, feat_types)))
    return out


def g10_mcmc_posterior_forests():
    """Forests produced by the reference's own Metropolis-Hastings sampler
    (bark_sampler.py:121-284 `_run_bark_sampler_multichain`): 2 chains x 3 samples after warm-up, on a mixed
    cont/int/cat domain.  Their node containers carry the sampler's history (pruned sub-trees left behind as
    inactive garbage, slots reused by later grows, `change`d rules) — the realistic input of the drop-in.
    Expected outputs from the reference's forest.py / mcmc_record_mll arithmetic."""
    _bofire_stubs()
    import bark.fitting.bark_sampler as S

    N, m = 40, 10
    X, y, bounds, ft = mixed_problem(N, seed=1010, d_cont=3, n_cat=1, n_int=1, cats=4)
    chains = 2
    forest0 = np.stack([ref.empty_forest(m) for _ in range(chains)])
    noise0 = np.array([0.1, 0.2])
    scale0 = np.array([1.0, 1.0])
    w = np.array([0.5, 0.5, 1.0])
    params = S.BARKTrainParamsNumba(
        warmup_steps=40, num_samples=3, steps_per_sample=6, num_chains=chains, alpha=0.95, beta=2.0,
        proposal_weights=w / w.sum(), verbose=False, use_softplus_transform=True, sample_scale=False,
        gamma_prior_shape=1.5, gamma_prior_rate=5.0)
    np.random.seed(1010)
    node_samples, noise_samples, scale_samples = S._run_bark_sampler_multichain(
        forest0, noise0, scale0, X, y, bounds, ft, params)
    flat = node_samples.reshape(-1, m, node_samples.shape[-1])
    leaves = np.stack([F.pass_through_forest(f, X, ft) for f in flat])
    K = F.batched_forest_gram_matrix(flat, X, X, ft)
    K_nn = F.batched_forest_gram_matrix_no_null(flat, X, X, ft)
    inactive_internal = int(((flat["active"] == 0) & (flat["is_leaf"] == 0) & (flat["left"] > 0)).sum())
    save("g10_mcmc_posterior_forests",
         {"mll_example_src": MLL_EXAMPLE_SRC, "src": "bark_sampler.py:121-284 run under the shim (np.random.seed(1010)); forest.py:58-111; "
                 "mcmc_record_mll.py:57-74; bark_sampler.py:153-162",
          "inactive_internal_slots": inactive_internal, "active_nodes_max": int(flat["active"].sum(-1).max())},
         forest=raw(node_samples), noise=noise_samples, scale=scale_samples, X=X, y=y, bounds=bounds, feat_types=ft,
         leaves=leaves, K=K, K_no_null=K_nn,
         mll_example=mll_example_ref(node_samples, noise_samples, X, y, ft),
         mll_sampler=mll_sampler(flat, noise_samples.reshape(-1), scale_samples.reshape(-1), X, y, ft))


def g11_sampler_steps():
    """Step-level trajectory of the reference's OWN Metropolis-Hastings step (`_step_bark_sampler`,
    bark_sampler.py:217-284), run unmodified under the shim with recording hooks on the names it calls
    (`get_tree_proposal`, `get_noise_scale_proposal`, `mll`, `np.random.uniform()`): for two independent chains, after a
    warm-up, every per-tree proposal of three consecutive steps — the tree it replaces, the proposed tree, log_q_prior,
    the uniform draw, new_mll, the decision, cur_mll after — and the noise/scale half of each step.  The GPU tests
    replay it through ChainState.propose_tree / accept and ChainBatch.sweep_trees."""
    _bofire_stubs()
    import bark.fitting.bark_sampler as S

    N, m, L, chains, warm, steps = 48, 8, 100, 2, 25, 3
    X, y, bounds, ft = mixed_problem(N, seed=1111, d_cont=3, n_cat=1, n_int=1, cats=4)
    w = np.array([0.5, 0.5, 1.0])
    params = S.BARKTrainParamsNumba(
        warmup_steps=0, num_samples=1, steps_per_sample=1, num_chains=1, alpha=0.95, beta=2.0,
        proposal_weights=w / w.sum(), verbose=False, use_softplus_transform=True, sample_scale=True,
        gamma_prior_shape=1.5, gamma_prior_rate=5.0)

    log = {"prop": [], "mll": [], "u": [], "ns": []}
    real_np, real_prop, real_ns, real_mll = S.np, S.get_tree_proposal, S.get_noise_scale_proposal, S.mll

    class _Random:
        def __getattr__(self, name):
            return getattr(real_np.random, name)

        def uniform(self, *a, **k):
            v = real_np.random.uniform(*a, **k)
            if not a and not k:  # the accept draws of bark_sampler.py:258,276 (proposals draw in their own modules)
                log["u"].append(float(v))
            return v

    class _Np:
        random = _Random()

        def __getattr__(self, name):
            return getattr(real_np, name)

    def prop(nodes, *a):
        old = np.array(nodes, copy=True)
        new_nodes, log_q = real_prop(nodes, *a)
        log["prop"].append((old, np.array(new_nodes, copy=True), float(log_q)))
        return new_nodes, log_q

    def ns(noise, scale, prm):
        (nn, nsc), lq = real_ns(noise, scale, prm)
        log["ns"].append((float(nn), float(nsc), float(lq)))
        return (nn, nsc), lq

    def mll_rec(*a):
        v = real_mll(*a)
        log["mll"].append(float(v))
        return v

    rec = {k: [] for k in ("old", "new", "log_q", "u", "new_mll", "accept", "cur_mll", "ns_prop", "ns_log_q", "ns_u",
                           "ns_new_mll", "ns_accept", "noise_after", "scale_after", "forest_after", "mll_after")}
    start = {"forest": [], "noise": [], "scale": [], "mll": []}
    S.np, S.get_tree_proposal, S.get_noise_scale_proposal, S.mll = _Np(), prop, ns, mll_rec
    try:
        for c in range(chains):
            np.random.seed(1111 + c)
            forest = ref.empty_forest(m, L)
            noise, scale = (0.1, 1.0) if c == 0 else (0.25, 0.8)
            K_s = scale * F.forest_gram_matrix(forest, X, X, ft) + (1e-6 + noise) * np.eye(N)
            K_inv, logdet = np.linalg.inv(K_s), np.linalg.slogdet(K_s)[1]
            cur = QI.mll(K_inv, logdet, y)
            for _ in range(warm):
                forest, noise, scale, K_inv, logdet, cur = S._step_bark_sampler(forest, noise, scale, X, y, bounds, ft, params,
                                                                                K_inv, logdet, cur)
            start["forest"].append(np.array(forest, copy=True)); start["noise"].append(noise); start["scale"].append(scale)
            start["mll"].append(float(cur))
            per = {k: [] for k in rec}
            for _ in range(steps):
                for k in log:
                    log[k].clear()
                before = float(cur)
                forest, noise, scale, K_inv, logdet, cur = S._step_bark_sampler(forest, noise, scale, X, y, bounds, ft, params,
                                                                                K_inv, logdet, cur)
                assert len(log["prop"]) == m and len(log["mll"]) == m + 1 and len(log["u"]) == m + 1 and len(log["ns"]) == 1
                run, acc, curs = before, [], []
                for t in range(m):  # bark_sampler.py:256-264
                    a = bool(np.log(log["u"][t]) <= min(log["prop"][t][2] + log["mll"][t] - run, 0))
                    if a:
                        run = log["mll"][t]
                    acc.append(a)
                    curs.append(run)
                a_ns = bool(np.log(log["u"][m]) <= min(log["ns"][0][2] + log["mll"][m] - run, 0))  # :272-276
                if a_ns:
                    run = log["mll"][m]
                assert run == float(cur), (run, cur)  # the derived decisions reproduce the step's own outcome
                per["old"].append(raw(np.stack([q[0] for q in log["prop"]])))
                per["new"].append(raw(np.stack([q[1] for q in log["prop"]])))
                per["log_q"].append([q[2] for q in log["prop"]]); per["u"].append(log["u"][:m])
                per["new_mll"].append(log["mll"][:m]); per["accept"].append(acc); per["cur_mll"].append(curs)
                per["ns_prop"].append(log["ns"][0][:2]); per["ns_log_q"].append(log["ns"][0][2]); per["ns_u"].append(log["u"][m])
                per["ns_new_mll"].append(log["mll"][m]); per["ns_accept"].append(a_ns)
                per["noise_after"].append(noise); per["scale_after"].append(scale)
                per["forest_after"].append(raw(np.array(forest, copy=True))); per["mll_after"].append(float(cur))
            for k in rec:
                rec[k].append(per[k])
    finally:
        S.np, S.get_tree_proposal, S.get_noise_scale_proposal, S.mll = real_np, real_prop, real_ns, real_mll
    arrays = {k: np.array(v) for k, v in rec.items()}
    n_acc = int(arrays["accept"].sum())
    assert 0 < n_acc < arrays["accept"].size, "the trajectory should hold both accepted and rejected proposals"
    save("g11_sampler_steps",
         {"src": "bark_sampler.py:217-284 `_step_bark_sampler` run unmodified under the shim (np.random.seed(1111 + chain)), "
                 "with recording wrappers on get_tree_proposal / get_noise_scale_proposal / mll / np.random.uniform()",
          "layout": "(chains, steps, trees, ...) for the per-tree arrays, (chains, steps, ...) for the noise/scale half",
          "tree_accepts": n_acc, "tree_proposals": int(arrays["accept"].size), "ns_accepts": int(arrays["ns_accept"].sum())},
         X=X, y=y, bounds=bounds, feat_types=ft, start_forest=raw(np.stack(start["forest"])),
         start_noise=np.array(start["noise"]), start_scale=np.array(start["scale"]), start_mll=np.array(start["mll"]),
         **arrays)


if __name__ == "__main__":
    g1_kat_tree()
    g2_two_tree_kat()
    g3_prior_mixed()
    g4_all_null()
    g5_boundaries()
    g6_predict()
    g7_tree_function()
    g8_batched_mll()
    g9_woodbury()
    g10_mcmc_posterior_forests()
    g11_sampler_steps()
    # the reference tree must stay clean
    import subprocess

    dirty = subprocess.run(["find", "/root/reference", "-name", "__pycache__"], capture_output=True, text=True).stdout
    assert dirty.strip() == "", dirty
