"""Boundary behaviour of the C ABI on the GPU: per-thread contexts (two host threads on two streams), hipGraph
capture of the sweep, the device-side Metropolis sweep of the sampler step, and the error paths the reference has
(singular low-rank systems, invalid categorical values met by a walk)."""
import ctypes
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch

    assert torch.cuda.is_available()
    import bark_amd.fitting as fit
    import bark_amd.forest as bf
    from bark_amd import _lib, synthetic
    from oracle import oracle as orc

    class NS:
        pass

    ns = NS()
    ns.torch, ns.fit, ns.bf, ns.lib, ns.syn, ns.orc = torch, fit, bf, _lib, synthetic, orc
    return ns


def test_context_lifecycle_and_argument_checks(env):
    L, lib = env.lib, env.lib.lib()
    h = ctypes.c_void_p()
    L.check(lib.bark_ctx_create(0, ctypes.byref(h)))
    p1, p2 = ctypes.c_void_p(), ctypes.c_void_p()
    L.check(lib.bark_ctx_workspace(h, 1 << 20, ctypes.byref(p1)))
    L.check(lib.bark_ctx_workspace(h, 1 << 16, ctypes.byref(p2)))  # smaller request: same buffer
    assert p1.value == p2.value and p1.value % 256 == 0 and lib.bark_ctx_workspace_bytes(h) == 1 << 20
    L.check(lib.bark_ctx_workspace(h, 1 << 22, ctypes.byref(p2)))
    assert lib.bark_ctx_workspace_bytes(h) == 1 << 22
    flag = ctypes.c_int32(7)
    L.check(lib.bark_ctx_status(h, None, ctypes.byref(flag)))
    assert flag.value == 0
    lib.bark_ctx_destroy(h)
    assert lib.bark_ctx_create(99, ctypes.byref(h)) != 0 and b"device" in lib.bark_last_error()
    # an entry point without a context refuses to run
    X = env.torch.zeros((4, 2), dtype=env.torch.float64, device="cuda")
    out = env.torch.zeros((1, 4, 1), dtype=env.torch.int32, device="cuda")
    info = L.PackInfo()
    rc = lib.bark_leaf_indices_hip(None, L.ptr(X), ctypes.byref(info), L.ptr(X), 4, 2, L.ptr(out), None)
    assert rc == L.BARK_ERR_ARG and b"bark_ctx" in lib.bark_last_error()


def test_two_host_threads_on_two_streams(env):
    """Each thread has its own context (scratch, helper stream, events) and its own stream: results are bit-identical
    to the serial ones however the launches interleave."""
    torch, fit, syn = env.torch, env.fit, env.syn
    X, y, bounds, ft = syn.mixed_problem(900, seed=11)
    jobs = []
    for k in range(2):
        F = syn.sample_prior_forests(24, 50, bounds, ft, seed=100 + k)
        noise = np.linspace(0.05, 0.2, 24) + 0.01 * k
        jobs.append((F, noise))
    serial = [fit.batched_mll(F, n, None, X, y, ft, include_scale=False, include_2pi=True) for F, n in jobs]
    results, errors = [None, None], []

    def work(k):
        try:
            torch.cuda.set_device(0)
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                F, n = jobs[k]
                outs = [fit.batched_mll(F, n, None, X, y, ft, include_scale=False, include_2pi=True) for _ in range(6)]
                s.synchronize()
            results[k] = outs
            env.lib.release_ctx()
        except Exception as exc:  # pragma: no cover
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    for k in range(2):
        for out in results[k]:
            assert np.array_equal(out, serial[k])


def test_device_side_wait_and_event_joins_give_identical_bits(env):
    """Sweeps of few matrices hand row-launch completion to the caller's stream through a device-side counter that
    diag_kernel waits for (include/bark_hip.h, bark_device_wait); switched off, the same kernels are ordered by events.
    Same arithmetic either way: identical bits — on the split-K layout (lone matrix, look-ahead) and on the pipelined
    schedule (gate kernels on the row streams) — and no timed-out wait (info == 0)."""
    import bench

    lib = env.lib.lib()
    for N, Bn in ((1500, 1), (6900, 1), (2100, 6), (1100, 24)):
        wl = bench.Workload(N, 8, 50, Bn, seed_base=N, rank_offset=0)
        assert lib.bark_device_wait(1) in (0, 1)
        wl.run()
        env.torch.cuda.synchronize()
        on = wl.mll_d.clone()
        assert int(wl.info_d.abs().max().item()) == 0
        assert lib.bark_device_wait(0) == 1
        wl.run()
        env.torch.cuda.synchronize()
        assert lib.bark_device_wait(1) == 0
        assert bool((wl.mll_d == on).all()) and int(wl.info_d.abs().max().item()) == 0


def test_sweep_is_hipgraph_capturable(env):
    """The ABI's claim: `*_hip` entry points only enqueue work (fork/join of the helper stream included), so a call can
    be captured into a graph and replayed — what a latency-bound caller (one small matrix) wants."""
    import bench

    torch = env.torch
    wl = bench.Workload(1024, 8, 50, 1, seed_base=1024, rank_offset=0)
    wl.run()  # creates the context's helper stream / events and sets the LDS attributes outside the capture
    torch.cuda.synchronize()
    eager = wl.mll_d.clone()
    g = torch.cuda.CUDAGraph()
    wl.mll_d.zero_()
    with torch.cuda.graph(g):
        wl.stream = env.lib.stream_ptr()  # the capture stream
        wl.run()
    wl.mll_d.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert bool((wl.mll_d == eager).all()) and int(wl.info_d.abs().max().item()) == 0
    # a larger batch through the non-split path as well
    wl2 = bench.Workload(700, 8, 50, 64, seed_base=700, rank_offset=0)
    wl2.run()
    torch.cuda.synchronize()
    eager2 = wl2.mll_d.clone()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        wl2.stream = env.lib.stream_ptr()
        wl2.run()
    wl2.mll_d.zero_()
    g2.replay()
    torch.cuda.synchronize()
    assert bool((wl2.mll_d == eager2).all())
    # matrices of one block row: leaf walk + one launch on the caller's stream
    wl4 = bench.Workload(64, 8, 50, 32, seed_base=64, rank_offset=0)
    wl4.run()
    torch.cuda.synchronize()
    eager4 = wl4.mll_d.clone()
    g4 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g4):
        wl4.stream = env.lib.stream_ptr()
        wl4.run()
    wl4.mll_d.zero_()
    g4.replay()
    torch.cuda.synchronize()
    assert bool((wl4.mll_d == eager4).all()) and int(wl4.info_d.abs().max().item()) == 0
    # and the split-K look-ahead schedule (third stream: the bulk of step j+2 forks after solve(j) and joins two steps later)
    wl3 = bench.Workload(6900, 8, 50, 1, seed_base=6900, rank_offset=0)  # 54 block rows: still the split-K layout, look-ahead in the middle steps
    wl3.run()
    torch.cuda.synchronize()
    eager3 = wl3.mll_d.clone()
    g3 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g3):
        wl3.stream = env.lib.stream_ptr()
        wl3.run()
    wl3.mll_d.zero_()
    g3.replay()
    g3.replay()
    torch.cuda.synchronize()
    assert bool((wl3.mll_d == eager3).all()) and int(wl3.info_d.abs().max().item()) == 0
    want = env.orc.batched_mll(wl3.forests[:2], wl3.noise[:2], None, wl3.X, wl3.y, wl3.ft, include_scale=False, include_2pi=True)
    assert np.allclose(eager3[:2].cpu().numpy(), want, rtol=1e-9, atol=1e-8)


def test_thread_exit_frees_its_context(env):
    """A host thread that used the library and exits WITHOUT release_ctx() must not leak its bark_ctx (helper streams,
    events, pinned page, grow-only hipMalloc scratch): the per-thread owner object is collected with the thread's
    locals and its finalizer destroys the context."""
    import gc
    import weakref

    torch, L = env.torch, env.lib
    torch.cuda.synchronize()
    gc.collect()
    free_before = torch.cuda.mem_get_info()[0]
    seen = {}

    def work():
        torch.cuda.set_device(0)
        ws = L.workspace(1 << 30)  # 1 GiB of scratch owned by this thread's context
        seen["bytes"] = L.workspace_bytes()
        seen["held"] = free_before - torch.cuda.mem_get_info()[0]
        seen["owner"] = weakref.ref(L._tls.handles[0])
        del ws

    t = threading.Thread(target=work)
    t.start()
    t.join()
    gc.collect()
    assert seen["bytes"] >= 1 << 30 and seen["held"] >= (1 << 30) - (64 << 20)
    assert seen["owner"]() is None  # the owner died with the thread ...
    torch.cuda.synchronize()
    assert free_before - torch.cuda.mem_get_info()[0] < (64 << 20)  # ... and the scratch went back to the driver


@pytest.mark.parametrize("N", [300, 301])
def test_device_side_metropolis_sweep_matches_host_loop(env, N):
    """ChainBatch.sweep_trees (decision on the device, one read-back per sweep) against the per-tree host loop of
    bark_sampler.py:233-264 driven through propose_trees / accept with the same proposals and uniform draws.
    N even: all chains in one launch sequence; N odd: one stream per chain."""
    fit, syn = env.fit, env.syn
    nc, m = 3, 8
    X, y, bounds, ft = syn.mixed_problem(N, seed=5)
    cur = syn.sample_prior_forests(nc, m, bounds, ft, seed=50)
    prop = syn.sample_prior_forests(nc, m, bounds, ft, seed=500)  # proposal for tree t of chain b: prop[b, t]
    noise, scale = np.array([0.1, 0.07, 0.2]), np.array([1.0, 0.8, 1.2])
    rng = np.random.default_rng(9)
    log_q = rng.normal(0.0, 0.5, size=(nc, m))
    log_u = np.log(rng.uniform(size=(nc, m)))
    host = fit.ChainBatch.from_forests(cur, noise, scale, X, y, ft)
    want_mask = np.zeros((nc, m), dtype=bool)
    for t in range(m):
        before = host.mll.copy()
        vals = host.propose_trees(cur[:, t], prop[:, t], X, ft, scale, m)
        want_mask[:, t] = log_u[:, t] <= np.minimum(log_q[:, t] + (vals - before), 0.0)
        host.accept(want_mask[:, t])
    dev = fit.ChainBatch.from_forests(cur, noise, scale, X, y, ft)
    mask = dev.sweep_trees(cur, prop, log_q, log_u, X, ft, scale, m)
    assert mask.shape == (nc, m) and np.array_equal(mask, want_mask)
    assert 0 < mask.sum() < mask.size  # both branches exercised
    assert np.allclose(dev.mll, host.mll, rtol=1e-12, atol=1e-10)
    assert bool((dev.K_inv == host.K_inv).all())  # same kernels, same order: identical bits
    # and the end state against a full recomputation by the oracle
    final = cur.copy()
    final[mask] = prop[mask]
    want = env.orc.batched_mll(final, noise, scale, X, y, ft, include_scale=True, include_2pi=False)
    assert np.allclose(dev.mll, want, rtol=1e-9, atol=1e-8)


def test_g11_reference_sampler_steps_replayed_on_the_device(env):
    """tests/golden/g11_sampler_steps.npz — the reference's own `_step_bark_sampler` (bark_sampler.py:217-284) recorded
    proposal by proposal — replayed (a) per chain through ChainState.propose_tree / accept / propose_noise_scale and
    (b) for both chains at once through ChainBatch.sweep_trees with the decision on the device: identical Metropolis
    decisions, new_mll and cur_mll within rtol 1e-9 of the reference's values."""
    from conftest import load_golden

    fit, orc = env.fit, env.orc
    g = load_golden("g11_sampler_steps")
    X, y, ft = g["X"], g["y"], g["feat_types"]
    chains, steps, m = g["accept"].shape
    tol = dict(rtol=1e-9, atol=1e-8)
    # (a) one chain at a time, host-side decisions from device MLLs
    for c in range(chains):
        forest = orc.nodes_from_raw(g["start_forest"][c]).copy()
        noise, scale = float(g["start_noise"][c]), float(g["start_scale"][c])
        st = fit.ChainState.from_forest(forest, noise, scale, X, y, ft)
        assert np.isclose(st.mll, g["start_mll"][c], **tol)
        for s in range(steps):
            old, new = orc.nodes_from_raw(g["old"][c, s]), orc.nodes_from_raw(g["new"][c, s])
            for t in range(m):
                before = st.mll
                val = st.propose_tree(old[t], new[t], X, ft, scale, m)
                assert np.isclose(val, g["new_mll"][c, s, t], **tol), (c, s, t, val, g["new_mll"][c, s, t])
                acc = bool(np.log(g["u"][c, s, t]) <= min(g["log_q"][c, s, t] + val - before, 0))
                assert acc == bool(g["accept"][c, s, t]), (c, s, t)
                if acc:
                    st.accept()
                    forest[t] = new[t]
                assert np.isclose(st.mll, g["cur_mll"][c, s, t], **tol)
            nn, nsc = (float(v) for v in g["ns_prop"][c, s])
            before = st.mll
            val = st.propose_noise_scale(forest, nn, nsc, X, ft)
            assert np.isclose(val, g["ns_new_mll"][c, s], **tol)
            acc = bool(np.log(g["ns_u"][c, s]) <= min(g["ns_log_q"][c, s] + val - before, 0))
            assert acc == bool(g["ns_accept"][c, s])
            if acc:
                st.accept()
                noise, scale = nn, nsc
            assert np.isclose(st.mll, g["mll_after"][c, s], **tol)
            assert np.array_equal(forest, orc.nodes_from_raw(g["forest_after"][c, s]))
    # (b) both chains, one device-side sweep per step (the noise/scale half rebuilds the batch, as the reference's
    # full inv + slogdet does)
    forests = orc.nodes_from_raw(g["start_forest"]).copy()
    noise, scale = g["start_noise"].copy(), g["start_scale"].copy()
    cb = fit.ChainBatch.from_forests(forests, noise, scale, X, y, ft)
    for s in range(steps):
        old, new = orc.nodes_from_raw(g["old"][:, s]), orc.nodes_from_raw(g["new"][:, s])
        assert np.array_equal(old, forests)
        mask = cb.sweep_trees(old, new, g["log_q"][:, s], np.log(g["u"][:, s]), X, ft, scale, m)
        assert np.array_equal(mask, g["accept"][:, s])
        forests[mask] = new[mask]
        assert np.allclose(cb.mll, g["cur_mll"][:, s, -1], **tol)
        vals = fit.batched_mll(forests, g["ns_prop"][:, s, 0], g["ns_prop"][:, s, 1], X, y, ft, include_scale=True,
                               include_2pi=False)
        assert np.allclose(vals, g["ns_new_mll"][:, s], **tol)
        acc = np.log(g["ns_u"][:, s]) <= np.minimum(g["ns_log_q"][:, s] + vals - cb.mll, 0)
        assert np.array_equal(acc, g["ns_accept"][:, s])
        noise = np.where(acc, g["ns_prop"][:, s, 0], noise)
        scale = np.where(acc, g["ns_prop"][:, s, 1], scale)
        assert np.array_equal(noise, g["noise_after"][:, s]) and np.array_equal(scale, g["scale_after"][:, s])
        if acc.any():
            cb = fit.ChainBatch.from_forests(forests, noise, scale, X, y, ft)
        assert np.allclose(cb.mll, g["mll_after"][:, s], **tol)


def test_singular_low_rank_system_raises_linalgerror(env):
    """quick_inverse.py:19,31: np.linalg.solve / slogdet raise on a singular (mul I + U' K_inv U)."""
    from bark_amd.fitting import quick_inverse as qi

    K_inv = np.eye(6)
    U = np.zeros((6, 1))
    U[0, 0] = 1.0  # subtract: -1 + e0' I e0 = 0
    with pytest.raises(np.linalg.LinAlgError):
        qi.low_rank_inv_update(K_inv, U, subtract=True)
    with pytest.raises(np.linalg.LinAlgError):
        qi.low_rank_det_update(K_inv, U, 0.0, subtract=True)
    with pytest.raises(np.linalg.LinAlgError):
        env.orc.low_rank_inv_update(K_inv, U, subtract=True)  # the oracle (numpy) behaves the same way
    out = qi.low_rank_inv_update(K_inv, U, subtract=False)
    assert np.allclose(out, env.orc.low_rank_inv_update(K_inv, U, subtract=False))


def test_invalid_category_raises_only_where_a_walk_evaluates_it(env):
    """forest.py:37-39: `1 << int(x)` raises for NaN / inf / negative x — but only when a walk reaches a categorical
    split with that value.  A bad value on a point whose walks never test the feature passes, as in the reference."""
    bf, syn = env.bf, env.syn
    ft = np.array([2, 0])  # feature 0 continuous, feature 1 categorical
    forest = bf.create_empty_forest(1, 8)
    # root: x0 <= 0.5 -> left = leaf 1 ; right = node 2: categorical split on feature 1 (mask 0b0101) -> leaves 3, 4
    forest[0, 0] = (0, 0, 0.5, 1, 2, 0xFFFFFFFF, 0, 1)
    forest[0, 1] = (1, 0, 0, 0, 0, 0, 1, 1)
    forest[0, 2] = (0, 1, float(0b0101), 3, 4, 0, 1, 1)
    forest[0, 3] = (1, 0, 0, 0, 0, 2, 2, 1)
    forest[0, 4] = (1, 0, 0, 0, 0, 2, 2, 1)
    X = np.array([[0.2, 1.0], [0.9, 0.0], [0.9, 1.0], [0.9, 2.0]])
    assert np.array_equal(bf.pass_through_forest(forest, X, ft)[:, 0], [1, 3, 4, 3])
    assert np.array_equal(bf.pass_through_forest(forest, X, ft), env.orc.pass_through_forest(forest, X, ft))
    for bad in (np.nan, -1.0, np.inf, -np.inf):
        ok = X.copy()
        ok[0, 1] = bad  # point 0 goes left at the root: the categorical split is never evaluated for it
        assert np.array_equal(bf.pass_through_forest(forest, ok, ft)[:, 0], [1, 3, 4, 3])
        assert np.array_equal(env.orc.pass_through_forest(forest, ok, ft)[:, 0], [1, 3, 4, 3])
        hit = X.copy()
        hit[2, 1] = bad
        with pytest.raises(ValueError):
            bf.pass_through_forest(forest, hit, ft)
        with pytest.raises(ValueError):
            bf.forest_gram_matrix(forest, hit, hit, ft)
        with pytest.raises(ValueError):
            env.fit.batched_mll(forest[None], [0.1], [1.0], hit, np.arange(4.0), ft, include_scale=True, include_2pi=False)
        # the fault flag does not leak into the next call
        assert np.array_equal(bf.pass_through_forest(forest, X, ft)[:, 0], [1, 3, 4, 3])
    assert np.array_equal(bf.pass_through_forest(forest, np.array([[0.9, -0.5]]), ft)[:, 0], [3])  # int(-0.5) == 0


def test_node_with_two_parents_walks_like_the_reference(env):
    """The device walk is bounded by the packer's max_depth: with a shared node discovered first on a shallow path the
    bound must still cover the deep one (longest path), otherwise a point ends on an internal node."""
    bf = env.bf
    nodes = bf.create_empty_forest(1, 8)
    nodes[0, 0] = (0, 0, 0.5, 1, 2, 0xFFFFFFFF, 0, 1)
    nodes[0, 1] = (0, 0, 0.25, 3, 7, 0, 1, 1)
    nodes[0, 2] = (0, 0, 0.75, 4, 7, 0, 1, 1)
    nodes[0, 4] = (0, 0, 0.6, 3, 7, 2, 2, 1)
    nodes[0, 3] = (0, 0, 0.1, 5, 6, 1, 2, 1)
    for leaf in (5, 6, 7):
        nodes[0, leaf] = (1, 0, 0, 0, 0, 0, 3, 1)
    ft = np.array([2], dtype=np.int64)
    X = np.array([[0.55], [0.2], [0.05], [0.9], [0.3]])
    want = env.orc.pass_through_forest(nodes, X, ft)
    assert np.array_equal(bf.pass_through_forest(nodes, X, ft), want)
    assert np.array_equal(bf.forest_gram_matrix(nodes, X, X, ft), env.orc.forest_gram_matrix(nodes, X, X, ft))


def test_product_path_launches_no_torch_compute_kernels(env):
    """torch is the device-memory container of the product, not its arithmetic: a pass over the API surface under the
    profiler must show only bark:: kernels and runtime copies / fills — no at::native::* elementwise, reduction or
    indexing kernels and no rocBLAS."""
    from torch.profiler import ProfilerActivity, profile

    import bark_amd.tree_kernels as tk
    from bark_amd.fitting import quick_inverse as qi

    torch, bf, fit, syn = env.torch, env.bf, env.fit, env.syn
    N, C, nb, m = 300, 70, 4, 12
    X, y, bounds, ft = syn.mixed_problem(N, seed=1)
    cand = syn.mixed_problem(C, seed=2)[0]
    F = syn.sample_prior_forests(nb, m, bounds, ft, seed=3)
    noise, scale = np.linspace(0.05, 0.2, nb), np.linspace(0.8, 1.2, nb)
    Xd, cd = torch.from_numpy(X).cuda(), torch.from_numpy(cand).cuda()
    rng = np.random.default_rng(0)

    def surface():
        bf.pass_through_forest(F[0], Xd, ft)
        U = bf.get_leaf_vectors(F[0][0], Xd, ft)
        bf.batched_forest_gram_matrix(F, Xd, Xd, ft)
        bf.batched_forest_gram_matrix_no_null(F, X, X, ft)
        fit.batched_mll(F, noise, scale, Xd, y, ft, include_scale=True, include_2pi=False, return_device=True)
        fit.batched_mll(F, noise, None, X, y, ft, include_scale=False, include_2pi=True, method="leafspace")
        mu, var = tk.forest_predict((F, noise, scale), (Xd, y), cd, ft)
        tk.forest_predict((F, noise, scale), (X, y), cand, ft, diag=False)
        tk.mixture_of_gaussians_as_normal(mu, var)
        K_inv, _, logdet = fit.batched_kernel_inverse(F, noise, scale, Xd, y, ft, no_null=False, return_device=True)
        fit.batched_kernel_inverse(F[:1], noise[:1], scale[:1], X, y, ft, no_null=True)
        qi.low_rank_inv_update(K_inv[0], U, subtract=False, assume_symmetric=True)
        qi.low_rank_det_update(K_inv[0], U, float(logdet[0].item()))
        qi.mll(K_inv[0], float(logdet[0].item()), y)
        st = fit.ChainState.from_forest(F[0], 0.1, 1.0, Xd, y, ft)
        st.propose_tree(F[0][0], F[1][0], Xd, ft, 1.0, m)
        st.accept()
        st.propose(bf.get_leaf_vectors(F[0][1], X, ft), bf.get_leaf_vectors(F[1][1], X, ft))
        st.accept()
        st.propose_noise_scale(F[0], 0.12, 1.1, Xd, ft)
        st.accept()
        cb = fit.ChainBatch.from_forests(F[:2], noise[:2], scale[:2], Xd, y, ft)
        cb.propose_trees(F[:2, 0], F[2:4, 0], Xd, ft, scale[:2], m)
        cb.accept([True, False])
        cb.sweep_trees(F[:2, :6], F[2:4, :6], rng.normal(size=(2, 6)), np.log(rng.uniform(size=(2, 6))), Xd, ft, scale[:2], m)
        torch.cuda.synchronize()

    surface()  # warm-up: library load, contexts
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        surface()
    names = [e.key for e in prof.key_averages()]
    assert any("bark" in n for n in names), names
    foreign = [n for n in names if "at::native" in n or "rocblas" in n.lower() or "Cijk" in n]
    assert not foreign, foreign


def test_pipelined_schedule_is_hipgraph_capturable(env):
    """Sweep::step_pipelined forks the bulk row launches onto both helper streams two steps ahead; every one of them is
    awaited on the caller's stream at its own step, so the call is still a capturable fork/join."""
    import bench

    torch = env.torch
    wl = bench.Workload(1200, 8, 50, 130, seed_base=1200, rank_offset=0)  # 10 block rows x 130 matrices: pipelined
    wl.run()
    torch.cuda.synchronize()
    eager = wl.mll_d.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        wl.stream = env.lib.stream_ptr()
        wl.run()
    wl.mll_d.zero_()
    g.replay()
    g.replay()
    torch.cuda.synchronize()
    assert bool((wl.mll_d == eager).all()) and int(wl.info_d.abs().max().item()) == 0


# ---- round 4: error paths of the sweep, the host-pointer entry points ---------------------------------------------------

# (N, forests, schedule, which checked launch fails: early / middle / late in the call)
_FAULT_SHAPES = ((1100, 24, "split-K layout, device-side hand-over and gates", (3, 11, 29)), (1100, 40, "pipelined, event joins", (3, 11, 29)),
                 (700, 40, "plain (6 block rows)", (3, 9, 17)))


def test_error_return_rejoins_helper_streams(env):
    """a non-OK return from the middle of Sweep::step used to leave the helper streams forked and never
    joined to the caller's stream.  bark_debug_fail_launch(k) makes the k-th checked launch report a failure: the call
    returns BARK_ERR_HIP (RuntimeError), the device drains, and the next call on the SAME context and workspace is
    bit-identical to the result before — for the three schedules, with the failure early, in the middle and late."""
    import bench

    torch, lib = env.torch, env.lib.lib()
    for N, Bn, what, ks in _FAULT_SHAPES:
        wl = bench.Workload(N, 8, 50, Bn, seed_base=N, rank_offset=0)
        wl.run()
        torch.cuda.synchronize()
        good = wl.mll_d.clone()
        for k in ks:
            assert lib.bark_debug_fail_launch(k) == 0
            with pytest.raises(RuntimeError, match="hipErrorLaunchFailure|unspecified launch failure|failed"):
                wl.run()
            assert lib.bark_debug_fail_launch(0) == 0, (what, k)  # the countdown was consumed by this call
            torch.cuda.synchronize()  # nothing of the failed call is left running or waiting
            wl.mll_d.zero_()
            wl.run()
            torch.cuda.synchronize()
            assert bool((wl.mll_d == good).all()) and int(wl.info_d.abs().max().item()) == 0, (what, k)


_CAPTURE_FAULT_SCRIPT = r"""
import sys
sys.path.insert(0, {root!r})
import torch
import bench
from bark_amd import _lib
lib = _lib.lib()
for N, Bn in ((1100, 24), (2100, 6), (4096, 16)):
    wl = bench.Workload(N, 8, 50, Bn, seed_base=N, rank_offset=0)
    wl.run()
    torch.cuda.synchronize()
    good = wl.mll_d.clone()
    for k in (4, 17):
        g = torch.cuda.CUDAGraph()
        raised = False
        try:
            with torch.cuda.graph(g):
                wl.stream = _lib.stream_ptr()
                lib.bark_debug_fail_launch(k)
                wl.run()
        except RuntimeError as e:  # the injected failure; capture_end ran in __exit__ and did not crash
            raised = "failed" in str(e) or "Failure" in str(e)
        lib.bark_debug_fail_launch(0)
        assert raised, (N, Bn, k)
        del g
        torch.cuda.synchronize()
        wl.stream = _lib.stream_ptr()
        wl.mll_d.zero_()
        wl.run()
        torch.cuda.synchronize()
        assert bool((wl.mll_d == good).all()), (N, Bn, k)
        # and a healthy capture still works after the failed one
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            wl.stream = _lib.stream_ptr()
            wl.run()
        wl.mll_d.zero_()
        g2.replay()
        torch.cuda.synchronize()
        assert bool((wl.mll_d == good).all()), (N, Bn, k, "replay")
        wl.stream = _lib.stream_ptr()
print("capture-fault ok")
"""


def test_error_return_under_stream_capture_ends_the_capture_cleanly():
    """The same injected failure while the caller's stream is being captured: every helper stream the call forked is joined
    back before it returns, so torch's capture_end (hipStreamEndCapture) succeeds instead of meeting an unjoined capture —
    which on this ROCm is a crash, not an error code (profiles/r04/capture_unjoined_probe.txt; round 3's segfault in
    capture_end).  Run in a child process: a regression would otherwise take the whole test session down with it."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _CAPTURE_FAULT_SCRIPT.format(root=root)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "capture-fault ok" in r.stdout, (r.returncode, r.stdout[-800:], r.stderr[-1500:])


def test_pipelined_sweep_of_16_matrices_is_capturable(env):
    """N = 4096 x 16 (pipelined schedule, 32 block rows, two bulk streams) is one of the two sizes at which round 3's
    chain-split experiment crashed hipGraphInstantiate; the shipped schedule's graph of that size is pinned here (the other,
    N = 6900 x 1, is in test_sweep_is_hipgraph_capturable)."""
    import bench

    torch = env.torch
    wl = bench.Workload(4096, 8, 50, 16, seed_base=4096, rank_offset=0)
    wl.run()
    torch.cuda.synchronize()
    eager = wl.mll_d.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        wl.stream = env.lib.stream_ptr()
        wl.run()
    wl.mll_d.zero_()
    g.replay()
    g.replay()
    torch.cuda.synchronize()
    # under capture the library uses event joins, eagerly the device-side hand-over: the same kernels in the same order
    assert bool((wl.mll_d == eager).all()) and int(wl.info_d.abs().max().item()) == 0


_SERIALISED_SCRIPT = """
import os, sys, time
sys.path.insert(0, {root!r})
import torch, bench
from bark_amd import _lib
wl = bench.Workload(1100, 8, 50, 8, seed_base=1100, rank_offset=0)
wl.run(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    wl.run()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("STATE", bench.device_wait_state(_lib.lib()), "INFO", int(wl.info_d.abs().max().item()), "SECONDS", dt)
print("MLL", wl.mll_d.cpu().numpy().tobytes().hex())
"""


@pytest.mark.parametrize("var", ["HIP_LAUNCH_BLOCKING", "AMD_SERIALIZE_KERNEL", "BARK_NO_DEVICE_WAIT"])
def test_dispatch_serialising_environments_switch_the_device_wait_off(env, var):
    """under serialised dispatch every device-side wait would run into its 2 s bound (the row launch a
    diag_kernel waits for cannot run beside it).  The library reads the environment once: with HIP_LAUNCH_BLOCKING,
    AMD_SERIALIZE_KERNEL or BARK_NO_DEVICE_WAIT set it uses event joins from the start.  A child process per variable (the
    switch is process-wide and read at first use): chunk of 8 matrices, 9 block rows — a shape that takes the device-side
    hand-over otherwise — mechanism reported off, no time-out (info == 0), three calls in well under one wait bound, and the
    same bits as this process computes with the mechanism on."""
    import os
    import subprocess
    import sys

    import bench

    torch = env.torch
    wl = bench.Workload(1100, 8, 50, 8, seed_base=1100, rank_offset=0)
    wl.run()
    torch.cuda.synchronize()
    want = wl.mll_d.cpu().numpy().tobytes().hex()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child_env = dict(os.environ)
    child_env[var] = "3" if var == "AMD_SERIALIZE_KERNEL" else "1"
    r = subprocess.run([sys.executable, "-c", _SERIALISED_SCRIPT.format(root=root)], capture_output=True, text=True, timeout=600,
                       env=child_env)
    assert r.returncode == 0, r.stderr[-2000:]
    state = [ln for ln in r.stdout.splitlines() if ln.startswith("STATE")][0].split()
    assert state[1] == "off" and int(state[3]) == 0 and float(state[5]) < 1.5, state
    got = [ln for ln in r.stdout.splitlines() if ln.startswith("MLL")][0].split()[1]
    assert got == want


_CONTENDED_SCRIPT = """
import os, sys, time
sys.path.insert(0, {root!r})
import torch, bench
from bark_amd import _lib
lib = _lib.lib()
wl = bench.Workload(1100, 8, 50, 8, seed_base=1100, rank_offset=0)
fallbacks = 0
for _ in range(40):
    wl.run()
    torch.cuda.synchronize()
    if bool((wl.info_d == -3).any().item()):  # a bounded wait ran out: the documented fallback (include/bark_hip.h, bark_device_wait)
        fallbacks += 1
        lib.bark_device_wait(0)
        wl.run()
        torch.cuda.synchronize()
print("STATE", bench.device_wait_state(lib), "INFO", int(wl.info_d.abs().max().item()), "FALLBACKS", fallbacks)
print("MLL", wl.mll_d.cpu().numpy().tobytes().hex())
"""


def test_device_wait_with_four_processes_on_one_gpu(env):
    """VERDICT r3 "what's weak" 8: the device-side hand-over had never run with several PROCESSES on a card (eight ranks on a node
    is where helper-stream concurrency may differ).  Four processes at once on this GPU, each 40 sweeps of a chunk of 8 matrices
    (device-side waits in every block step) while this process keeps a fifth stream of the same work going: every process ends
    with info == 0 and the bits of an undisturbed run, and says whether a wait ever ran into its bound (then it took the
    documented fallback; on the boxes so far none did).  The loop has a wall-clock deadline and the children are reaped on
    every exit."""
    import os
    import subprocess
    import sys
    import time

    import bench

    torch = env.torch
    wl = bench.Workload(1100, 8, 50, 8, seed_base=1100, rank_offset=0)
    wl.run()
    torch.cuda.synchronize()
    want = wl.mll_d.cpu().numpy().tobytes().hex()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = [subprocess.Popen([sys.executable, "-c", _CONTENDED_SCRIPT.format(root=root)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True) for _ in range(4)]
    deadline = time.monotonic() + 300.0  # four children x 40 sweeps of ~1 ms + start-up: minutes would mean a hang
    lib, parent_fallbacks = env.lib.lib(), 0
    try:
        while any(p.poll() is None for p in procs):  # the parent competes for the card as well
            assert time.monotonic() < deadline, "a child process is still running after 300 s"
            wl.run()
            torch.cuda.synchronize()
            if bool((wl.info_d == -3).any().item()):  # the parent takes the documented fallback like its children
                parent_fallbacks += 1
                lib.bark_device_wait(0)
                wl.run()
                torch.cuda.synchronize()
        assert int(wl.info_d.abs().max().item()) == 0 and wl.mll_d.cpu().numpy().tobytes().hex() == want
        outcomes = [("parent", parent_fallbacks)]
        for p in procs:
            out, err = p.communicate(timeout=60)
            assert p.returncode == 0, err[-2000:]
            state = [ln for ln in out.splitlines() if ln.startswith("STATE")][0].split()
            # bits: always those of an undisturbed run.  Whether a 2 s wait ever ran out with five processes on the card's
            # queues is a property of the box's scheduling, not of the code: a child that took the documented fallback
            # (info == -3, mechanism off, same call again) reports it, and must then have ended with the mechanism off
            assert int(state[3]) == 0, state
            assert (state[1] == "on") == (int(state[5]) == 0), state
            assert [ln for ln in out.splitlines() if ln.startswith("MLL")][0].split()[1] == want
            outcomes.append((state[1], int(state[5])))
        print("device-side wait under five processes: (state, fallbacks) per child =", outcomes)
    finally:
        lib.bark_device_wait(1)  # process-wide switch: later tests expect the default
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            try:
                p.wait(timeout=30)
            except Exception:  # noqa: BLE001
                pass


def test_device_wait_from_a_side_stream_and_from_two_threads(env):
    """the device-side hand-over had only been driven from the default stream.  Here (a) from a non-default
    caller stream and (b) from two host threads at once, each with its own context and stream: bit-identical to the serial
    default-stream result, no time-out."""
    import bench

    torch, L = env.torch, env.lib
    shapes = ((2100, 6), (1500, 1))
    ref = []
    for N, Bn in shapes:
        wl = bench.Workload(N, 8, 50, Bn, seed_base=N, rank_offset=0)
        wl.run()
        torch.cuda.synchronize()
        ref.append(wl.mll_d.clone())
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for (N, Bn), want in zip(shapes, ref):
            wl = bench.Workload(N, 8, 50, Bn, seed_base=N, rank_offset=0)
            assert wl.stream.value == side.cuda_stream
            for _ in range(3):
                wl.run()
            side.synchronize()
            assert bool((wl.mll_d == want).all()) and int(wl.info_d.abs().max().item()) == 0
    out, errs = {}, []

    def work(i):
        try:
            torch.cuda.set_device(0)
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                N, Bn = shapes[i]
                wl = bench.Workload(N, 8, 50, Bn, seed_base=N, rank_offset=0)
                for _ in range(5):
                    wl.run()
                st.synchronize()
                out[i] = (wl.mll_d.clone(), int(wl.info_d.abs().max().item()))
            L.release_ctx()
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for i, want in enumerate(ref):
        assert out[i][1] == 0 and bool((out[i][0] == want).all())


def test_g11_replayed_through_host_pointer_entry_points_only():
    """the reference's sampler step is numba nopython code (bark_sampler.py:216 `@njit _step_bark_sampler`):
    it can reach C through ctypes function pointers with integers, floats and array addresses — no tensor objects.  This
    replays chain 0 of g11 (the reference's own recorded trajectory) with exactly that vocabulary: a private ctypes handle,
    plain ints for every pointer, numpy host arrays, bark_dev_alloc / bark_ctx_upload / bark_ctx_download for the device
    side, bark_tree_swap_eval_host_pair per tree proposal and bark_mll_batched_hip for the noise/scale half.  No torch in the
    loop (torch is only in the process because other tests imported it)."""
    import ctypes as C

    from conftest import load_golden

    from bark_amd import _lib
    from oracle import oracle as orc

    _lib.lib()  # (loads torch's HIP runtime first, as every process of this suite does)
    h = C.CDLL(_lib.LIB_PATH)
    I, D, P = C.c_int64, C.c_double, C.c_void_p
    h.bark_ctx_create.argtypes = [C.c_int, C.POINTER(P)]
    h.bark_dev_alloc.argtypes = [P, C.c_size_t, C.POINTER(P)]
    h.bark_dev_free.argtypes = [P, P]
    h.bark_ctx_upload.argtypes = [P, P, P, C.c_size_t, P]
    h.bark_ctx_download.argtypes = [P, P, P, C.c_size_t, P]
    h.bark_forest_pack_info.argtypes = [P, I, I, I, P, I, P]
    h.bark_forest_pack.argtypes = [P, P, I, P, P]
    h.bark_mll_workspace_bytes.restype = C.c_size_t
    h.bark_mll_workspace_bytes.argtypes = [I, I, I, I]
    h.bark_tree_swap_workspace_bytes.restype = C.c_size_t
    h.bark_tree_swap_workspace_bytes.argtypes = [I, I]
    h.bark_mll_batched_hip.argtypes = [P, P, P, P, I, I, P, P, P, P, C.c_int, P, I, P, P, P, P, P, P, C.c_size_t, I, P, P]
    h.bark_tree_swap_eval_host_pair.argtypes = [P, P, I, P, I, P, I, P, D, P, P, P, P, C.c_size_t, P]
    h.bark_lowrank_swap_apply_hip.argtypes = [P, I, I, P, P, P]
    h.bark_ctx_destroy.argtypes = [P]
    h.bark_last_error.restype = C.c_char_p

    def ok(rc):
        assert rc == 0, h.bark_last_error()

    g = load_golden("g11_sampler_steps")
    X, y, ft = np.ascontiguousarray(g["X"]), np.ascontiguousarray(g["y"]).reshape(-1), np.ascontiguousarray(g["feat_types"], dtype=np.int64)
    N, d = X.shape
    chains, steps, m = g["accept"].shape
    tol = dict(rtol=1e-9, atol=1e-8)
    ctx = P()
    ok(h.bark_ctx_create(0, C.byref(ctx)))
    allocs = []

    def dev(nbytes):
        p = P()
        ok(h.bark_dev_alloc(ctx, int(nbytes), C.byref(p)))
        allocs.append(p)
        return p.value  # a plain int from here on

    def up(dst, arr):
        ok(h.bark_ctx_upload(ctx, dst, arr.ctypes.data, arr.nbytes, None))

    Xd, yd = dev(X.nbytes), dev(y.nbytes)
    up(Xd, X)
    up(yd, y)
    kinv, kinv_y, kdiag = dev(8 * N * N), dev(8 * N), dev(8 * N)
    scal, mll_d, info_d = dev(64), dev(8), dev(8)
    L = g["start_forest"].shape[-2] if g["start_forest"].ndim == 4 else orc.nodes_from_raw(g["start_forest"][0]).shape[-1]
    packed_d = dev(m * L * 16 + 256)
    ws_bytes = int(h.bark_mll_workspace_bytes(N, N, m, 1))
    ws = dev(ws_bytes)
    swap_bytes = int(h.bark_tree_swap_workspace_bytes(N, 64))
    swap_ws = dev(swap_bytes)
    info = _lib.PackInfo()

    def full_state(forest, noise, scale):
        """bark_sampler.py:153-162 / 267-272: K_inv, quad = y'K^-1 y and log|K| of the whole forest (RHS = identity)."""
        nodes = np.ascontiguousarray(forest)
        ok(h.bark_forest_pack_info(nodes.ctypes.data, 1, m, nodes.shape[-1], ft.ctypes.data, d, C.addressof(info)))
        host_packed = np.empty(info.packed_bytes, dtype=np.uint8)
        ok(h.bark_forest_pack(nodes.ctypes.data, ft.ctypes.data, d, C.addressof(info), host_packed.ctypes.data))
        up(packed_d, host_packed)
        ns = np.array([noise, scale, 0.0, 0.0])
        up(scal, ns)
        ok(h.bark_mll_batched_hip(ctx, packed_d, C.addressof(info), Xd, N, d, yd, scal, scal + 8, None, 1 | 4, None, N, mll_d, kinv_y,
                                  kdiag, kinv, info_d, ws, ws_bytes, 1, None, None))
        out, ky, flag = np.empty(1), np.empty(N), np.empty(1, dtype=np.int32)
        ok(h.bark_ctx_download(ctx, out.ctypes.data, mll_d, 8, None))
        ok(h.bark_ctx_download(ctx, ky.ctypes.data, kinv_y, 8 * N, None))
        ok(h.bark_ctx_download(ctx, flag.ctypes.data, info_d, 4, None))
        assert flag[0] == 0
        quad = float(ky @ y)
        return quad, -2.0 * float(out[0]) - quad  # mll = 0.5 (-quad - logdet)

    c = 0
    forest = orc.nodes_from_raw(g["start_forest"][c]).copy()
    noise, scale = float(g["start_noise"][c]), float(g["start_scale"][c])
    quad, logdet = full_state(forest, noise, scale)
    assert np.isclose(0.5 * (-quad - logdet), g["start_mll"][c], **tol)
    scalars, r_out = np.empty(2), np.zeros(1, dtype=np.int64)
    n_acc = 0
    for s in range(steps):
        old, new = orc.nodes_from_raw(g["old"][c, s]), orc.nodes_from_raw(g["new"][c, s])
        for t in range(m):
            pair = np.ascontiguousarray(np.stack([forest[t], new[t]]))
            assert np.array_equal(forest[t], old[t])
            ok(h.bark_tree_swap_eval_host_pair(ctx, kinv, N, pair.ctypes.data, pair.shape[1], ft.ctypes.data, d, Xd,
                                               float(np.sqrt(scale / m)), yd, scalars.ctypes.data, r_out.ctypes.data, swap_ws,
                                               swap_bytes, None))
            cur_mll = 0.5 * (-quad - logdet)
            new_mll = 0.5 * (-(quad - scalars[0]) - (logdet + scalars[1]))
            assert np.isclose(new_mll, g["new_mll"][c, s, t], **tol), (s, t, new_mll, g["new_mll"][c, s, t])
            acc = bool(np.log(g["u"][c, s, t]) <= min(g["log_q"][c, s, t] + new_mll - cur_mll, 0))
            assert acc == bool(g["accept"][c, s, t]), (s, t)
            if acc:
                ok(h.bark_lowrank_swap_apply_hip(kinv, N, int(r_out[0]), swap_ws, kinv, None))
                quad, logdet = quad - scalars[0], logdet + scalars[1]
                forest[t] = new[t]
                n_acc += 1
            assert np.isclose(0.5 * (-quad - logdet), g["cur_mll"][c, s, t], **tol)
        nn, nsc = (float(v) for v in g["ns_prop"][c, s])
        cur_mll = 0.5 * (-quad - logdet)
        q2, l2 = full_state(forest, nn, nsc)  # rewrites K_inv; on a rejection the old state is rebuilt below
        val = 0.5 * (-q2 - l2)
        assert np.isclose(val, g["ns_new_mll"][c, s], **tol)
        acc = bool(np.log(g["ns_u"][c, s]) <= min(g["ns_log_q"][c, s] + val - cur_mll, 0))
        assert acc == bool(g["ns_accept"][c, s])
        if acc:
            noise, scale, quad, logdet = nn, nsc, q2, l2
        else:
            quad, logdet = full_state(forest, noise, scale)
        assert np.isclose(0.5 * (-quad - logdet), g["mll_after"][c, s], **tol)
        assert np.array_equal(forest, orc.nodes_from_raw(g["forest_after"][c, s]))
    assert n_acc > 0
    for p in allocs:
        ok(h.bark_dev_free(ctx, p))
    h.bark_ctx_destroy(ctx)


def test_host_pointer_entry_points_refuse_what_they_cannot_do(env):
    """bark_dev_alloc / bark_ctx_upload / bark_ctx_download / bark_tree_swap_eval_host_pair (include/bark_hip.h): status codes, not
    crashes, for null pointers, a zero-byte allocation and a tree pair with more than 64 leaves (where the reference's own
    subtract-then-add chain through bark_lowrank_update_hip is the route); a staged round trip of 1 byte, 64 KiB (the pinned
    page) and 1 MiB (pageable) returns the bytes it was given."""
    import ctypes as C

    L = env.lib
    lib, ctx = L.lib(), L.ctx()
    p = C.c_void_p()
    assert lib.bark_dev_alloc(ctx, 0, C.byref(p)) == L.BARK_ERR_ARG
    assert lib.bark_dev_alloc(ctx, 16, None) == L.BARK_ERR_ARG
    assert lib.bark_ctx_upload(ctx, None, None, 8, None) == L.BARK_ERR_ARG
    assert lib.bark_ctx_download(ctx, None, None, 8, None) == L.BARK_ERR_ARG
    for nbytes in (1, 64 * 1024, 1 << 20):
        src = np.random.default_rng(nbytes).integers(0, 256, size=nbytes, dtype=np.uint8)
        back = np.zeros_like(src)
        assert lib.bark_dev_alloc(ctx, nbytes, C.byref(p)) == 0
        assert lib.bark_ctx_upload(ctx, p, src.ctypes.data, nbytes, L.stream_ptr()) == 0
        src_copy = src.copy()
        src[:] = 0  # the host buffer may be reused as soon as the call returns
        assert lib.bark_ctx_download(ctx, back.ctypes.data, p, nbytes, L.stream_ptr()) == 0
        assert np.array_equal(back, src_copy)
        assert lib.bark_dev_free(ctx, p) == 0
    # a pair of complete depth-6 trees: 64 + 64 leaves > 64
    syn = env.syn
    N, d = 256, 4
    X = np.random.default_rng(1).uniform(size=(N, d))
    ft = np.full(d, 2, dtype=np.int64)
    big = syn.full_binary_forests(1, 2, d, 6, np.random.default_rng(3), node_limit=128)[0]  # (2, 128): 64 leaves per tree
    Xd = L.to_device(X)
    yd = L.to_device(np.zeros(N))
    K = L.to_device(np.eye(N))
    ws = env.torch.empty(int(lib.bark_tree_swap_workspace_bytes(N, 64)), dtype=env.torch.uint8, device=Xd.device)
    scal, r_out = np.zeros(2), np.zeros(1, dtype=np.int64)
    pair = np.ascontiguousarray(big)
    rc = lib.bark_tree_swap_eval_host_pair(ctx, L.ptr(K), N, pair.ctypes.data, pair.shape[1], ft.ctypes.data, d, L.ptr(Xd), 0.1, L.ptr(yd),
                                           scal.ctypes.data, r_out.ctypes.data, L.ptr(ws), ws.numel(), L.stream_ptr())
    assert rc == L.BARK_ERR_ARG and b"64" in lib.bark_last_error()
    assert lib.bark_tree_swap_eval_host_pair(ctx, L.ptr(K), N, None, pair.shape[1], ft.ctypes.data, d, L.ptr(Xd), 0.1, L.ptr(yd),
                                             scal.ctypes.data, r_out.ctypes.data, L.ptr(ws), ws.numel(), L.stream_ptr()) == L.BARK_ERR_ARG
