"""BASELINE.json configs c1 … c5 at their stated sizes (c3 at B = 256 here; its size-independent properties are in
test_gpu_parity.py::test_c3_full_size_properties).  Inputs follow SURVEY §8d; the checker is the CPU oracle (LU route = the reference's
arithmetic) where it finishes in seconds and a CPU Cholesky for the one N = 16384 matrix of c5."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MLL_RTOL, MLL_ATOL = 1e-9, 1e-8


def test_c1_tree_function_n64():
    """configs[0]: TreeFunction benchmark (dim 5, 50 trees, function_seed 1), N = 64, one prior forest."""
    import bark_amd.fitting as fit
    from bark_amd import synthetic as syn
    from oracle import oracle as orc

    _forest, _leaf_values, f = syn.tree_function()
    X = np.random.default_rng(64).uniform(size=(64, 5))
    y = f(X).reshape(-1, 1)
    y = (y - y.mean()) / y.std()
    ft = np.full(5, 2)
    prior = syn.sample_prior_forests(1, 50, np.tile([[0.0, 1.0]], (5, 1)), ft, seed=64)
    for conv in (dict(include_scale=True, include_2pi=False), dict(include_scale=False, include_2pi=True)):
        got = fit.batched_mll(prior, [0.1], [1.0], X, y, ft, **conv)
        want = orc.batched_mll(prior, [0.1], [1.0], X, y, ft, **conv)
        assert np.allclose(got, want, rtol=MLL_RTOL, atol=MLL_ATOL)


def test_c2_n1024_single_forest():
    """configs[1]: N = 1024, d = 8, 50 trees, one forest: leaves and Gram bit-exact, MLL within tolerance."""
    import bark_amd.fitting as fit
    import bark_amd.forest as bf
    from bark_amd import synthetic as syn
    from oracle import oracle as orc

    X, y, bounds, ft = syn.unit_cube_problem(1024, 8, seed=1024)
    F = syn.sample_prior_forests(1, 50, bounds, ft, seed=1024)
    assert np.array_equal(bf.pass_through_forest(F[0], X, ft), orc.pass_through_forest(F[0], X, ft))
    assert np.array_equal(bf.forest_gram_matrix(F[0], X, X, ft), orc.forest_gram_matrix(F[0], X, X, ft))
    got = fit.batched_mll(F, [0.1], [1.0], X, y, ft, include_scale=True, include_2pi=True)
    want = orc.batched_mll(F, [0.1], [1.0], X, y, ft, include_scale=True, include_2pi=True)
    assert np.allclose(got, want, rtol=MLL_RTOL, atol=MLL_ATOL), (got, want)


def test_c3_b256_plain_schedule():
    """configs[2], the headline shape exactly: N = 4096, d = 8, m = 50, B = 256 forests seeded 4096 + b, noise
    U[0.05, 0.15) (examples/mcmc/mcmc_record_mll.py:57-74 convention).  256 resident matrices of 32 block
    rows is the one shape class that takes Sweep's PLAIN schedule with ragged-round splitting off (bc % 256 == 0 and
    nrb >= 16; since round 4 its row launches end with one SYRK workgroup per matrix for the diagonal tile, and two block rows
    share a launch: `paired`).  Sixteen samples spread over the batch against the oracle's LU route (the reference's arithmetic;
    the remaining 240 are vendor-checked in test_c3_b256_all_samples_against_vendor_cholesky), bit-reproducibility of the call,
    and agreement with the pipelined schedule (chunk = 64)."""
    import torch

    import bark_amd.fitting as fit
    from bark_amd import synthetic as syn
    from oracle import oracle as orc

    N, B, m = 4096, 256, 50
    X, y, bounds, ft = syn.unit_cube_problem(N, 8, seed=N)
    F = syn.sample_prior_forests(B, m, bounds, ft, seed=N)
    noise = np.random.default_rng(N).uniform(0.05, 0.15, B)
    Xd = torch.from_numpy(X).cuda()
    kw = dict(include_scale=False, include_2pi=True)
    got = fit.batched_mll(F, noise, None, Xd, y, ft, chunk=256, **kw)  # a non-PD pivot (info != 0) raises LinAlgError
    again = fit.batched_mll(F, noise, None, Xd, y, ft, chunk=256, **kw)
    assert np.array_equal(got, again)
    piped = fit.batched_mll(F, noise, None, Xd, y, ft, chunk=64, **kw)
    assert np.allclose(got, piped, rtol=1e-12, atol=0.0)
    # 16 of the 256 samples against the reference's own arithmetic (LU inv + slogdet, ~3 s of host time each): the first sample
    # of every even group of 16 and the last of every odd one — both ends of the batch, every XCD residue (b % 8 in {0, 7}).
    # The other 240 are checked against the vendor's Cholesky in the next test, not against the oracle.
    pick = [16 * c + (0 if c % 2 == 0 else 15) for c in range(B // 16)]
    assert pick[0] == 0 and pick[-1] == B - 1 and len(pick) == 16
    want = orc.batched_mll(F[pick], noise[pick], None, X, y, ft, **kw)
    assert np.allclose(got[pick], want, rtol=MLL_RTOL, atol=MLL_ATOL), (got[pick], want)
    assert np.isfinite(got).all()


def test_c3_b256_all_samples_against_vendor_cholesky():
    """The headline shape again, ALL 256 samples against a solver that shares nothing with the sweep (the oracle takes ~3 s per
    sample, hence sixteen of them in the test above; the other 240 have this check only): the device Gram matrices (bit-exact against the oracle in
    test_gpu_parity.py) through torch.linalg.cholesky, the vendor's batched fp64 factorisation.  A checker, like the oracle:
    nothing in the product calls it."""
    import torch

    import bark_amd.fitting as fit
    import bark_amd.forest as bf
    from bark_amd import synthetic as syn

    N, B, m = 4096, 256, 50
    X, y, bounds, ft = syn.unit_cube_problem(N, 8, seed=N)
    F = syn.sample_prior_forests(B, m, bounds, ft, seed=N)
    noise = np.random.default_rng(N).uniform(0.05, 0.15, B)
    Xd = torch.from_numpy(X).cuda()
    got = fit.batched_mll(F, noise, None, Xd, y, ft, chunk=256, include_scale=False, include_2pi=True)
    yd = torch.from_numpy(np.asarray(y, dtype=np.float64).reshape(-1)).cuda()
    eye = torch.eye(N, dtype=torch.float64, device="cuda")
    ref = np.empty(B)
    for c0 in range(0, B, 16):
        K = bf.batched_forest_gram_matrix(F[c0:c0 + 16], Xd, Xd, ft)
        K += (1e-6 + torch.from_numpy(noise[c0:c0 + 16]).cuda())[:, None, None] * eye
        try:
            L = torch.linalg.cholesky(K)
        except RuntimeError as e:  # a torch build without a batched potrf
            pytest.skip(f"torch.linalg.cholesky unavailable on this box: {e}")
        z = torch.linalg.solve_triangular(L, yd.expand(L.shape[0], N)[:, :, None], upper=False)[:, :, 0]
        logdet = 2.0 * L.diagonal(dim1=1, dim2=2).log().sum(1)
        ref[c0:c0 + 16] = (0.5 * (-(z * z).sum(1) - logdet - N * np.log(2.0 * np.pi))).cpu().numpy()
        del K, L, z
    assert np.allclose(got, ref, rtol=MLL_RTOL, atol=MLL_ATOL), np.abs(got / ref - 1.0).max()


def test_n2200_b256_one_chunk_of_256():
    """A second shape of the plain schedule's class (18 block rows >= 16, 256 resident matrices), ragged N, mixed feature
    types, scale included; the chunk = 96 call is the pipelined schedule on the same forests."""
    import bark_amd.fitting as fit
    from bark_amd import synthetic as syn
    from oracle import oracle as orc

    N, B, m = 2200, 256, 50
    X, y, bounds, ft = syn.mixed_problem(N, seed=2200)
    F = syn.sample_prior_forests(B, m, bounds, ft, seed=2201)
    rng = np.random.default_rng(2202)
    noise, scale = rng.uniform(0.05, 0.3, B), rng.uniform(0.6, 1.5, B)
    kw = dict(include_scale=True, include_2pi=False)
    got = fit.batched_mll(F, noise, scale, X, y, ft, chunk=256, **kw)
    assert np.array_equal(got, fit.batched_mll(F, noise, scale, X, y, ft, chunk=256, **kw))
    assert np.allclose(got, fit.batched_mll(F, noise, scale, X, y, ft, chunk=96, **kw), rtol=1e-12, atol=0.0)
    pick = [0, 1, 127, 128, 255]
    want = orc.batched_mll(F[pick], noise[pick], scale[pick], X, y, ft, **kw)
    assert np.allclose(got[pick], want, rtol=MLL_RTOL, atol=MLL_ATOL), (got[pick], want)


def test_sixteen_block_rows_b256_is_the_boundary_of_the_paired_schedule():
    """chunks of exactly 16 block rows (N = 1921..2048) x a multiple of 256 matrices sit ON the boundary of the schedule
    rule (plan_chunk: paired from PLAIN_MIN_NRB = 16 block rows on, pipelined below) and no test covered them.  N = 2048 and
    the ragged N = 1930 (16 block rows, the last one of 10 points) at B = 256: bit-reproducible, equal to the pipelined schedule
    (chunk = 64) to 1e-12, five samples against the oracle's LU route; N = 1920 x 256 (15 block rows) stays pipelined."""
    import bark_amd.fitting as fit
    from bark_amd import synthetic as syn
    from bark_amd.fitting import schedule_plan
    from oracle import oracle as orc

    assert schedule_plan(2048, 256)["schedule"] == "paired" and schedule_plan(1930, 256)["schedule"] == "paired"
    assert schedule_plan(1920, 256)["schedule"] == "pipelined"
    B, m = 256, 50
    for N in (2048, 1930):
        X, y, bounds, ft = syn.unit_cube_problem(N, 8, seed=N)
        F = syn.sample_prior_forests(B, m, bounds, ft, seed=N + 1)
        noise = np.random.default_rng(N + 2).uniform(0.05, 0.15, B)
        kw = dict(include_scale=False, include_2pi=True)
        got = fit.batched_mll(F, noise, None, X, y, ft, chunk=256, **kw)
        assert np.array_equal(got, fit.batched_mll(F, noise, None, X, y, ft, chunk=256, **kw))
        assert np.allclose(got, fit.batched_mll(F, noise, None, X, y, ft, chunk=64, **kw), rtol=1e-12, atol=0.0)
        pick = [0, 7, 128, 200, 255]
        want = orc.batched_mll(F[pick], noise[pick], None, X, y, ft, **kw)
        assert np.allclose(got[pick], want, rtol=MLL_RTOL, atol=MLL_ATOL), (N, got[pick], want)


@pytest.mark.parametrize("noise", [1e-1, 1e-4, 1e-6])
def test_mll_tolerance_as_the_kernel_matrix_approaches_singularity(noise):
    """DESIGN §2 "Conditioning" under the driver's run (was tests/validate_conditioning.py, a script): K has numerical rank
    ~150, so cond(K_s) ~ N / (1e-6 + noise) — 6e3 at the benchmark's noise, 1e7 at 1e-4, 5e8 at 1e-6.  Stated tolerance: the HIP
    Cholesky sweep agrees with the reference's LU route (mcmc_record_mll.py:69-70) to rtol 1e-9 while cond(K_s) <= 1e5, and
    beyond that to within 30 x the disagreement between the CPU's own LU and Cholesky routes on the same matrix (floor 1e-9) and
    never worse than cond(K_s) x 2^-52 x 100 — the formulation (blocked factor, explicit 128 x 128 inverses) adds nothing to
    what the conditioning of the problem costs any solver."""
    import bark_amd.fitting as fit
    from bark_amd import synthetic as syn
    from oracle import oracle as orc

    N = 1024
    X, y, bounds, ft = syn.unit_cube_problem(N, 8, seed=N)
    F = syn.sample_prior_forests(1, 50, bounds, ft, seed=N)
    kw = dict(include_scale=False, include_2pi=True)
    got = fit.batched_mll(F, [noise], None, X, y, ft, **kw)[0]
    lu = orc.batched_mll(F, [noise], None, X, y, ft, **kw)[0]
    ch = orc.batched_mll(F, [noise], None, X, y, ft, cholesky=True, **kw)[0]
    K = orc.forest_gram_matrix(F[0], X, X, ft)
    ev = np.linalg.eigvalsh(K + (1e-6 + noise) * np.eye(N))
    cond = ev[-1] / ev[0]
    rel, cpu_rel = abs(got - lu) / abs(lu), abs(lu - ch) / abs(ch)
    if cond <= 1e5:
        assert rel <= MLL_RTOL, (noise, cond, rel)
    assert rel <= max(1e-9, 30 * cpu_rel), (noise, cond, rel, cpu_rel)
    assert rel <= 100 * cond * 2.0**-52, (noise, cond, rel)


def test_plain_schedule_n4200_b256():
    """A third shape of the plain schedule's class (33 block rows, 256 resident matrices), ragged N (4200 = 32 x 128 + 104:
    identity padding in the last block row), mixed feature types, scale included."""
    import bark_amd.fitting as fit
    from bark_amd import synthetic as syn
    from oracle import oracle as orc

    N, B, m = 4200, 256, 50
    X, y, bounds, ft = syn.mixed_problem(N, seed=4200)
    F = syn.sample_prior_forests(B, m, bounds, ft, seed=4201)
    rng = np.random.default_rng(4202)
    noise, scale = rng.uniform(0.05, 0.3, B), rng.uniform(0.6, 1.5, B)
    kw = dict(include_scale=True, include_2pi=False)
    got = fit.batched_mll(F, noise, scale, X, y, ft, chunk=256, **kw)
    assert np.array_equal(got, fit.batched_mll(F, noise, scale, X, y, ft, chunk=256, **kw))
    assert np.allclose(got, fit.batched_mll(F, noise, scale, X, y, ft, chunk=96, **kw), rtol=1e-12, atol=0.0)
    pick = [0, 255]
    want = orc.batched_mll(F[pick], noise[pick], scale[pick], X, y, ft, **kw)
    assert np.allclose(got[pick], want, rtol=MLL_RTOL, atol=MLL_ATOL), (got[pick], want)


def test_c5_b4_variant():
    """SURVEY §8d's B = 4 variant of configs[4]: four N = 16384 mixed-type matrices resident at once (the pipelined
    schedule with split-K launches instead of the lone matrix's).  Leaf indices of all four forests bit-exact, one
    MLL against a CPU Cholesky, forest 0 equal to the B = 1 call to rounding."""
    import scipy.linalg as sla
    import torch

    import bark_amd.fitting as fit
    import bark_amd.forest as bf
    from bark_amd import synthetic as syn
    from oracle import oracle as orc

    N, B = 16384, 4
    X, y, bounds, ft = syn.mixed_problem(N, seed=16384)
    F = syn.sample_prior_forests(B, 50, bounds, ft, seed=16384)
    noise, scale = np.array([0.1, 0.07, 0.13, 0.2]), np.array([1.0, 0.8, 1.3, 1.1])
    for b in range(B):
        assert np.array_equal(bf.pass_through_forest(F[b], X, ft), orc.pass_through_forest(F[b], X, ft))
    Xd = torch.from_numpy(X).cuda()
    kw = dict(include_scale=True, include_2pi=False)
    got = fit.batched_mll(F, noise, scale, Xd, y, ft, **kw)
    one = fit.batched_mll(F[:1], noise[:1], scale[:1], Xd, y, ft, **kw)
    assert np.allclose(got[:1], one, rtol=1e-12, atol=0.0)
    b = 2
    K = orc.forest_gram_matrix(F[b], X, X, ft)
    K *= scale[b]
    K[np.diag_indices(N)] += 1e-6 + noise[b]
    c = sla.cholesky(K, lower=True, check_finite=False, overwrite_a=True)
    z = sla.solve_triangular(c, y, lower=True, check_finite=False)
    mll0 = 0.5 * (-(z.T @ z)[0, 0] - 2.0 * np.log(np.diag(c)).sum())
    assert abs(got[b] - mll0) <= MLL_ATOL + MLL_RTOL * abs(mll0), (got[b], mll0)


def test_c5_n16384_mixed_posterior_10k_candidates():
    """configs[4]: N = 16384 mixed categorical + integer + continuous, fp64 blocked Cholesky (split-K rows, 128 block
    rows) + posterior predictive at 10^4 candidates, one forest.  Checker: CPU Cholesky of the oracle's Gram matrix
    (the LU inverse of a 16384^2 matrix takes minutes), plus the leaf-space closed form of the same posterior."""
    import scipy.linalg as sla
    import torch

    import bark_amd.fitting as fit
    import bark_amd.forest as bf
    import bark_amd.tree_kernels as tk
    from bark_amd import synthetic as syn
    from oracle import oracle as orc

    N, C = 16384, 10000
    X, y, bounds, ft = syn.mixed_problem(N, seed=16384)
    cand, _, _, _ = syn.mixed_problem(C, seed=16385)
    F = syn.sample_prior_forests(1, 50, bounds, ft, seed=16384)
    noise, scale = np.array([0.1]), np.array([1.0])
    leaves = bf.pass_through_forest(F[0], X, ft)
    assert np.array_equal(leaves, orc.pass_through_forest(F[0], X, ft))  # 16384 x 50 leaf indices, bit-exact
    Xd, cd = torch.from_numpy(X).cuda(), torch.from_numpy(cand).cuda()
    mu, var = tk.forest_predict((F, noise, scale), (Xd, y), cd, ft)
    mll = fit.batched_mll(F, noise, scale, Xd, y, ft, include_scale=True, include_2pi=False)
    mu, var = mu.cpu().numpy(), var.cpu().numpy()
    # oracle: Gram matrices from the C restatement, Cholesky route on the host
    K = orc.forest_gram_matrix(F[0], X, X, ft)
    Kg = bf.forest_gram_matrix(F[0], Xd, Xd, ft)
    assert np.array_equal(Kg.cpu().numpy(), K)  # 2.1 GB of Gram entries, bit-exact
    del Kg
    K *= scale[0]
    K[np.diag_indices(N)] += 1e-6 + noise[0]
    Kx = scale[0] * orc.forest_gram_matrix(F[0], cand, X, ft)
    c = sla.cholesky(K, lower=True, check_finite=False, overwrite_a=True)
    z = sla.solve_triangular(c, y, lower=True, check_finite=False)
    V = sla.solve_triangular(c, Kx.T, lower=True, check_finite=False)
    mu0, var0 = (V.T @ z).ravel(), scale[0] - (V * V).sum(0)
    mll0 = 0.5 * (-(z.T @ z)[0, 0] - 2.0 * np.log(np.diag(c)).sum())
    assert abs(mll[0] - mll0) <= MLL_ATOL + MLL_RTOL * abs(mll0), (mll[0], mll0)
    assert np.allclose(mu[0], mu0, rtol=1e-9, atol=1e-9), np.abs(mu[0] - mu0).max()
    assert np.allclose(var[0], var0, rtol=1e-9, atol=1e-9), np.abs(var[0] - var0).max()
    assert var.min() > 0.0
    # leaf-space closed form of the same posterior (no N x N matrix at all)
    mu_l, var_l = tk.forest_predict((F, noise, scale), (Xd, y), cd, ft, method="leafspace")
    assert np.allclose(mu_l.cpu().numpy(), mu, rtol=1e-8, atol=1e-9) and np.allclose(var_l.cpu().numpy(), var, rtol=1e-8, atol=1e-9)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _c4_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import bark_amd.fitting as fit
    from bark_amd import synthetic
    from bark_amd.distributed import gather_mll, shard_range

    total, N = 512, 4096
    X, y, bounds, ft = synthetic.unit_cube_problem(N, 8, seed=N)
    lo, hi = shard_range(total, rank, world)
    F = synthetic.sample_prior_forests(hi - lo, 50, bounds, ft, seed=N + lo)  # forest b is seeded by its global index
    noise = np.random.default_rng(7).uniform(0.05, 0.15, total)[lo:hi]
    local = fit.batched_mll(F, noise, None, X, y, ft, include_scale=False, include_2pi=True, return_device=True)
    full = gather_mll(local.cpu(), total)
    q.put((rank, full.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_c4_512_samples_sharded():
    """configs[3]: N = 4096, 512 forest samples in contiguous shards (forest.py:92-98 loop order).  Here 4 ranks of
    128 share the one GPU of the test box (gloo gathers host copies; on a node each rank has its own GPU and RCCL
    gathers device tensors): every rank ends with the same (512,) vector, equal (to rounding: 1e-12) to a call
    with a different batch size, and a sample of it agrees with the oracle."""
    import torch.multiprocessing as mp

    import bark_amd.fitting as fit
    from bark_amd import synthetic
    from oracle import oracle as orc

    world = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_c4_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=900) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for r in range(1, world):
        assert np.array_equal(results[0], results[r])
    total, N = 512, 4096
    X, y, bounds, ft = synthetic.unit_cube_problem(N, 8, seed=N)
    noise = np.random.default_rng(7).uniform(0.05, 0.15, total)
    pick = sorted(set([0, 1, 127, 128, 255, 256, 383, 511] + list(range(5, 512, 16))))  # shard edges + a spread: 40 forests,
    # enough resident matrices for the non-split-K schedule the 128-forest shards ran
    F = np.stack([synthetic.sample_prior_forests(1, 50, bounds, ft, seed=N + b)[0] for b in pick])
    one = fit.batched_mll(F, noise[pick], None, X, y, ft, include_scale=False, include_2pi=True)
    # the shard size decides which tiles of a step take the split-K route: equal to rounding, not bit for bit
    assert np.allclose(results[0][pick], one, rtol=1e-12, atol=0.0)
    want = orc.batched_mll(F[:1], noise[pick][:1], None, X, y, ft, include_scale=False, include_2pi=True)
    assert np.allclose(one[:1], want, rtol=MLL_RTOL, atol=MLL_ATOL)
    assert np.isfinite(results[0]).all()


def test_schedule_plan_is_what_the_sweep_runs():
    """the schedule table of DESIGN.md section 4 (asserted without a GPU in
    tests/test_host_cpu.py::test_schedule_table_of_the_baseline_configs) describes what is launched: the instrumented call's
    launch counts (bark_mll_timing) are those of the schedule bark_mll_plan_query names — the headline shape c3 runs `paired`
    (16 row launches for 32 block rows), c4's per-GPU share `pipelined` (a row launch per block row with tiles), c2 the split-K
    layout — and the instrumented call agrees with the production call's MLL."""
    import torch

    import bench
    from bark_amd import _lib
    from bark_amd.fitting import schedule_plan

    #                N     B   schedule     diag panel solve launches
    cases = ((4096, 256, "paired", 32, 16, 31), (4096, 64, "pipelined", 32, 31, 31), (1024, 1, "splitk", 8, 6, 7),
             (512, 256, "plain", 4, 3, 3))
    for N, B, sched, nd, npan, nsol in cases:
        assert schedule_plan(N, B, timing=True)["schedule"] == sched, (N, B)
        wl = bench.Workload(N, 8, 50, B, seed_base=N, rank_offset=0)
        wl.settle_device_wait()
        wl.run()
        torch.cuda.synchronize()
        want = wl.mll_d.clone()
        t = _lib.MllTiming()
        wl.run(timing=t)
        torch.cuda.synchronize()
        assert (t.n_diag_launches, t.n_panel_launches, t.n_solve_launches) == (nd, npan, nsol), (N, B, sched)
        assert torch.allclose(wl.mll_d, want, rtol=1e-12, atol=0.0) and int(wl.info_d.abs().max().item()) == 0
        del wl
        torch.cuda.empty_cache()
