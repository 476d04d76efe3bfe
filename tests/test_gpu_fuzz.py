"""Seeded sweep over shapes of the dense MLL / posterior entry point: batch sizes and matrix sizes that land in every
schedule of Sweep::step — split-K layout with and without look-ahead, filled chunks in the plain (ragged last round
split over K) and the pipelined schedule, chunks smaller than the batch, candidate columns — each checked against the oracle's LU route (the reference's
arithmetic) on a few forests, and for reproducibility of the same call."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MLL_RTOL, MLL_ATOL = 1e-9, 1e-8

# (N, B, m, C, chunk) — N not a multiple of 128, batch sizes around the schedule thresholds
CASES = [
    (1, 1, 3, 0, None), (2, 5, 1, 0, None), (130, 300, 7, 0, None),
    # one block row (N <= 128): the single-launch evaluation (chol.hip OneBlock), incl. chunks and the full 128
    (64, 256, 50, 0, None), (100, 7, 13, 0, None), (128, 40, 50, 0, 16), (20, 3, 50, 0, None), (257, 64, 50, 0, None), (513, 48, 20, 0, 16),
    # two block rows (128 < N <= 256): the single-launch evaluation by two_block_kernel, ragged second blocks, chunks, one matrix
    # (chunks of up to 384 matrices, more up to N = 224: plan_chunk; the others take the multi-launch sweep)
    (129, 5, 50, 0, None), (144, 1, 50, 0, None), (200, 256, 50, 0, None), (255, 33, 13, 0, None), (256, 64, 50, 0, 24), (210, 600, 20, 0, None),
    # three / four block rows (256 < N <= 512) in chunks of at least 80 / 112 matrices: multi_block_kernel; ragged last blocks, a chunked batch
    (300, 200, 50, 0, None), (400, 170, 13, 0, None), (512, 256, 50, 0, None), (511, 400, 20, 0, 200), (385, 160, 50, 0, None),
    # five / six block rows (512 < N <= 768) in chunks of at least 144 matrices that nearly fill whole rounds of the CUs: the same kernel
    (640, 160, 50, 0, None), (768, 256, 50, 0, None), (601, 150, 20, 0, None), (700, 520, 13, 0, 260),
    (700, 100, 13, 0, None), (777, 33, 50, 0, None), (900, 12, 50, 0, None), (1100, 64, 50, 0, None),
    (1300, 40, 30, 0, 24), (1500, 9, 50, 0, None), (2100, 2, 50, 0, None), (2500, 20, 50, 0, None),
    (640, 16, 50, 90, None), (1000, 70, 25, 130, None), (1400, 3, 50, 300, None),
    # filled chunks of >= 8 block rows whose size is not a multiple of 256: the pipelined schedule (Sweep::step_pipelined),
    # generated and materialised Gram, candidate columns, a shorter last chunk, and B = 256 at 9 block rows (pipelined as well: the plain schedule needs more
    # than 16 block rows beside the multiple of 256 — tests/test_gpu_configs.py::test_c3_b256_plain_schedule, ::test_plain_schedule_n2200_b256)
    (1100, 130, 50, 0, None), (1300, 200, 20, 0, 120), (2100, 70, 50, 0, None), (1000, 150, 25, 130, None),
    (1500, 100, 30, 200, None), (1100, 256, 50, 0, None),
    # split-K layout with look-ahead, its last step bound by the bulk (second bulk stream): one small matrix, many candidate
    # columns — (16 + 171 block columns) x 16 block rows stays below the layout threshold
    (2000, 1, 50, 21800, None),
]


@pytest.mark.parametrize("N,B,m,C,chunk", CASES)
def test_mll_and_posterior_across_schedules(N, B, m, C, chunk):
    import bark_amd.fitting as fit
    from bark_amd import _lib, synthetic as syn
    from bark_amd.fitting.mll import _run
    from oracle import oracle as orc

    seed = 1000 * N + B
    X, y, bounds, ft = syn.mixed_problem(N, seed=seed)
    F = syn.sample_prior_forests(B, m, bounds, ft, seed=seed + 1)
    rng = np.random.default_rng(seed)
    noise, scale = rng.uniform(0.05, 0.3, B), rng.uniform(0.6, 1.5, B)
    pick = sorted(set([0, B - 1, B // 2]))[:3]
    if C == 0:
        got = fit.batched_mll(F, noise, scale, X, y, ft, include_scale=True, include_2pi=True, chunk=chunk)
        again = fit.batched_mll(F, noise, scale, X, y, ft, include_scale=True, include_2pi=True, chunk=chunk)
        assert np.array_equal(got, again)  # fixed summation orders: reproducible for a given (B, chunk)
        want = orc.batched_mll(F[pick], noise[pick], scale[pick], X, y, ft, include_scale=True, include_2pi=True)
        assert np.allclose(got[pick], want, rtol=MLL_RTOL, atol=MLL_ATOL), (got[pick], want)
        return
    cand = syn.mixed_problem(C, seed=seed + 2)[0]
    mll, mu, var = _run(F, noise, scale, X, y, ft, _lib.MLL_INCLUDE_SCALE, cand=cand, chunk=chunk)
    mu, var, mll = mu.cpu().numpy(), var.cpu().numpy(), mll.cpu().numpy()
    mu0, var0 = orc.forest_predict((F[pick], noise[pick], scale[pick]), (X, y), cand, ft)
    assert np.allclose(mu[pick], mu0, rtol=1e-9, atol=1e-9) and np.allclose(var[pick], var0, rtol=1e-9, atol=1e-9)
    want = orc.batched_mll(F[pick], noise[pick], scale[pick], X, y, ft, include_scale=True, include_2pi=False)
    assert np.allclose(mll[pick], want, rtol=MLL_RTOL, atol=MLL_ATOL)


def test_inverse_export_through_the_pipelined_schedule():
    """`batched_kernel_inverse` (N identity columns appended: 18 block columns at N = 1100) with enough forests for the
    pipelined schedule, and a batch whose chunks take different schedules (256 = plain, the remaining 44 = pipelined)."""
    import bark_amd.fitting as fit
    from bark_amd import synthetic as syn
    from oracle import oracle as orc

    N, B, m = 1100, 40, 30
    X, y, bounds, ft = syn.mixed_problem(N, seed=77)
    F = syn.sample_prior_forests(B, m, bounds, ft, seed=78)
    rng = np.random.default_rng(79)
    noise, scale = rng.uniform(0.05, 0.3, B), rng.uniform(0.6, 1.5, B)
    K_inv, K_inv_y, logdet = fit.batched_kernel_inverse(F, noise, scale, X, y, ft, no_null=False)
    for b in (0, 17, 39):
        K = orc.forest_gram_matrix(F[b], X, X, ft)
        K_s = scale[b] * K + (1e-6 + noise[b]) * np.eye(N)
        assert np.allclose(K_inv[b] @ K_s, np.eye(N), atol=1e-8)
        assert np.allclose(K_inv_y[b], np.linalg.solve(K_s, y[:, 0]), rtol=1e-8, atol=1e-9)
        assert np.isclose(logdet[b], np.linalg.slogdet(K_s)[1], rtol=1e-10)

    B2 = 300
    F2 = syn.sample_prior_forests(B2, m, bounds, ft, seed=80)
    nz = np.random.default_rng(81).uniform(0.05, 0.3, B2)
    whole = fit.batched_mll(F2, nz, None, X, y, ft, include_scale=False, include_2pi=True, chunk=256)
    parts = fit.batched_mll(F2, nz, None, X, y, ft, include_scale=False, include_2pi=True, chunk=100)
    assert np.allclose(whole, parts, rtol=1e-12, atol=0)
    pick = [0, 255, 256, 299]
    want = orc.batched_mll(F2[pick], nz[pick], None, X, y, ft, include_scale=False, include_2pi=True)
    assert np.allclose(whole[pick], want, rtol=MLL_RTOL, atol=MLL_ATOL)


def test_two_block_kernel_against_the_multi_launch_sweep():
    """128 < N <= 256 (round 5): `two_block_kernel` evaluates a chunk in one launch after the leaf walk.  Against the sweep it
    replaces — the instrumented call (`bark_mll_timing`) still takes the multi-launch plain schedule: leaf walk, Gram tile,
    right-hand side, diag, rows, solve, diag — to 1e-12 relative (the right-hand-side update sums in another order), against the
    oracle's LU route, and with the bushy forests whose leaf codes are bytes (13 code words per point instead of 5)."""
    import torch

    import bench
    from bark_amd import _lib
    from bark_amd.fitting import schedule_plan
    from oracle import oracle as orc

    assert schedule_plan(200, 64)["schedule"] == "two_block" and schedule_plan(200, 64, timing=True)["schedule"] == "plain"
    assert schedule_plan(200, 64, leaf_words=84)["schedule"] == "plain"  # codes of 256 points no longer fit behind the factor image
    assert schedule_plan(256, 1)["schedule"] == "two_block" and schedule_plan(256, 512)["schedule"] == "plain"  # (where the sweep is faster)
    assert schedule_plan(200, 512)["schedule"] == "two_block" and schedule_plan(256, 300, chunk=256)["last_schedule"] == "two_block"
    for N, Bn, problem in ((129, 1, "unit"), (177, 40, "unit"), (256, 256, "unit"), (200, 17, "stress"), (240, 3, "mixed")):
        wl = bench.Workload(N, 8, 50, Bn, seed_base=N, rank_offset=0, problem=problem)
        wl.run()
        torch.cuda.synchronize()
        got = wl.mll_d.clone()
        assert int(wl.info_d.abs().max().item()) == 0
        wl.run()
        torch.cuda.synchronize()
        assert bool((wl.mll_d == got).all())  # reproducible
        t = _lib.MllTiming()
        wl.run(timing=t)
        torch.cuda.synchronize()
        assert (t.n_diag_launches, t.n_panel_launches, t.n_solve_launches) == (2, 1, 1)  # the sweep it replaces
        assert torch.allclose(wl.mll_d, got, rtol=1e-12, atol=0.0), (N, Bn, float((wl.mll_d / got - 1).abs().max()))
        pick = sorted({0, Bn // 2, Bn - 1})
        want = orc.batched_mll(wl.forests[pick], wl.noise[pick], None, wl.X, wl.y, wl.ft, include_scale=False, include_2pi=True)
        assert np.allclose(got.cpu().numpy()[pick], want, rtol=MLL_RTOL, atol=MLL_ATOL), (N, Bn)


def test_multi_block_kernel_against_the_multi_launch_sweep():
    """256 < N <= 768 (round 5): `multi_block_kernel` evaluates a chunk of at least 80 (three block rows) / 112 (four) / 144 (five, six) matrices in one launch after the leaf walk —
    three to six block rows, the off-diagonal GEMMs in the same workgroup.  Against the sweep it replaces (the instrumented call takes
    the multi-launch schedule) to 1e-12 relative, against the oracle's LU route, with byte codes (bushy forests), and the index of a
    non-positive pivot in the second, third and fourth block."""
    import re

    import torch

    import bark_amd.fitting as fit
    import bench
    from bark_amd import _lib, synthetic as syn
    from bark_amd.fitting import schedule_plan
    from oracle import oracle as orc

    assert schedule_plan(512, 256)["schedule"] == "multi_block" and schedule_plan(300, 160)["schedule"] == "multi_block"
    assert schedule_plan(512, 64)["schedule"] != "multi_block" and schedule_plan(512, 400)["schedule"] == "multi_block"
    assert schedule_plan(384, 64)["schedule"] != "multi_block" and schedule_plan(384, 80)["schedule"] == "multi_block"
    # five / six block rows: from 144 matrices, and only chunks that fill at least 80 % of their rounds of 256 CUs
    assert schedule_plan(512, 300)["schedule"] != "multi_block" and schedule_plan(512, 360)["schedule"] == "multi_block"  # (70 % of two rounds)
    assert schedule_plan(768, 128)["schedule"] != "multi_block" and schedule_plan(768, 144)["schedule"] == "multi_block"
    assert schedule_plan(640, 384)["schedule"] != "multi_block" and schedule_plan(640, 512)["schedule"] == "multi_block"
    assert schedule_plan(769, 256)["schedule"] != "multi_block" and schedule_plan(768, 256, leaf_words=7)["schedule"] != "multi_block"
    assert schedule_plan(512, 256, timing=True)["schedule"] == "plain" and schedule_plan(512, 256, leaf_words=40)["schedule"] == "plain"
    # (… and the ends of the window: the smallest chunks that take the kernel, one of more matrices than the chip has CUs)
    for N, Bn, problem in ((257, 160, "unit"), (384, 200, "unit"), (512, 256, "unit"), (380, 161, "stress"), (500, 170, "mixed"),
                           (512, 112, "unit"), (300, 80, "unit"), (300, 600, "unit"), (640, 144, "unit"), (768, 256, "unit"), (700, 150, "unit")):
        wl = bench.Workload(N, 8, 50, Bn, seed_base=N, rank_offset=0, problem=problem)
        words = int(_lib.lib().bark_leaf_words(wl.pf.info_ref))  # 5 for prior forests, 13 (bytes) for the bushy ones: three block rows only
        assert schedule_plan(N, Bn, leaf_words=words)["schedule"] == "multi_block", (N, Bn, words)
        wl.run()
        torch.cuda.synchronize()
        got = wl.mll_d.clone()
        assert int(wl.info_d.abs().max().item()) == 0
        wl.run()
        torch.cuda.synchronize()
        assert bool((wl.mll_d == got).all())  # reproducible
        t = _lib.MllTiming()
        wl.run(timing=t)
        torch.cuda.synchronize()
        assert t.n_diag_launches == (N + 127) // 128  # the sweep it replaces
        assert torch.allclose(wl.mll_d, got, rtol=1e-12, atol=0.0), (N, Bn, float((wl.mll_d / got - 1).abs().max()))
        pick = sorted({0, Bn // 2, Bn - 1})
        want = orc.batched_mll(wl.forests[pick], wl.noise[pick], None, wl.X, wl.y, wl.ft, include_scale=False, include_2pi=True)
        assert np.allclose(got.cpu().numpy()[pick], want, rtol=MLL_RTOL, atol=MLL_ATOL), (N, Bn)
    # K - eps I: the elimination fails where the forest's rank runs out, or at a duplicated point (tests/test_gpu_parity.py::
    # test_not_positive_definite_reports_the_first_bad_pivot has the construction): pivot 150 (second block) and 264 (third block);
    # 160 copies of the forest -> multi_block_kernel.  (A fourth-block failure needs a forest of rank > 384, i.e. more leaf-code
    # words than fit beside four blocks' LDS: such forests take the sweep, where the parity suite covers the index.)
    from bark_amd.forest import PackedForest

    X1, y1, bounds1, ft1 = syn.mixed_problem(300, seed=300)
    X1 = X1.copy()
    X1[149] = X1[0]
    F1 = syn.sample_prior_forests(1, 170, bounds1, ft1, seed=300)  # 170 trees: ~14 code words, the most that fit beside three blocks
    X2, y2, _b2, ft2 = syn.unit_cube_problem(384, 8, seed=384)
    F2 = syn.full_binary_forests(1, 100, 8, 2, np.random.default_rng(384))
    for X, y, ft, F, eps, expect in ((X1, y1, ft1, F1, 1e-3, 150), (X2, y2, ft2, F2, 1e-4, 264)):
        N = X.shape[0]
        noise = -1e-6 - eps
        A, want = orc.forest_gram_matrix(F[0], X, X, ft) + (1e-6 + noise) * np.eye(N), 0
        for k in range(N):
            if not A[k, k] > 0.0:
                want = k + 1
                assert A[k, k] < -1e-4, A[k, k]
                break
            A[k + 1:, k + 1:] -= np.outer(A[k + 1:, k], A[k, k + 1:]) / A[k, k]
        assert want == expect
        Frep = np.repeat(F, 160, axis=0)
        words = int(_lib.lib().bark_leaf_words(PackedForest(Frep, ft).info_ref))
        assert schedule_plan(N, 160, m=F.shape[1], leaf_words=words)["schedule"] == "multi_block", words
        with pytest.raises(np.linalg.LinAlgError, match="not positive definite") as exc:
            fit.batched_mll(Frep, np.full(160, noise), None, X, y, ft, include_scale=False, include_2pi=True)
        assert int(re.search(r"pivot (\d+)", str(exc.value)).group(1)) == want, (str(exc.value), want)
