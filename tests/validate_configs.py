#!/usr/bin/env python3
"""One-off measurement + validation of the BASELINE.json configs other than the bench headline (c3).
   c1: TreeFunction N=64 single forest (MLL value parity, us/eval)
   c2: N=1024 d=8 m=50 single forest
   c5: N=16384 mixed cat+int+cont, 10k-candidate posterior predictive, B=1 (oracle check: Cholesky route on CPU)
Not collected by pytest (no test_ prefix); run on the GPU box:  python tests/validate_configs.py [--skip-c5-oracle]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import bark_amd.fitting as fit, bark_amd.forest as bf, bark_amd.tree_kernels as tk
from bark_amd import synthetic as syn
from oracle import oracle as orc

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps

out = {}
# ---- c1
forest, leaf_values, f = syn.tree_function()
X = np.random.default_rng(64).uniform(size=(64, 5)); y = f(X).reshape(-1, 1); y = (y - y.mean()) / y.std()
ft = np.full(5, 2); prior = syn.sample_prior_forests(1, 50, np.tile([[0., 1.]], (5, 1)), ft, seed=64)
got = fit.batched_mll(prior, [0.1], [1.0], X, y, ft, include_scale=True, include_2pi=True)
want = orc.batched_mll(prior, [0.1], [1.0], X, y, ft, include_scale=True, include_2pi=True)
dt = timeit(lambda: fit.batched_mll(prior, [0.1], [1.0], X, y, ft, include_scale=True, include_2pi=True), 20)
t0 = time.perf_counter(); orc.batched_mll(prior, [0.1], [1.0], X, y, ft, include_scale=True, include_2pi=True); tc = time.perf_counter() - t0
out["c1"] = dict(N=64, mll_gpu=float(got[0]), mll_oracle=float(want[0]), abs_err=float(abs(got[0]-want[0])), api_ms_per_eval=dt*1e3, oracle_ms=tc*1e3)
# ---- c2
X, y, bounds, ft = syn.unit_cube_problem(1024, 8, seed=1024)
F = syn.sample_prior_forests(1, 50, bounds, ft, seed=1024); Xd = torch.from_numpy(X).cuda()
got = fit.batched_mll(F, [0.1], [1.0], Xd, y, ft, include_scale=True, include_2pi=True)
want = orc.batched_mll(F, [0.1], [1.0], X, y, ft, include_scale=True, include_2pi=True)
dt = timeit(lambda: fit.batched_mll(F, [0.1], [1.0], Xd, y, ft, include_scale=True, include_2pi=True, return_device=True), 20)
dg = timeit(lambda: bf.forest_gram_matrix(F[0], Xd, Xd, ft), 20)
t0 = time.perf_counter(); orc.batched_mll(F, [0.1], [1.0], X, y, ft, include_scale=True, include_2pi=True); tc = time.perf_counter() - t0
out["c2"] = dict(N=1024, rel_err=float(abs(got[0]-want[0])/abs(want[0])), api_ms_per_eval=dt*1e3, gram_api_ms=dg*1e3, oracle_ms=tc*1e3)
# ---- c5
N, C = 16384, 10000
X, y, bounds, ft = syn.mixed_problem(N, seed=16384); cand, _, _, _ = syn.mixed_problem(C, seed=16385)
F = syn.sample_prior_forests(1, 50, bounds, ft, seed=16384); noise, scale = np.array([0.1]), np.array([1.0])
Xd, cd = torch.from_numpy(X).cuda(), torch.from_numpy(cand).cuda()
t_pred = timeit(lambda: tk.forest_predict((F, noise, scale), (Xd, y), cd, ft), 3)
t_mll = timeit(lambda: fit.batched_mll(F, noise, scale, Xd, y, ft, include_scale=True, include_2pi=False, return_device=True), 3)
mu, var = tk.forest_predict((F, noise, scale), (Xd, y), cd, ft); mu, var = mu.cpu().numpy(), var.cpu().numpy()
mll = fit.batched_mll(F, noise, scale, Xd, y, ft, include_scale=True, include_2pi=False)
out["c5"] = dict(N=N, C=C, predict_s=t_pred, mll_s=t_mll, chol_tflops_mll=N**3/3/t_mll/1e12,
                 predict_tflops=(N**3/3 + N*N*C)/t_pred/1e12, var_min=float(var.min()), var_max=float(var.max()))
if "--skip-c5-oracle" not in sys.argv:
    import scipy.linalg as sla
    t0 = time.perf_counter()
    K = orc.forest_gram_matrix(F[0], X, X, ft); Ks = scale[0]*K + (1e-6+noise[0])*np.eye(N)
    Kx = scale[0]*orc.forest_gram_matrix(F[0], cand, X, ft)
    c = sla.cholesky(Ks, lower=True, check_finite=False); z = sla.solve_triangular(c, y, lower=True, check_finite=False)
    V = sla.solve_triangular(c, Kx.T, lower=True, check_finite=False)
    mu0 = (V.T @ z).ravel(); var0 = scale[0] - (V*V).sum(0); mll0 = 0.5*(-(z.T@z)[0,0] - 2*np.log(np.diag(c)).sum())
    out["c5"].update(oracle_chol_s=time.perf_counter()-t0, mu_max_abs_err=float(np.abs(mu[0]-mu0).max()),
                     var_max_abs_err=float(np.abs(var[0]-var0).max()), mll_rel_err=float(abs(mll[0]-mll0)/abs(mll0)))
print(json.dumps(out, indent=1))
