#!/usr/bin/env python3
"""One-off (not collected): MLL agreement between the HIP Cholesky sweep and the oracle's LU route as the
noise shrinks (cond(K_s) ~ N * scale / (1e-6 + noise))."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bark_amd.fitting as fit
from bark_amd import synthetic as syn
from oracle import oracle as orc
import scipy.linalg as sla
for N in (1024, 2048):
    X, y, bounds, ft = syn.unit_cube_problem(N, 8, seed=N)
    F = syn.sample_prior_forests(1, 50, bounds, ft, seed=N)
    K = orc.forest_gram_matrix(F[0], X, X, ft)
    for noise in (1e-1, 1e-2, 1e-3, 1e-4, 1e-6, 0.0):
        got = fit.batched_mll(F, [noise], None, X, y, ft, include_scale=False, include_2pi=True)[0]
        lu = orc.batched_mll(F, [noise], None, X, y, ft, include_scale=False, include_2pi=True)[0]
        ch = orc.batched_mll(F, [noise], None, X, y, ft, include_scale=False, include_2pi=True, cholesky=True)[0]
        Ks = K + (1e-6 + noise) * np.eye(N)
        ev = np.linalg.eigvalsh(Ks); cond = ev[-1] / ev[0]
        print(f"N={N} noise={noise:g} cond={cond:.2e}  gpu={got:.10e}  rel(gpu,LU)={abs(got-lu)/abs(lu):.1e}  rel(gpu,cpuChol)={abs(got-ch)/abs(ch):.1e}  rel(LU,cpuChol)={abs(lu-ch)/abs(ch):.1e}")
