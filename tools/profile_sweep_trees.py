"""Profile target: ChainBatch.sweep_trees (device-side Metropolis sweep over the trees) at N points, nc chains.
   python tools/profile_sweep_trees.py [N] [chains] [reps]     (under rocprofv3 --kernel-trace for the per-kernel view)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bark_amd.fitting as fit
from bark_amd import synthetic as syn

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
nc = int(sys.argv[2]) if len(sys.argv) > 2 else 1
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
m = 50
X, y, bounds, ft = syn.unit_cube_problem(N, 8, seed=N)
cur = syn.sample_prior_forests(nc, m, bounds, ft, seed=7000)
prop = syn.sample_prior_forests(nc, m, bounds, ft, seed=8000)
noise, scale = np.full(nc, 0.1), np.ones(nc)
rng = np.random.default_rng(5)
log_q, log_u = rng.normal(0.0, 0.5, size=(nc, m)), np.log(rng.uniform(size=(nc, m)))
Xd = torch.from_numpy(X).cuda()
for r in range(reps + 1):
    cb = fit.ChainBatch.from_forests(cur, noise, scale, Xd, y, ft)
    torch.cuda.synchronize()
    t = time.perf_counter()
    mask = cb.sweep_trees(cur, prop, log_q, log_u, Xd, ft, scale, m)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
print(f"N={N} chains={nc}: sweep of {m} trees {dt * 1e3:.3f} ms, {dt * 1e3 / m:.4f} ms per tree step, accepted {int(mask.sum())}")
