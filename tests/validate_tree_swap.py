#!/usr/bin/env python3
"""One-off timing of the sampler's per-tree step at N = 4096 (not collected by pytest):
fused ChainState.propose/accept vs the drop-in quick_inverse chain on device vs the CPU oracle."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bark_amd.fitting as fit, bark_amd.forest as bf
from bark_amd import synthetic as syn
from oracle import oracle as orc
qi = fit.quick_inverse
N, m, scale, noise = 4096, 50, 1.0, 0.1
X, y, bounds, ft = syn.unit_cube_problem(N, 8, seed=1)
forest = syn.sample_prior_forests(1, m, bounds, ft, seed=1)[0]
fresh = syn.sample_prior_forests(1, m, bounds, ft, seed=2)[0]
Xd = torch.from_numpy(X).cuda(); yd = torch.from_numpy(y).cuda()
t0 = time.perf_counter(); state = fit.ChainState.from_forest(forest, noise, scale, Xd, y, ft); torch.cuda.synchronize()
print("init (Gram + Cholesky + explicit inverse) s:", round(time.perf_counter() - t0, 4))
s = np.sqrt(scale / m)
def leafv(nodes): return bf.get_leaf_vectors(nodes, Xd, ft) * s
cur, new = leafv(forest[0]), leafv(fresh[0]); torch.cuda.synchronize()
def fused():
    v = state.propose(cur, new); return v
def chain():
    i1 = qi.low_rank_inv_update(state.K_inv, cur, subtract=True, assume_symmetric=True)
    d1 = qi.low_rank_det_update(state.K_inv, cur, state.logdet, subtract=True)
    i2 = qi.low_rank_inv_update(i1, new, assume_symmetric=True)
    d2 = qi.low_rank_det_update(i1, new, d1)
    return float(qi.mll(i2, d2, yd))
for name, fn in (("fused propose", fused), ("drop-in chain on device", chain)):
    fn(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): v = fn()
    torch.cuda.synchronize(); print(name, "ms:", round((time.perf_counter() - t) / 20 * 1e3, 3), "mll", v)
t = time.perf_counter(); state.propose(cur, new); state.accept(); torch.cuda.synchronize()
print("propose + accept ms:", round((time.perf_counter() - t) * 1e3, 3))
Kc = state.K_inv.cpu().numpy(); cu, nw = cur.cpu().numpy(), new.cpu().numpy(); ld = state.logdet
t = time.perf_counter()
i1 = orc.low_rank_inv_update(Kc, cu, subtract=True); d1 = orc.low_rank_det_update(Kc, cu, ld, subtract=True)
i2 = orc.low_rank_inv_update(i1, nw); d2 = orc.low_rank_det_update(i1, nw, d1); v = orc.mll(i2, d2, y)
print("CPU oracle chain ms:", round((time.perf_counter() - t) * 1e3, 1))
