"""Driver for profiling the per-tree sampler step (ChainState.propose / accept) at N points.
Usage: python tools/profile_swap.py [N] [reps]   (run with PYTHONPATH=$PWD, optionally under rocprofv3)"""
import sys
import time

import numpy as np
import torch

import bark_amd.fitting as fit
import bark_amd.forest as bf
from bark_amd import synthetic as syn

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
m, scale, noise = 50, 1.0, 0.1
X, y, bounds, ft = syn.unit_cube_problem(N, 8, seed=1)
forest = syn.sample_prior_forests(1, m, bounds, ft, seed=1)[0]
fresh = syn.sample_prior_forests(1, m, bounds, ft, seed=2)[0]
Xd = torch.from_numpy(X).cuda()
state = fit.ChainState.from_forest(forest, noise, scale, Xd, y, ft)
s = np.sqrt(scale / m)
cur = bf.get_leaf_vectors(forest[0], Xd, ft) * s
new = bf.get_leaf_vectors(fresh[0], Xd, ft) * s
state.propose(cur, new)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(reps):
    state.propose(cur, new)
torch.cuda.synchronize()
print(f"propose            : {(time.perf_counter() - t) / reps * 1e3:.3f} ms  (r = {cur.shape[1]} + {new.shape[1]})")
t = time.perf_counter()
for _ in range(reps):
    state.propose_tree(forest[0], fresh[0], Xd, ft, scale, m)
torch.cuda.synchronize()
print(f"propose_tree       : {(time.perf_counter() - t) / reps * 1e3:.3f} ms")
state.propose(cur, new)  # warm-up of the accept path (first launch of its kernels)
state.accept()
state.propose(new, cur)
state.accept()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(reps):
    state.propose(cur, new)
    state.accept()
    state.propose(new, cur)
    state.accept()
torch.cuda.synchronize()
print(f"propose + accept   : {(time.perf_counter() - t) / (2 * reps) * 1e3:.3f} ms")

# several chains in one call (each chain's sequence on its own stream)
nc = 4
forests = syn.sample_prior_forests(nc, m, bounds, ft, seed=11)
others = syn.sample_prior_forests(nc, m, bounds, ft, seed=12)
batch = fit.ChainBatch.from_forests(forests, np.full(nc, noise), np.full(nc, scale), Xd, y, ft)
batch.propose_trees(forests[:, 0], others[:, 0], Xd, ft, np.full(nc, scale), m)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(reps):
    batch.propose_trees(forests[:, 0], others[:, 0], Xd, ft, np.full(nc, scale), m)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / reps * 1e3
print(f"propose_trees x{nc}   : {dt:.3f} ms per call = {dt / nc:.3f} ms per chain")
