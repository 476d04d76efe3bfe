"""Single-forest leaf-space MLL (the noise/scale proposal of one chain): wall time, host-side share, kernel list.
Usage: python tools/profile_leafspace.py [N]   (PYTHONPATH=$PWD; optionally under rocprofv3 --kernel-trace)"""
import cProfile
import pstats
import sys
import time

import numpy as np
import torch

import bark_amd.fitting as fit
from bark_amd import synthetic as syn

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
X, y, bounds, ft = syn.unit_cube_problem(N, 8, seed=1)
forest = syn.sample_prior_forests(1, 50, bounds, ft, seed=1)
Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()
call = lambda: fit.batched_mll(forest, [0.1], [1.0], Xd, yd, ft, include_scale=True, include_2pi=False,  # noqa: E731
                               method="leafspace", return_device=True)
for _ in range(5):
    call()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(100):
    out = call()
torch.cuda.synchronize()
print(f"wall per call: {(time.perf_counter() - t) * 1e4:.1f} us")
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    call()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
