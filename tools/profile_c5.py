#!/usr/bin/env python3
"""Profile target: one c5-shaped MLL sweep (N=16384 mixed, B=1), run under rocprofv3 --kernel-trace --stats."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bark_amd.fitting as fit
from bark_amd import synthetic as syn
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
X, y, bounds, ft = syn.mixed_problem(N, seed=16384)
F = syn.sample_prior_forests(1, 50, bounds, ft, seed=16384)
Xd = torch.from_numpy(X).cuda()
for _ in range(2):
    out = fit.batched_mll(F, np.array([0.1]), np.array([1.0]), Xd, y, ft, include_scale=True, include_2pi=False, return_device=True)
torch.cuda.synchronize()
print(out)
