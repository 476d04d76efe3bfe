#!/usr/bin/env python3
"""Host-side profile of one small API call (N=1024, B=1): where do the milliseconds go?"""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bark_amd.fitting as fit
from bark_amd import synthetic as syn
X, y, bounds, ft = syn.unit_cube_problem(1024, 8, seed=1024)
F = syn.sample_prior_forests(1, 50, bounds, ft, seed=1024)
call = lambda: fit.batched_mll(F, [0.1], [1.0], X, y, ft, include_scale=True, include_2pi=True)
for _ in range(5): call()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(50): call()
torch.cuda.synchronize(); print("ms per call:", (time.perf_counter() - t) / 50 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(50): call()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
