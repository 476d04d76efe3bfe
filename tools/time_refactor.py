"""Steady-state cost of the sampler's noise/scale step for ONE chain (bark_sampler.py:267-272): full rebuild of
K_inv + logdet (dense), MLL only (dense), MLL only (leaf space).  Usage: python tools/time_refactor.py [N]"""
import sys
import time

import numpy as np
import torch

import bark_amd.fitting as fit
from bark_amd import synthetic as syn

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = 50
X, y, bounds, ft = syn.unit_cube_problem(N, 8, seed=1)
forest = syn.sample_prior_forests(1, m, bounds, ft, seed=1)
Xd, yd = torch.from_numpy(X).cuda(), torch.from_numpy(y).cuda()


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3, out


ms, _ = timed(lambda: fit.batched_kernel_inverse(forest, [0.1], [1.0], Xd, yd, ft, no_null=False, return_device=True))
print(f"N={N}: dense K_inv + logdet rebuild      {ms:8.3f} ms")
ms, a = timed(lambda: fit.batched_mll(forest, [0.1], [1.0], Xd, yd, ft, include_scale=True, include_2pi=False))
print(f"N={N}: dense MLL only                    {ms:8.3f} ms   {float(a[0]):.9f}")
ms, b = timed(lambda: fit.batched_mll(forest, [0.1], [1.0], Xd, yd, ft, include_scale=True, include_2pi=False,
                                      method="leafspace"))
print(f"N={N}: leaf-space MLL only               {ms:8.3f} ms   {float(b[0]):.9f}")
ms, _ = timed(lambda: fit.batched_kernel_inverse(forest, [0.1], [1.0], Xd, yd, ft, no_null=False, return_device=True,
                                                 method="leafspace"))
print(f"N={N}: leaf-space K_inv + logdet rebuild {ms:8.3f} ms")
