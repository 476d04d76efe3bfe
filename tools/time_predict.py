"""Wall time of forest_predict, dense vs leaf-space (device-resident points; forests on the host as always).
Usage: python tools/time_predict.py [N] [C] [B]"""
import sys
import time

import numpy as np
import torch

import bark_amd.synthetic as syn
import bark_amd.tree_kernels as tk

N, C, B = (int(v) for v in (sys.argv[1:4] + ["16384", "10000", "1"][len(sys.argv) - 1:]))
X, y, bounds, ft = syn.unit_cube_problem(N, 8, seed=1)
cand = np.random.default_rng(2).random((C, 8))
F = syn.sample_prior_forests(B, 50, bounds, ft, seed=3)
model = (F, np.full(B, 0.1), np.full(B, 1.0))
Xd, yd, cd = (torch.as_tensor(v, device="cuda") for v in (X, y, cand))
res = {}
for method in ("dense", "leafspace"):
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        mu, var = tk.forest_predict(model, (Xd, yd), cd, ft, method=method)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    res[method] = (dt, mu.cpu().numpy() if hasattr(mu, "cpu") else mu, var.cpu().numpy() if hasattr(var, "cpu") else var)
    print(f"{method:10s} N={N} C={C} B={B}: {dt * 1e3:9.2f} ms")
print("max |mu diff|", np.abs(res["dense"][1] - res["leafspace"][1]).max(), " max |var diff|",
      np.abs(res["dense"][2] - res["leafspace"][2]).max())
