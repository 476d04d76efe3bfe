#!/usr/bin/env python3
"""One-off randomised hunt over shapes of the dense sweep (not part of the test-suite): python3 tests/fuzz/fuzz_shapes.py [n] [seed]; it checks against the oracle, hence under tests/."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_fuzz import test_mll_and_posterior_across_schedules as check
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
t0 = time.time()
for k in range(n):
    N = int(rng.choice([rng.integers(1, 300), rng.integers(300, 1400), rng.integers(1400, 2600)]))
    B = int(rng.choice([rng.integers(1, 10), rng.integers(10, 80), rng.integers(80, 400)]))
    if N > 1400: B = min(B, 60)
    m = int(rng.integers(1, 61))
    C = int(rng.choice([0, 0, rng.integers(1, 400)]))
    chunk = None if rng.uniform() < 0.6 else int(rng.integers(1, B + 1))
    check(N, B, m, C, chunk)
    print(f"[{time.time()-t0:6.1f}s] ok N={N} B={B} m={m} C={C} chunk={chunk}", flush=True)
print("all ok")
