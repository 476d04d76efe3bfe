"""One-off randomised check of the device-side Metropolis sweep (ChainBatch.sweep_trees) against the host loop and the oracle
(not collected by pytest): python3 tests/fuzz/fuzz_sweep.py from the repo root, on a GPU box."""
import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import bark_amd.fitting as fit
from bark_amd import synthetic as syn
from oracle import oracle as orc
rng = np.random.default_rng(3)
for case in range(6):
    N = int(rng.choice([96, 255, 256, 640, 1001]))
    nc = int(rng.integers(1, 6)); m = int(rng.integers(2, 7))
    d = 6; ft = np.full(d, 2)
    X = rng.uniform(size=(N, d)); y = rng.standard_normal((N, 1))
    depth = int(rng.integers(1, 5))  # 2..16 leaves per tree
    cur = np.stack([syn.full_binary_forest(m, d, depth, rng, node_limit=40) for _ in range(nc)])
    prop = np.stack([syn.full_binary_forest(m, d, int(rng.integers(1, 5)), rng, node_limit=40) for _ in range(nc)])
    noise, scale = rng.uniform(0.05, 0.2, nc), rng.uniform(0.8, 1.2, nc)
    lq = rng.normal(0, 0.5, (nc, m)); lu = np.log(rng.uniform(size=(nc, m)))
    host = fit.ChainBatch.from_forests(cur, noise, scale, X, y, ft)
    want = np.zeros((nc, m), bool)
    for t in range(m):
        before = host.mll.copy()
        vals = host.propose_trees(cur[:, t], prop[:, t], X, ft, scale, m)
        want[:, t] = lu[:, t] <= np.minimum(lq[:, t] + (vals - before), 0.0)
        host.accept(want[:, t])
    dev = fit.ChainBatch.from_forests(cur, noise, scale, X, y, ft)
    mask = dev.sweep_trees(cur, prop, lq, lu, X, ft, scale, m)
    final = cur.copy(); final[mask] = prop[mask]
    ref = orc.batched_mll(final, noise, scale, X, y, ft, include_scale=True, include_2pi=False)
    ok = np.array_equal(mask, want) and np.allclose(dev.mll, ref, rtol=1e-8, atol=1e-7) and bool((dev.K_inv == host.K_inv).all())
    print(f"N={N} nc={nc} m={m} depth={depth} accepted={int(mask.sum())}/{mask.size} ok={ok}", flush=True)
    assert ok
print("all ok")
