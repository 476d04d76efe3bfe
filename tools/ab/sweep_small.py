"""Tree sweep (ChainBatch.sweep_trees) at small N: wall against device time, and where the host time goes.
   python3 tools/ab/sweep_small.py [N ...]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bark_amd.fitting as fit
from bark_amd import synthetic as syn

m = 50
for N in [int(a) for a in sys.argv[1:]] or [128, 256, 512, 1000]:
    X, y, bounds, ft = syn.unit_cube_problem(N, 8, seed=N)
    cur = syn.sample_prior_forests(1, m, bounds, ft, seed=7000)
    prop = syn.sample_prior_forests(1, m, bounds, ft, seed=8000)
    noise, scale = np.full(1, 0.1), np.ones(1)
    rng = np.random.default_rng(5)
    log_q, log_u = rng.normal(0.0, 0.5, size=(1, m)), np.log(rng.uniform(size=(1, m)))
    Xd = torch.from_numpy(X).cuda()
    res = []
    for r in range(6):
        cb = fit.ChainBatch.from_forests(cur, noise, scale, Xd, y, ft)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t = time.perf_counter()
        e0.record()
        cb.sweep_trees(cur, prop, log_q, log_u, Xd, ft, scale, m)
        t_enq = time.perf_counter() - t
        e1.record()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t, e0.elapsed_time(e1) * 1e-3, t_enq))
    wall, dev, enq = sorted(res)[len(res) // 2]
    print(f"N={N}: wall {wall*1e3:.3f} ms, device {dev*1e3:.3f} ms, host returns after {enq*1e3:.3f} ms  ({wall*1e6/m:.1f} us per tree step)", flush=True)
cb = fit.ChainBatch.from_forests(cur, noise, scale, Xd, y, ft)
pr = cProfile.Profile()
pr.enable()
cb.sweep_trees(cur, prop, log_q, log_u, Xd, ft, scale, m)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
