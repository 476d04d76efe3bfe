"""Soak of multi_block_kernel: random shapes in its window against the multi-launch sweep (the instrumented call) to 1e-12, twice for reproducibility."""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from bark_amd import _lib
from bark_amd.fitting import schedule_plan
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
done = 0
while done < n:
    N = random.randint(257, 768)
    Bn = random.choice([80, 96, 112, 144, 150, 200, 256, 300, 512, 520])
    problem = random.choice(["unit", "unit", "stress", "mixed"])
    wl = bench.Workload(N, 8, random.choice([13, 20, 50]), Bn, seed_base=N + done, rank_offset=0, problem=problem)
    words = int(_lib.lib().bark_leaf_words(wl.pf.info_ref))
    if schedule_plan(N, Bn, leaf_words=words)["schedule"] != "multi_block":
        continue
    wl.run(); torch.cuda.synchronize()
    got = wl.mll_d.clone()
    assert int(wl.info_d.abs().max().item()) == 0, (N, Bn, problem)
    for _ in range(3):
        wl.run()
    torch.cuda.synchronize()
    assert bool((wl.mll_d == got).all()), ("not reproducible", N, Bn, problem)
    t = _lib.MllTiming()
    wl.run(timing=t); torch.cuda.synchronize()
    rel = float((wl.mll_d / got - 1).abs().max())
    assert rel < 1e-12, (N, Bn, problem, rel)
    done += 1
    print(done, N, Bn, problem, words, f"{rel:.1e}", flush=True)
print("soak ok")
