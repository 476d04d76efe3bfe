"""Graph-capture probe: each shape in its own process (a runtime crash must not hide the others).
   python tools/ab/capture_probe.py            -> runs every shape as a child
   python tools/ab/capture_probe.py N B        -> one shape"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) == 1:
    for shape in ("1024 1", "2000 1", "3000 1", "6900 1", "2100 2", "1100 16", "2100 8", "4096 16"):
        r = subprocess.run([sys.executable, __file__, *shape.split()], capture_output=True, text=True)
        print(shape, "rc", r.returncode, (r.stdout.strip().splitlines() or [""])[-1], flush=True)
        if r.returncode != 0:
            print("   ", r.stderr.strip().splitlines()[-3:], flush=True)
    sys.exit(0)
import torch
import bench
from bark_amd import _lib
N, B = int(sys.argv[1]), int(sys.argv[2])
wl = bench.Workload(N, 8, 50, B, seed_base=N, rank_offset=0)
wl.run(); torch.cuda.synchronize()
eager = wl.mll_d.clone()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    wl.stream = _lib.stream_ptr()
    wl.run()
wl.mll_d.zero_()
g.replay(); g.replay(); torch.cuda.synchronize()
print("ok" if bool((wl.mll_d == eager).all()) else "MISMATCH", float(wl.mll_d[0]))
