#!/usr/bin/env python3
"""Gram kernel alone (forest.py:78-98 API path): 16 forests at N = 4096, full fp64 output -> ms and TB/s of writes.
   python3 tools/time_gram.py [N] [B]        (library variant via $BARK_LIB_PATH)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from bark_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
wl = bench.Workload(N, 8, 50, B, seed_base=N, rank_offset=0)
lib = wl.lib
leaves = torch.empty((B, int(lib.bark_leaf_words(wl.pf.info_ref)), int(lib.bark_leaf_npad(N))), dtype=torch.int32, device="cuda")
K = torch.empty((B, N, N), dtype=torch.float64, device="cuda")
_lib.check(lib.bark_leaf_codes_hip(_lib.ctx(), _lib.ptr(wl.pf.packed), wl.pf.info_ref, _lib.ptr(wl.Xd), N, wl.d, _lib.ptr(leaves), wl.stream))
def run():
    _lib.check(lib.bark_gram_from_leaves_hip(_lib.ptr(leaves), N, _lib.ptr(leaves), N, wl.pf.info_ref, None, None, None, _lib.ptr(K), N, N * N, wl.stream))
run(); torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8): run()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 8)
ms = sorted(ts)[2]
print(f"gram N={N} B={B}: {ms:.4f} ms  {B*8.0*N*N/ms/1e9:.3f} TB/s  sym={bool((K[0]==K[0].T).all())} diag1={bool((K[0].diagonal()==1).all())} lib={os.environ.get('BARK_LIB_PATH','product')}")
