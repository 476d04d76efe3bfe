"""Coarse phase profile of diag_kernel (tuning build tools/ab/stamps2.so from tools/ab/make_stamps2.py, never the product):
thread 0 of workgroup 0, last launch of the sweep (j = nrb - 1).   python tools/ab/stamps2.py [N]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["BARK_LIB_PATH"] = os.path.abspath("tools/ab/stamps2.so")
import torch, bench
from bark_amd import _lib
_lib.SIGNATURES["bark_debug_stamps"] = (ctypes.c_int, [ctypes.c_void_p])
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
wl = bench.Workload(N, 8, 50, 1, seed_base=N, rank_offset=0)
for _ in range(3):
    wl.run()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
assert _lib.lib().bark_debug_stamps(buf) == 0
t = list(buf)
names = {1: "tile assembled (+ barrier)", 2: "factor16(0) + barrier", 3: "loop of 8 sub-block steps", 4: "last X column + barrier", 5: "W written",
         6: "(G)", 7: "z, sums (3 barriers)"}
prev = t[0]
for i in (1, 2, 3, 4, 5, 6, 7):
    print(f"{names[i]:30s} {t[i]-t[0]:8d} cyc  (+{t[i]-prev})")
    prev = t[i]
for kb in range(8):
    nxt = t[8 + kb + 1] if kb < 7 else t[3]
    f = t[24 + kb] - t[16 + kb] if kb < 7 else 0
    print(f"step {kb}: (B)+barrier {t[16+kb]-t[8+kb]:6d} | wave 0: update + factor16 {f:6d} | wait for the other waves {nxt - (t[24+kb] if kb < 7 else t[16+kb]):6d} | total {nxt-t[8+kb]:6d}")
