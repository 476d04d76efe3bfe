"""Coarse phase profile of diag_kernel (tuning build: tools/ab/build_variant.sh <name> -DBARK_DIAG_STAMPS [...]): thread 0 of
workgroup 0, last launch of the sweep (j = nrb - 1).   python tools/ab/diag_stamps.py tools/ab/<name>.so [N] [B]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["BARK_LIB_PATH"] = os.path.abspath(sys.argv[1])
import torch, bench
from bark_amd import _lib
_lib.SIGNATURES["bark_debug_diag_stamps"] = (ctypes.c_int, [ctypes.c_void_p])
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
wl = bench.Workload(N, 8, 50, B, seed_base=N, rank_offset=0)
for _ in range(3):
    wl.run()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
assert _lib.lib().bark_debug_diag_stamps(buf) == 0
t = list(buf)
print(f"{sys.argv[1]}  N={N} B={B}")
names = {1: "tile assembled (+ barrier)", 2: "factor + inverse (+ barrier)", 3: "W written", 4: "(G)", 5: "z, sums (3 barriers)"}
prev = t[0]
for i in (1, 2, 3, 4, 5):
    print(f"{names[i]:30s} {t[i]-t[0]:8d} cyc  (+{t[i]-prev})")
    prev = t[i]
for kb in range(8):
    nxt = t[8 + kb + 1] if kb < 7 else t[2]
    f = t[24 + kb] - t[16 + kb] if kb < 7 else 0
    print(f"step {kb}: (B)+barrier {t[16+kb]-t[8+kb]:6d} | wave 0: update + factor16 {f:6d} | wait for the other waves {nxt - (t[24+kb] if kb < 7 else t[16+kb]):6d} | total {nxt-t[8+kb]:6d}")
