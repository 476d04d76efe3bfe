"""Phase profile of diag_kernel from s_memtime stamps (tuning build tools/ab/stamps.so, never the product)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["BARK_LIB_PATH"] = os.path.abspath("tools/ab/stamps.so")
import torch, bench
from bark_amd import _lib
wl = bench.Workload(4096, 8, 50, 1, seed_base=4096, rank_offset=0)
for _ in range(3):
    wl.run()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 32)()
assert _lib.lib().bark_debug_stamps(buf) == 0
t = list(buf)
names = {0: "start", 2: "rank-128 update + assembly done", 7: "factor16(0) done", 3: "pipelined factor+inverse loop done", 4: "last X column stored",
         5: "W written", 6: "z / sums done"}
prev = t[0]
for i in (2, 7, 3, 4, 5, 6):
    print(f"{names[i]:36s} {t[i]-t[0]:8d} cyc  (+{t[i]-prev})")
    prev = t[i]
print("step kb=3: (B)+barrier", t[11] - t[10], "| wave0 C+factor16", t[12] - t[11], "| wave1 work", t[15] - t[14], "| closing barrier (wave 0)", t[13] - t[12])
