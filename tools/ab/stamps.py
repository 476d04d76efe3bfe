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
names = {0: "start", 1: "rank-128 update done", 2: "assembled D", 3: "blocked Cholesky done", 4: "block inverse done", 5: "W written", 6: "z / sums done"}
prev = t[0]
for i in range(1, 7):
    print(f"{names[i]:28s} {t[i]-t[0]:8d} cyc  (+{t[i]-prev})")
    prev = t[i]
print("factor16 (kb=3):", t[11] - t[10], "cyc; barrier after:", t[12] - t[11])
