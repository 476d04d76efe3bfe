"""Phase profile of diag_kernel from cycle-counter stamps (tuning build tools/ab/stamps.so from tools/ab/make_stamps.py,
never the product).  Lone N = 4096 matrix; thread 0 of workgroup 0; the last launch of the sweep wins."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["BARK_LIB_PATH"] = os.path.abspath("tools/ab/stamps.so")
import torch, bench
from bark_amd import _lib
_lib.SIGNATURES["bark_debug_stamps"] = (ctypes.c_int, [ctypes.c_void_p])
wl = bench.Workload(4096, 8, 50, 1, seed_base=4096, rank_offset=0)
for _ in range(3):
    wl.run()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
assert _lib.lib().bark_debug_stamps(buf) == 0
t = list(buf)
names = {0: "start", 2: "rank-128 update + assembly done", 7: "factor16(0) done", 3: "pipelined factor+inverse loop done", 4: "logs / last X column stored",
         5: "W written", 6: "z / sums done"}
prev = t[0]
for i in (2, 7, 3, 4, 5, 6):
    print(f"{names[i]:36s} {t[i]-t[0]:8d} cyc  (+{t[i]-prev})")
    prev = t[i]
print("step kb=3: (B)+barrier", t[11] - t[10], "| wave0 link", t[16] - t[11], "| factor16", t[12] - t[16], "| closing barrier (wave 0)", t[13] - t[12])
print("last factor16 call: entry->LDL start", t[20] - t[19], "| LDL", t[21] - t[20], "| rows+MFMA (Q=0)", t[22] - t[21], "| Q=1", t[23] - t[22], "| Q=2", t[24] - t[23], "| Q=3", t[25] - t[24], "| tail (rsqrt, W store)", t[27] - t[26], "| total", t[27] - t[19])
