"""Cycle stamps of multi_block_kernel (tuning build: tools/ab/build_variant.sh twost -DBARK_TWO_STAMPS; thread 0 of matrix 0)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["BARK_LIB_PATH"] = os.path.abspath("tools/ab/twost.so")
import torch, bench
from bark_amd import _lib
_lib.SIGNATURES["bark_debug_mb_stamps"] = (ctypes.c_int, [ctypes.c_void_p])
N, B = int(sys.argv[1]), int(sys.argv[2])
wl = bench.Workload(N, 8, 50, B, seed_base=N, rank_offset=0)
for _ in range(3):
    wl.run()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
assert _lib.lib().bark_debug_mb_stamps(buf) == 0
t = list(buf)
nrb = (N + 127) // 128
names = ["start"] + [f"j={j} {w}" for j in range(nrb) for w in ("assembled", "factored", "z", "off-diagonal tiles")]
for i, n in enumerate(names):
    print(f"{n:26s} {t[i] - t[0]:9d} cyc (+{t[i] - t[max(i - 1, 0)]:7d})")
print("mb_offdiag (j = 1, c = 2), thread 0:", "GEMM K=128", t[41] - t[40], "| T = gen - sum", t[42] - t[41], "| product", t[43] - t[42], "| store + y", t[44] - t[43])
