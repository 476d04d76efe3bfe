import ctypes, numpy as np, torch, sys
import bark_amd.fitting as fit
from bark_amd import synthetic as syn, _lib
N=4096
X, y, bounds, ft = syn.unit_cube_problem(N, 8, seed=1)
F = syn.sample_prior_forests(1, 50, bounds, ft, seed=1)
Xd=torch.from_numpy(X).cuda()
lib=_lib.lib()
for _ in range(2): fit.batched_mll(F,[0.1],[1.0],Xd,y,ft,include_scale=True,include_2pi=False)
torch.cuda.synchronize()
lib.bark_debug_stamps_clear()
fit.batched_mll(F,[0.1],[1.0],Xd,y,ft,include_scale=True,include_2pi=False)
torch.cuda.synchronize()
out=(ctypes.c_ulonglong*16)()
lib.bark_debug_stamps(out,16)
v=np.array(list(out),dtype=np.float64)
# stamps are from the LAST diag launch (j=31); cycle counter runs at 100 MHz? print raw diffs
d=np.diff(v[:6])
print("raw stamps", v[:6]); print("phase cycles: gemm+assemble, chol, inverse, Wout, z/acc:", d, "sum", d.sum()); print("factor16 accumulated over 32 launches x 8 calls:", v[8])
