#!/usr/bin/env python3
"""Profile target: the dense MLL sweep through the C ABI (production path) — run under rocprofv3.
   python3 tools/profile_mll.py [N] [B] [steps] [C]      (library variant via $BARK_LIB_PATH)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
C = int(sys.argv[4]) if len(sys.argv) > 4 else 0
wl = bench.Workload(N, 8, 50, B, seed_base=N, rank_offset=0, C=C, problem="mixed" if C else "unit")
avg, med = wl.device_ms(steps, warm=1)
digest = ""
if not os.environ.get("BARK_PROFILE_NOCHECK"):  # timing-only ablation builds produce finite garbage
    import hashlib

    digest = " mll=" + hashlib.sha256(wl.check().tobytes()).hexdigest()[:10]  # bit-identity of variants across processes
print(f"N={N} B={B} C={C}: {med:.3f} ms per call (median of {steps}), lib={os.environ.get('BARK_LIB_PATH', 'product')}{digest}")
