"""Soak of the device-side hand-over (include/bark_hip.h, bark_device_wait): many back-to-back sweeps of chain-bound shapes
— lone matrices, few matrices, with and without candidates, two host threads on their own streams — each compared bit for
bit with the same call under event joins, and every info vector checked for -3 (a timed-out wait).
   python tests/validate_device_wait_soak.py [seconds]        (GPU box; prints one summary line per shape)"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from bark_amd import _lib  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
lib = _lib.lib()
shapes = [(1024, 1, 0), (1500, 1, 0), (2100, 2, 0), (4096, 1, 0), (2100, 6, 0), (1100, 24, 0), (4096, 8, 0), (3000, 3, 0),
          (6900, 1, 0), (1400, 3, 300), (2000, 1, 3000), (4096, 16, 0), (8192, 2, 0)]
per_shape = budget / len(shapes)
total_calls = total_bad = 0
for N, B, C in shapes:
    wl = bench.Workload(N, 8, 50, B, seed_base=N, rank_offset=0, C=C, problem="mixed" if C else "unit")
    lib.bark_device_wait(0)
    wl.run()
    torch.cuda.synchronize()
    ref = wl.mll_d.clone()
    ref_mu = wl.mu_d.clone() if C else None
    lib.bark_device_wait(1)
    t0, calls, bad, mismatch = time.perf_counter(), 0, 0, 0
    while time.perf_counter() - t0 < per_shape:
        for _ in range(8):  # a burst without host synchronisation in between
            wl.run()
        torch.cuda.synchronize()
        calls += 8
        info = wl.info_d.cpu().numpy()
        bad += int((info != 0).sum())
        if not bool((wl.mll_d == ref).all()) or (C and not bool((wl.mu_d == ref_mu).all())):
            mismatch += 1
    total_calls += calls
    total_bad += bad + mismatch
    print(f"N={N:5d} B={B:2d} C={C:5d}: {calls:6d} calls, info != 0: {bad}, bursts differing from the event-join result: {mismatch}", flush=True)

# two host threads, each with its own context and stream, hammering small chain-bound shapes at once
errors = []


def worker(k):
    try:
        torch.cuda.set_device(0)
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            wl = bench.Workload(1200 + 300 * k, 8, 50, 1 + k, seed_base=77 + k, rank_offset=0)
            wl.stream = _lib.stream_ptr()
            wl.run()
            st.synchronize()
            ref = wl.mll_d.clone()
            for _ in range(300):
                wl.run()
            st.synchronize()
            if not bool((wl.mll_d == ref).all()) or int(wl.info_d.abs().max().item()) != 0:
                errors.append((k, "mismatch or info"))
        _lib.release_ctx()
    except Exception as exc:  # pragma: no cover
        errors.append((k, repr(exc)))


ts = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
for t in ts:
    t.start()
for t in ts:
    t.join()
print(f"two threads x 300 calls: {'ok' if not errors else errors}", flush=True)
print(f"TOTAL {total_calls} calls, failures {total_bad + len(errors)}")
sys.exit(1 if (total_bad or errors) else 0)
