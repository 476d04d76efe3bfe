#!/usr/bin/env python3
"""Does the HBM-bound solve of one half of a chunk hide behind the MFMA-bound rows of the other half?
   Two host threads (own bark_ctx, own stream) each sweep B/2 forests at the same time; against one thread sweeping B.
   python3 tools/ab/two_stream_probe.py [N] [B] [calls] [stagger_ms]      (library variant via $BARK_LIB_PATH)"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import bench  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 4
stagger = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0

whole = bench.Workload(N, 8, 50, B, seed_base=N, rank_offset=0)
avg, med = whole.device_ms(calls, warm=1)
print(f"one stream, B={B}: {med:.3f} ms per call (median), {avg:.3f} avg", flush=True)
del whole
torch.cuda.empty_cache()

halves = [None, None]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
go = threading.Barrier(3)
done = threading.Barrier(3)


def worker(i):
    with torch.cuda.stream(streams[i]):
        wl = bench.Workload(N, 8, 50, B // 2, seed_base=N, rank_offset=i, chunk=B // 2)
        halves[i] = wl
        wl.run()
        streams[i].synchronize()
        for rnd in range(3):
            go.wait()
            if i == 1 and stagger:
                time.sleep(stagger * 1e-3)
            for _ in range(calls):
                wl.run()
            streams[i].synchronize()
            done.wait()


th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
for t in th:
    t.start()
for rnd in range(3):
    torch.cuda.synchronize()
    go.wait()
    t0 = time.perf_counter()
    done.wait()
    dt = (time.perf_counter() - t0) * 1e3 / calls
    print(f"two streams x B={B // 2}, stagger {stagger} ms: {dt:.3f} ms per {B} evaluations (wall, {calls} calls each)", flush=True)
for t in th:
    t.join()
halves[0].check()
halves[1].check()
print("lib =", os.environ.get("BARK_LIB_PATH", "product"))
