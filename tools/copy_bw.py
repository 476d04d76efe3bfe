#!/usr/bin/env python3
"""HBM read + write bandwidth of a plain device-to-device copy (what a perfectly streaming solve kernel — read T, write U — could reach)."""
import torch
n = 1 << 30  # doubles: 8 GiB per buffer
a = torch.empty(n, dtype=torch.float64, device="cuda").normal_()
b = torch.empty_like(a)
for label, fn, bytes_moved in (("copy (read + write)", lambda: b.copy_(a), 16 * n), ("fill (write only)", lambda: b.fill_(1.5), 8 * n),
                               ("sum (read only)", lambda: a.sum(), 8 * n)):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{label:22s} {bytes_moved / ms / 1e9:7.2f} TB/s  ({ms:.2f} ms)")
