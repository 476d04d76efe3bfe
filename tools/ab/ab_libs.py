#!/usr/bin/env python3
"""Same-box, same-process A/B of libbarkhip.so variants on the dense MLL sweep (rule: perf deltas come from
interleaved rounds in ONE process on ONE device).

    python tools/ab/ab_libs.py [--n 4096] [--batch 256] [--rounds 5] [--steps 2] [--cand 0] label=path.so [label=path.so ...]

Every variant is loaded side by side with ctypes (never copied over the product library), gets the same device
inputs and its own workspace view, and is timed with HIP events on the launch stream; rounds interleave the variants.
Prints median / min ms per call and checks that all variants agree on the MLL values (max relative difference).
Build variants with tools/ab/build_variant.sh.
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--trees", type=int, default=50)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--cand", type=int, default=0)
    ap.add_argument("--mixed", action="store_true")
    ap.add_argument("libs", nargs="+")
    a = ap.parse_args()

    import numpy as np
    import torch

    from bark_amd import _lib, synthetic
    from bark_amd.forest import PackedForest

    base = _lib.lib()  # product library: host packer + torch's HIP runtime in the process
    N, B, m, C = a.n, a.batch, a.trees, a.cand
    if a.mixed:
        X, y, bounds, ft = synthetic.mixed_problem(N, seed=N)
        cand = synthetic.mixed_problem(C, seed=N + 1)[0] if C else None
    else:
        X, y, bounds, ft = synthetic.unit_cube_problem(N, 8, seed=N)
        cand = np.random.default_rng(1).uniform(size=(C, 8)) if C else None
    d = X.shape[1]
    F = synthetic.sample_prior_forests(B, m, bounds, ft, seed=N)
    noise = np.random.default_rng(N).uniform(0.05, 0.15, size=B)
    pf = PackedForest(F, ft)
    Xd, yd, nd = _lib.to_device(X), _lib.to_device(y.reshape(-1)), _lib.to_device(noise)
    sd = _lib.to_device(np.ones(B)) if C else None
    cd = _lib.to_device(cand) if C else None
    dev = Xd.device
    flags = (_lib.MLL_INCLUDE_SCALE if C else _lib.MLL_INCLUDE_2PI)
    variants = []
    # ONE workspace shared by all variants (they run one after the other): where a 34 GB buffer lands in HBM moves the
    # timing by several percent, so per-variant buffers would bias the comparison
    handles = []
    for spec in a.libs:
        label, path = spec.split("=", 1)
        handles.append((label, ctypes.CDLL(os.path.abspath(path))))
    for _, h in handles:
        h.bark_mll_workspace_bytes.restype = ctypes.c_size_t
        h.bark_mll_workspace_bytes.argtypes = [ctypes.c_int64] * 4
    ws = torch.empty(max(int(h.bark_mll_workspace_bytes(N, C, m, B)) for _, h in handles), dtype=torch.uint8, device=dev)
    for spec in a.libs:
        label, path = spec.split("=", 1)
        h = ctypes.CDLL(os.path.abspath(path))
        for name, (res, args) in _lib.SIGNATURES.items():
            if hasattr(h, name):
                fn = getattr(h, name)
                fn.restype, fn.argtypes = res, args
        hctx = ctypes.c_void_p()
        assert h.bark_ctx_create(dev.index, ctypes.byref(hctx)) == 0
        out = torch.empty(B, dtype=torch.float64, device=dev)
        info = torch.empty(B, dtype=torch.int32, device=dev)
        mu = torch.empty((B, C), dtype=torch.float64, device=dev) if C else None
        var = torch.empty((B, C), dtype=torch.float64, device=dev) if C else None

        def run(h=h, hctx=hctx, ws=ws, out=out, info=info, mu=mu, var=var):
            rc = h.bark_mll_batched_hip(hctx, _lib.ptr(pf.packed), pf.info_ref, _lib.ptr(Xd), N, d, _lib.ptr(yd), _lib.ptr(nd),
                                        _lib.ptr(sd), None, flags, _lib.ptr(cd), C, _lib.ptr(out), _lib.ptr(mu), _lib.ptr(var),
                                        None, _lib.ptr(info), _lib.ptr(ws), ws.numel(), B, None, _lib.stream_ptr())
            assert rc == 0, h.bark_last_error()

        variants.append((label, run, out, info, []))
    for _, run, *_ in variants:  # warm-up (first launch of a code object, LDS attributes, streams)
        run()
    torch.cuda.synchronize()
    for _ in range(a.rounds):
        for label, run, out, info, times in variants:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.steps):
                run()
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) / a.steps)
    ref = variants[0][2].cpu().numpy()
    print(f"N={N} B={B} m={m} C={C}  rounds={a.rounds} x {a.steps} calls")
    for label, run, out, info, times in variants:
        ts = sorted(times)
        v = out.cpu().numpy()
        bad = int(info.abs().max().item())
        rel = float(np.max(np.abs(v - ref) / np.abs(ref)))
        tf = B * N**3 / 3.0 / (ts[len(ts) // 2] * 1e-3) / 1e12
        print(f"  {label:16s} median {ts[len(ts)//2]:9.3f} ms  min {ts[0]:9.3f} ms  ({tf:6.2f} TFLOP/s algorithmic)  "
              f"info={bad} max_rel_diff_vs_first={rel:.2e} bit_identical={bool((v == ref).all())}")


if __name__ == "__main__":
    main()
