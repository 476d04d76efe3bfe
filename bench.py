#!/usr/bin/env python3
"""bench.py — forest-Gram + Cholesky MLL evaluations per second (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path over one rank's batch of synthetic forest samples:
leaf traversal -> N x N Gram (+ jitter) -> blocked fp64 Cholesky -> triangular solve -> log-det
-> MLL, for every forest sample (one "eval" each), followed by the only cross-rank exchange of the
path, the all-gather of the (B,) log-likelihoods over RCCL.  Inputs (X, y, packed forests, noise)
are resident in HBM before the timed region; the output is the (B,) MLL vector on the device.

Workload (BASELINE.json configs[2], SURVEY §8d "c3"): N=4096 points, d=8 continuous features,
m=50 trees, B=256 forest samples PER GPU drawn from the BART depth prior (alpha .95, beta 2),
noise_b ~ U[0.05,0.15), scale 1, MLL convention of examples/mcmc/mcmc_record_mll.py:57-74.
Weak scaling: every rank evaluates its own 256 samples (c4's sharding, at c3's per-GPU batch).
"""

from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F64_MFMA_PEAK_TFLOPS = 78.6  # MI355X dense fp64 matrix peak (256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz)
HBM_PEAK_GBS = 8000.0


def hbm_traffic_from_profile():
    """HBM bytes per step of the Cholesky launch sequence, from the committed rocprofv3 PMC passes of this
    same command (profiles/rNN/*hbm_counters*.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc
    runs, KiB units, FETCH doubled as MI355X_MICROARCH.md prescribes for 16 B/lane streaming reads on gfx950).
    bench.py cannot collect PMC counters itself; None when no profile is committed."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "*hbm_counters*.json")))
    if not files:
        return None
    try:
        prof = json.load(open(files[-1]))
        total = 0.0
        for k in ("diag_kernel", "panel_kernel", "solve_kernel"):
            total += 2.0 * prof["FETCH_SIZE"][k]["sum_KiB"] * 1024.0 + prof["WRITE_SIZE"][k]["sum_KiB"] * 1024.0
        return {"bytes_per_step": total, "source": os.path.relpath(files[-1], ROOT)}
    except Exception:
        return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=4096, help="training points")
    ap.add_argument("--batch", type=int, default=256, help="forest samples per GPU")
    ap.add_argument("--trees", type=int, default=50)
    ap.add_argument("--dim", type=int, default=8)
    ap.add_argument("--cpu-sample", type=int, default=3, help="forest samples timed on the host oracle (0 = skip)")
    ap.add_argument("--chunk", type=int, default=0, help="forests factorised concurrently (0 = fit HBM)")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from bark_amd import _lib, synthetic
    from bark_amd.distributed import gather_mll
    from bark_amd.fitting.mll import choose_chunk
    from bark_amd.forest import PackedForest

    N, B, m, d = args.n, args.batch, args.trees, args.dim
    # ---- synthetic inputs: same X, y on every rank; rank r owns forests r*B .. r*B+B-1
    X, y, bounds, ft = synthetic.unit_cube_problem(N, d, seed=N)
    forests = synthetic.sample_prior_forests(B, m, bounds, ft, seed=N + rank * B)
    noise = np.random.default_rng(N + 7919 * (rank + 1)).uniform(0.05, 0.15, size=B)

    lib = _lib.lib()
    pf = PackedForest(forests, ft)  # host format conversion + upload: outside the timed region
    Xd = _lib.to_device(X)
    yd = _lib.to_device(y.reshape(-1))
    noise_d = _lib.to_device(noise)
    mll_d = torch.empty(B, dtype=torch.float64, device=Xd.device)
    info_d = torch.empty(B, dtype=torch.int32, device=Xd.device)
    Bc = args.chunk or choose_chunk(B, N, 0, m)
    ws = _lib.workspace(int(lib.bark_mll_workspace_bytes(N, 0, m, Bc)))
    flags = _lib.MLL_INCLUDE_2PI
    stream = _lib.stream_ptr()
    timing = _lib.MllTiming()
    tsum = dict(total_ms=0.0, gram_ms=0.0, chol_ms=0.0, diag_ms=0.0, panel_ms=0.0, solve_ms=0.0)

    def step(timed: bool):
        _lib.check(lib.bark_mll_batched_hip(
            _lib.ptr(pf.packed), pf.info_ref, _lib.ptr(Xd), N, d, _lib.ptr(yd), _lib.ptr(noise_d), None, None, flags,
            None, 0, _lib.ptr(mll_d), None, None, None, _lib.ptr(info_d), _lib.ptr(ws), ws.numel(), Bc,
            ctypes.byref(timing) if timed else None, stream))
        if timed:  # HIP-event spans of this step's launches, recorded on the launch stream
            for k in tsum:
                tsum[k] += getattr(timing, k)
        return gather_mll(mll_d, B * world) if world > 1 else mll_d

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        all_mll = step(True)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=Xd.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    nocheck = os.environ.get("BARK_BENCH_NOCHECK") == "1"  # tuning aid for timing-only ablation builds
    assert nocheck or int(info_d.abs().max().item()) == 0, "a kernel matrix was not positive definite"
    mll_host = all_mll.cpu().numpy()
    assert nocheck or np.isfinite(mll_host).all()

    # ---- the standalone Gram kernel (forest.py:78-98 API path), HBM-write bound: measured outside the timed
    # region on 16 forests, full N x N fp64 output each (in the MLL sweep the Gram is generated inside the
    # panel kernel and never written, so the sweep itself has no Gram stage to price)
    gram_probe = None
    if rank == 0:
        Bg = min(16, B)
        sub = _lib.PackInfo.from_buffer_copy(pf.info)
        sub.B = Bg
        leaves = torch.empty((Bg, int(lib.bark_leaf_words(ctypes.byref(sub))), int(lib.bark_leaf_npad(N))),
                             dtype=torch.int32, device=Xd.device)
        Kg = torch.empty((Bg, N, N), dtype=torch.float64, device=Xd.device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 8
        for it in range(reps + 1):
            if it == 1:
                e0.record()
            _lib.check(lib.bark_leaf_codes_hip(_lib.ptr(pf.packed), ctypes.byref(sub), _lib.ptr(Xd), N, d,
                                               _lib.ptr(leaves), stream))
            _lib.check(lib.bark_gram_from_leaves_hip(_lib.ptr(leaves), N, _lib.ptr(leaves), N, ctypes.byref(sub), None,
                                                     None, None, _lib.ptr(Kg), N, N * N, stream))
        e1.record()
        torch.cuda.synchronize()
        g_ms = e0.elapsed_time(e1) / reps
        g_bytes = Bg * (8.0 * N * N + 4.0 * m * 2 * N)  # SURVEY §8d bytes_gram per matrix
        gram_probe = {"bound": "hbm", "forests": Bg, "algorithmic_bytes": g_bytes, "ms": g_ms,
                      "leaf_code": "one-hot bits" if lib.bark_leaf_encoding(ctypes.byref(sub)) == 1 else "packed bytes",
                      "leaf_code_words": int(lib.bark_leaf_words(ctypes.byref(sub))),
                      "achieved_GBs": g_bytes / (g_ms * 1e-3) / 1e9, "peak_GBs": HBM_PEAK_GBS,
                      "frac": g_bytes / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        del Kg, leaves

    # ---- informational only: the leaf-space evaluation of the SAME MLLs (R x R system over the leaves instead
    # of the N x N matrix).  It does not do the Gram + Cholesky work the metric counts and is not part of `value`.
    leaf_probe = None
    if rank == 0:
        lws = torch.empty(int(lib.bark_mll_leafspace_workspace_bytes(N, int(pf.info.max_bits), m, B, 0)), dtype=torch.uint8,
                          device=Xd.device)
        lmll = torch.empty(B, dtype=torch.float64, device=Xd.device)
        linfo = torch.empty(B, dtype=torch.int32, device=Xd.device)
        l0, l1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        lreps = 5
        for it in range(lreps + 1):
            if it == 1:
                l0.record()
            _lib.check(lib.bark_mll_leafspace_hip(_lib.ptr(pf.packed), pf.info_ref, _lib.ptr(Xd), N, d, _lib.ptr(yd),
                                                  _lib.ptr(noise_d), None, flags, None, 0, _lib.ptr(lmll), None, None,
                                                  _lib.ptr(linfo), _lib.ptr(lws), lws.numel(), B, stream))
        l1.record()
        torch.cuda.synchronize()
        l_ms = l0.elapsed_time(l1) / lreps
        dense_local = mll_d.cpu().numpy()
        leaf_local = lmll.cpu().numpy()
        leaf_probe = {"note": "exact Woodbury/determinant-lemma evaluation over the forest's leaves; NOT the benchmark "
                              "metric (no N x N Gram, no N^3/3 Cholesky)",
                      "leaves_per_forest_max": int(pf.info.max_bits), "ms_per_%d_evals" % B: l_ms,
                      "evals_per_s": B / (l_ms * 1e-3),
                      "max_rel_diff_vs_dense": float(np.max(np.abs(leaf_local - dense_local) / np.abs(dense_local)))}
        del lws

    # one chain of the sampler on this workload's data: per-tree proposal, accept, noise/scale
    # proposal and the rebuild of the resident inverse — secondary numbers, never `value`
    chain_probe = None
    if rank == 0:
        from bark_amd.fitting import ChainState

        def wall_ms(fn, reps):
            fn()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / reps * 1e3

        chain = ChainState.from_forest(forests[0], float(noise[0]), 1.0, Xd, y, ft)
        old_tree, new_tree = forests[0][0], forests[-1][-1]  # any two trees of this rank's forests

        def swap_there_and_back():
            chain.propose_tree(old_tree, new_tree, Xd, ft, 1.0, m)
            chain.accept()
            chain.propose_tree(new_tree, old_tree, Xd, ft, 1.0, m)
            chain.accept()

        def rebuild():
            chain.propose_noise_scale(forests[0], float(noise[0]), 1.0, Xd, ft)
            chain.accept()

        t_prop = wall_ms(lambda: chain.propose_tree(old_tree, new_tree, Xd, ft, 1.0, m), 50)
        t_pair = wall_ms(swap_there_and_back, 25) / 2
        t_ns = wall_ms(lambda: chain.propose_noise_scale(forests[0], float(noise[0]), 1.0, Xd, ft), 20)
        t_rebuild = wall_ms(rebuild, 10)
        chain_probe = {"note": "one chain, N=%d, m=%d, wall time per call incl. host (bark_sampler.py:226-272)" % (N, m),
                       "tree_proposal_ms": t_prop, "tree_proposal_plus_accept_ms": t_pair,
                       "noise_scale_proposal_ms": t_ns, "noise_scale_proposal_plus_rebuild_ms": t_rebuild}
        del chain

    steps = args.steps
    evals = B * world * steps
    value = evals / elapsed
    per_step = {k: v / steps for k, v in tsum.items()}
    chol_flops = B * N**3 / 3.0  # algorithmic flops of one launch sequence (SURVEY §8d flops_chol x B)
    chol_tflops = chol_flops / (per_step["chol_ms"] * 1e-3) / 1e12
    traffic = hbm_traffic_from_profile()
    result = {
        "metric": "forest-Gram + Cholesky MLL evals/sec at N=4096, 50 trees",
        "value": value,
        "unit": "evals/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "c3: N=%d d=%d m=%d, %d prior forest samples per GPU (BASELINE configs[2]; "
                        "sharding of configs[3]), noise U[0.05,0.15), mcmc_record_mll convention" % (N, d, m, B),
            "N": N, "d": d, "trees": m, "forests_per_gpu": B, "chunk": Bc, "parallelism": "samples/%d" % world,
        },
        "roofline": {
            "bound": "mfma",
            "kernel": "Cholesky launch sequence per step (diag_kernel + panel_kernel + solve_kernel; panel_kernel "
                      "dominates). diag_kernel(j) runs beside panel_kernel(j) on a helper stream, so the per-kernel "
                      "event spans below overlap and do not add up to chol_ms",
            "achieved": chol_tflops,
            "peak": F64_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": chol_tflops / F64_MFMA_PEAK_TFLOPS,
            "traffic": traffic["bytes_per_step"] if traffic else None,
            "traffic_unit": "bytes of HBM traffic per step (one Cholesky launch sequence, 256 evals)",
            "traffic_source": traffic["source"] if traffic else None,
            "algorithmic_flops_per_step": chol_flops,
            "ms_per_step": {k: round(v, 3) for k, v in per_step.items()},
            "panel_kernel": {
                "launches_per_step": int(timing.n_panel_launches),
                "avg_ms": per_step["panel_ms"] / max(int(timing.n_panel_launches), 1),
                "executed_tflops": timing.panel_flops / (max(per_step["panel_ms"], 1e-9) * 1e-3) / 1e12,
            },
            "solve_kernel": {
                "launches_per_step": int(timing.n_solve_launches),
                "avg_ms": per_step["solve_ms"] / max(int(timing.n_solve_launches), 1),
                "executed_tflops": timing.solve_flops / (max(per_step["solve_ms"], 1e-9) * 1e-3) / 1e12,
            },
            "diag_kernel": {
                "launches_per_step": int(timing.n_diag_launches),
                "avg_ms": per_step["diag_ms"] / max(int(timing.n_diag_launches), 1),
            },
            "gram_stage_ms_per_step": round(per_step["gram_ms"], 3),
            "gram_kernel": gram_probe,
        },
        "leafspace_probe": leaf_probe,
        "chain_step_probe": chain_probe,
    }

    if rank == 0 and world == 1 and args.cpu_sample > 0:
        # the oracle (checker) timed on this box's host cores on a bounded sample of the same workload
        from oracle import oracle as orc

        ns = min(args.cpu_sample, B)
        t1 = time.perf_counter()
        ref = orc.batched_mll(forests[:ns], noise[:ns], None, X, y, ft, include_scale=False, include_2pi=True)
        cpu_s = time.perf_counter() - t1
        t2 = time.perf_counter()
        orc.batched_mll(forests[:ns], noise[:ns], None, X, y, ft, include_scale=False, include_2pi=True, cholesky=True)
        chol_s = time.perf_counter() - t2
        rel = float(np.max(np.abs(mll_host[:ns] - ref) / np.abs(ref)))
        assert np.allclose(mll_host[:ns], ref, rtol=1e-9, atol=1e-8), (mll_host[:ns], ref)
        result["cpu_baseline"] = {
            "value": ns / cpu_s,
            "unit": "evals/s",
            "cores": len(os.sched_getaffinity(0)),
            "kind": "port",
            "sample": "%d of the %d forest samples of this workload (N=%d): C leaf walk + N*N*m compare-count Gram "
                      "(1 thread, as the reference) + numpy.linalg.inv + slogdet (LAPACK threads = cores), %.1f s"
                      % (ns, B, N, cpu_s),
            "cholesky_variant_evals_per_s": ns / chol_s,
            "gpu_vs_oracle_max_rel_err": rel,
        }
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()  # rank 0 ran the untimed probes above; leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
