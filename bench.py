#!/usr/bin/env python3
"""bench.py — forest-Gram + Cholesky MLL evaluations per second (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]          # plain shell: starts its own N workers
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
           bench.py --gpus N ...                                  # or under an external launcher

Launcher: without RANK/WORLD_SIZE in the environment this process never touches torch or the GPU; it starts
N fresh worker processes (one per GPU, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set) and relays rank 0's JSON line.
`--gpus 1` takes the same path.  Under torch.distributed.run the process IS a worker.

One step = one pass of the hot path over one rank's batch of synthetic forest samples:
leaf traversal -> N x N Gram (+ jitter) -> blocked fp64 Cholesky -> triangular solve -> log-det
-> MLL, for every forest sample (one "eval" each), followed by the only cross-rank exchange of the
path, the all-gather of the (B,) log-likelihoods over RCCL.  Inputs (X, y, packed forests, noise)
are resident in HBM before the timed region; the output is the (B,) MLL vector on the device.
The timed steps run the PRODUCTION path of the C ABI (no timing struct, no host synchronisation inside
the library); the per-kernel breakdown comes from one separate instrumented step after the timed region.

Workload (BASELINE.json configs[2], SURVEY §8d "c3"): N=4096 points, d=8 continuous features,
m=50 trees, B=256 forest samples PER GPU drawn from the BART depth prior (alpha .95, beta 2),
noise_b ~ U[0.05,0.15), scale 1, MLL convention of examples/mcmc/mcmc_record_mll.py:57-74.
Default = weak scaling (every rank evaluates its own 256 samples).  `--total B` = strong scaling: B samples
in all, contiguous shards per rank (configs[3] "c4" is `--total 512 --gpus 8` = 64 per GPU); in weak mode
with N > 1 the c4 split (512 / N per rank) is timed as a second region and reported under "c4_strong".
"""

from __future__ import annotations

import argparse
import ctypes
import hashlib
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F64_MFMA_PEAK_TFLOPS = 78.6  # MI355X dense fp64 matrix peak (256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz)
HBM_PEAK_GBS = 8000.0
METRIC = "forest-Gram + Cholesky MLL evals/sec at N=4096, 50 trees"


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=4096, help="training points")
    ap.add_argument("--batch", type=int, default=256, help="forest samples per GPU (weak scaling)")
    ap.add_argument("--total", type=int, default=0, help="forest samples in all, sharded over the GPUs (strong scaling)")
    ap.add_argument("--trees", type=int, default=50)
    ap.add_argument("--dim", type=int, default=8)
    ap.add_argument("--cpu-sample", type=int, default=3, help="forest samples timed on the host oracle (0 = skip)")
    ap.add_argument("--chunk", type=int, default=0, help="forests factorised concurrently (0 = fit HBM)")
    ap.add_argument("--no-configs", action="store_true", help="skip the secondary per-config measurements")
    ap.add_argument("--live-traffic", action=argparse.BooleanOptionalAction, default=True,
                    help="N = 1, launcher path only: after the worker has finished, the GPU-free parent runs the two rocprofv3 --pmc "
                         "passes (FETCH_SIZE, WRITE_SIZE) on one c3 sweep as child processes and puts the measured HBM bytes per step "
                         "into roofline.traffic (otherwise: the committed profile of this build, if any)")
    ap.add_argument("--selftest-launcher", action="store_true",
                    help="CPU-only check of the launcher / rendezvous / gather / timing protocol (gloo, no GPU work)")
    ap.add_argument("--require-rccl", action=argparse.BooleanOptionalAction, default=None,
                    help="with --gpus > 1 and the nccl backend: an RCCL init / probe failure ends the run with a non-zero exit "
                         "(default).  --no-require-rccl lets the 8-byte-per-sample MLL gather fall back to gloo instead; "
                         "the JSON then says so (collective_backend != collective_backend_requested)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------
# launcher (parent): no torch, no GPU
# ---------------------------------------------------------------------------------------------
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(args) -> int:
    n = args.gpus
    if n < 1:
        raise SystemExit("--gpus must be >= 1")
    port = _free_port()
    procs = []
    live = n == 1 and args.live_traffic and not args.selftest_launcher and shutil.which("rocprofv3") is not None
    line_file = None
    if live:  # rank 0 hands its JSON line to this process instead of printing it: see live_traffic()
        fd, line_file = tempfile.mkstemp(prefix="bark_bench_line_", suffix=".json")
        os.close(fd)
    job_token = os.environ.get("BARK_RCCL_ID_TOKEN") or os.urandom(16).hex()  # bark_amd.distributed.exchange_unique_id
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), BARK_BENCH_WORKER="1", BARK_RCCL_ID_TOKEN=job_token)
        if line_file:
            env["BARK_BENCH_LINE_FILE"] = line_file
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this host
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in pending:  # a rank died: the others would wait in a collective for ever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    if line_file:
        try:
            text = open(line_file).read().strip()
        finally:
            os.unlink(line_file)
        if text:
            if rc == 0:
                text = live_traffic(text)
            print(text, flush=True)
    return rc


def live_traffic(line: str) -> str:
    """roofline.traffic measured in THIS run: the worker process has exited, this (parent) process has never touched the GPU, so
    it may start the profiler: two `rocprofv3 --pmc` passes (FETCH_SIZE, then WRITE_SIZE: they do not fit one pass on gfx950;
    counters only, no trace domains, the interpreter directly after `--`) over one warm-up + one c3 sweep of
    tools/profile_mll.py, event joins (BARK_NO_DEVICE_WAIT: counter passes serialise dispatches).  HBM bytes per step =
    2 x FETCH_SIZE + WRITE_SIZE over the sweep's kernels (KiB as reported; FETCH doubled as MI355X_MICROARCH.md prescribes for
    16 B/lane streaming reads on gfx950), per launch sequence.  Any failure leaves the line as the worker wrote it."""
    import csv
    import glob
    import re

    try:
        res = json.loads(line)
        cfg = res["config"]
        N, B = int(cfg["N"]), int(cfg["forests_per_gpu"])
        env = dict(os.environ, BARK_NO_DEVICE_WAIT="1", PYTHONPATH=ROOT, TMPDIR="/tmp")
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "BARK_BENCH_WORKER", "BARK_BENCH_LINE_FILE"):
            env.pop(k, None)
        sweep = ("diag_kernel", "row_kernel", "solve_kernel", "panel_split_kernel", "panel_reduce_kernel")
        calls, totals = 2, {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = tempfile.mkdtemp(prefix="bark_pmc_%s_" % counter)
            try:
                subprocess.run(["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "t", "--", sys.executable,
                                os.path.join(ROOT, "tools", "profile_mll.py"), str(N), str(B), str(calls - 1)],
                               env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240, check=True)
                kib, launches = 0.0, 0
                for f in glob.glob(os.path.join(out, "**", "*counter_collection*.csv"), recursive=True):
                    for row in csv.DictReader(open(f)):
                        m = re.search(r"(\w+_kernel)", row.get("Kernel_Name", ""))
                        if row.get("Counter_Name") == counter and m and m.group(1) in sweep:
                            kib += float(row["Counter_Value"])
                            launches += 1
                if launches == 0:
                    raise RuntimeError("no %s rows for the sweep's kernels" % counter)
                totals[counter] = kib * 1024.0 / calls
            finally:
                shutil.rmtree(out, ignore_errors=True)
        res["roofline"]["traffic"] = 2.0 * totals["FETCH_SIZE"] + totals["WRITE_SIZE"]
        res["roofline"]["traffic_source"] = {
            "measured_in_this_run": True, "fetch_bytes_per_step_raw": totals["FETCH_SIZE"], "write_bytes_per_step": totals["WRITE_SIZE"],
            "command": "rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE> --output-format csv -- python3 tools/profile_mll.py %d %d 1 (two passes, "
                       "started by the GPU-free launcher after the timed worker exited; FETCH doubled on gfx950)" % (N, B),
            "csrc_digest": csrc_digest()}
        return json.dumps(res)
    except Exception as exc:  # the committed profile's figure (or null) stays in the line
        try:
            res = json.loads(line)
            src = res["roofline"].get("traffic_source")
            note = "live PMC passes failed (%s)" % (repr(exc)[:200])
            if isinstance(src, dict):
                src["live_attempt"] = note
            else:
                res["roofline"]["traffic_source"] = "%s; %s" % (src, note)
            return json.dumps(res)
        except Exception:
            return line


# ---------------------------------------------------------------------------------------------
# helpers (workers)
# ---------------------------------------------------------------------------------------------
def host_cores() -> int:
    """Cores this process can really run on: CPU affinity capped by the cgroup CPU quota (cpu.max)."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, int(quota / period + 0.5)))
            break
        except Exception:
            continue
    return n


def csrc_digest() -> str:
    """Content hash of the kernel sources (there is no .git on the GPU box): ties a committed PMC profile to a build."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "bark_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".cpp", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def hbm_traffic_from_profile(N, B, m):
    """HBM bytes per step of the Cholesky launch sequence from a committed rocprofv3 PMC profile of this command
    (tools/hbm_counters.sh -> profiles/rNN/*hbm_counters*.json: FETCH_SIZE and WRITE_SIZE in separate --pmc passes,
    FETCH doubled as MI355X_MICROARCH.md prescribes for 16 B/lane streaming reads on gfx950).  bench.py cannot
    collect PMC counters itself, so the figure is only reported when the profile was taken on THIS build of the
    kernels (csrc digest) and this workload; otherwise null."""
    import glob

    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "*hbm_counters*.json"))):
        try:
            prof = json.load(open(f))
            meta = prof.get("meta", {})
            if meta.get("csrc_digest") != csrc_digest() or (meta.get("N"), meta.get("B"), meta.get("m")) != (N, B, m):
                continue
            steps = max(int(meta.get("steps_profiled", 1)), 1)
            total = 0.0
            for k, v in prof["FETCH_SIZE"].items():
                if k in meta.get("sweep_kernels", []):
                    total += 2.0 * v["sum_KiB"] * 1024.0
            for k, v in prof["WRITE_SIZE"].items():
                if k in meta.get("sweep_kernels", []):
                    total += v["sum_KiB"] * 1024.0
            best = {"bytes_per_step": total / steps, "source": os.path.relpath(f, ROOT), "from_committed_profile": True,
                    "csrc_digest": meta.get("csrc_digest")}
        except Exception:
            continue
    return best


class Workload:
    """Device-resident inputs of one rank's batch + the production-path call."""

    def __init__(self, N, d, m, B, seed_base, rank_offset, chunk=0, problem="unit", C=0, noise_seed=None,
                 include_scale=False):
        import numpy as np
        import torch

        from bark_amd import _lib, synthetic
        from bark_amd.fitting.mll import choose_chunk
        from bark_amd.forest import PackedForest

        self.N, self.d, self.m, self.B, self.C = N, d, m, B, C
        if problem in ("unit", "stress"):
            X, y, bounds, ft = synthetic.unit_cube_problem(N, d, seed=seed_base)
            cand = None
        else:  # c5: mixed categorical + integer + continuous
            X, y, bounds, ft = synthetic.mixed_problem(N, seed=seed_base)
            cand = synthetic.mixed_problem(C, seed=seed_base + 1)[0] if C else None
            self.d = d = X.shape[1]
        self.X, self.y, self.ft = X, y, ft
        if problem == "stress":  # SURVEY §8d stress variant: complete depth-5 trees, 32 leaves each (prior trees average 2.4)
            self.forests = synthetic.full_binary_forests(B, m, d, 5, np.random.default_rng(seed_base + rank_offset))
        else:
            self.forests = synthetic.sample_prior_forests(B, m, bounds, ft, seed=seed_base + rank_offset)
        rng = np.random.default_rng(seed_base + 7919 * ((noise_seed if noise_seed is not None else rank_offset) + 1))
        self.noise = rng.uniform(0.05, 0.15, size=B)
        self.lib = lib = _lib.lib()
        self._lib = _lib
        self.pf = PackedForest(self.forests, ft)  # host format conversion + upload: outside every timed region
        self.Xd = _lib.to_device(X)
        self.yd = _lib.to_device(y.reshape(-1))
        self.noise_d = _lib.to_device(self.noise)
        self.scale_d = _lib.to_device(np.ones(B)) if (include_scale or C) else None
        self.cand_d = _lib.to_device(cand) if cand is not None else None
        dev = self.Xd.device
        self.mll_d = torch.empty(B, dtype=torch.float64, device=dev)
        self.info_d = torch.empty(B, dtype=torch.int32, device=dev)
        self.mu_d = torch.empty((B, C), dtype=torch.float64, device=dev) if C else None
        self.var_d = torch.empty((B, C), dtype=torch.float64, device=dev) if C else None
        self.Bc = chunk or choose_chunk(B, N, C, m)
        self.ws = torch.empty(int(lib.bark_mll_workspace_bytes(N, C, m, self.Bc)), dtype=torch.uint8, device=dev)
        self.flags = (_lib.MLL_INCLUDE_SCALE if self.scale_d is not None else 0) | (0 if C else _lib.MLL_INCLUDE_2PI)
        self.stream = _lib.stream_ptr()

    def run(self, timing=None):
        L = self._lib
        L.check(self.lib.bark_mll_batched_hip(
            L.ctx(), L.ptr(self.pf.packed), self.pf.info_ref, L.ptr(self.Xd), self.N, self.d, L.ptr(self.yd), L.ptr(self.noise_d),
            L.ptr(self.scale_d), None, self.flags, L.ptr(self.cand_d), self.C, L.ptr(self.mll_d), L.ptr(self.mu_d),
            L.ptr(self.var_d), None, L.ptr(self.info_d), L.ptr(self.ws), self.ws.numel(), self.Bc,
            ctypes.byref(timing) if timing is not None else None, self.stream))
        return self.mll_d

    def settle_device_wait(self):
        """One checked call before anything is timed: chunks of at most 32 matrices hand their row launches over by device-side
        counters (include/bark_hip.h, bark_device_wait), which needs the library's helper streams to run beside the caller's.
        Where they cannot (eight processes on one node is the first place that may differ from a one-GPU box) the call
        reports info = -3 after a bounded wait: drain, switch the mechanism off for the process and repeat — what
        bark_amd.fitting.mll does for API callers.  -> True if the fallback was taken."""
        import torch

        if self.Bc > 32 or device_wait_state(self.lib) == "off":  # chunks of more than 32 matrices never use the mechanism
            return False
        self.run()
        if not bool((self.info_d == -3).any().item()):
            return False
        torch.cuda.synchronize()
        self.lib.bark_device_wait(0)
        self.run()
        torch.cuda.synchronize()
        assert not bool((self.info_d == -3).any().item()), "device-side wait timed out with the mechanism switched off"
        return True

    def device_ms(self, reps, warm=1):
        """Average duration of one production-path call from HIP events on the launch stream."""
        import torch

        self.settle_device_wait()
        for _ in range(warm):
            self.run()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        e[0].record()
        for i in range(reps):
            self.run()
            e[i + 1].record()
        torch.cuda.synchronize()
        spans = sorted(e[i].elapsed_time(e[i + 1]) for i in range(reps))
        return sum(spans) / reps, spans[len(spans) // 2]

    def graph_ms(self, reps):
        """Device time of one call replayed from a captured hipGraph (the `*_hip` entry points only enqueue work, so a
        latency-bound caller can capture the launch sequence once and replay it): median over `reps` replays."""
        import torch

        from bark_amd import _lib

        self.run()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        keep = self.stream
        with torch.cuda.graph(g):
            self.stream = _lib.stream_ptr()  # the capture stream
            self.run()
        self.stream = keep
        g.replay()
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        e[0].record()
        for i in range(reps):
            g.replay()
            e[i + 1].record()
        torch.cuda.synchronize()
        spans = sorted(e[i].elapsed_time(e[i + 1]) for i in range(reps))
        return spans[len(spans) // 2]

    def check(self):
        import numpy as np

        assert int(self.info_d.abs().max().item()) == 0, "a kernel matrix was not positive definite"
        host = self.mll_d.cpu().numpy()
        assert np.isfinite(host).all()
        return host


class TorchCollective:
    """The exchanges of a run over torch.distributed (RCCL = the `nccl` backend, or gloo)."""

    def __init__(self, dist, backend):
        self.dist, self.backend = dist, backend

    def barrier(self):
        self.dist.barrier()

    def max_over_ranks(self, value, device):
        import torch

        t = torch.tensor([value], dtype=torch.float64, device=device if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather(self, local, total):
        from bark_amd.distributed import gather_mll

        return gather_mll(local if self.backend == "nccl" else local.cpu(), total)  # gloo moves host tensors

    def close(self):
        self.dist.destroy_process_group()


class AbiCollective:
    """The same exchanges through the C ABI's RCCL entry points (bark_amd.distributed.RcclGroup): no torch.distributed."""

    def __init__(self, group):
        self.group, self.backend = group, "rccl-abi"

    def barrier(self):
        self.group.barrier()

    def max_over_ranks(self, value, device):
        import torch

        t = torch.tensor([value], dtype=torch.float64, device=device)
        self.group.all_reduce(t, "max")
        return float(t.item())

    def gather(self, local, total):
        return self.group.gather_mll(local, total)

    def close(self):
        self.group.close()


def device_wait_state(lib) -> str:
    """on / off: the process-wide switch of the device-side hand-over (chunks of at most 32 matrices; include/bark_hip.h) at the
    end of the run — `off` with $BARK_NO_DEVICE_WAIT, under a dispatch-serialising environment, or after a time-out."""
    prev = lib.bark_device_wait(1)
    lib.bark_device_wait(prev)
    return "on" if prev else "off"


class PowerSampler:
    """Socket power and shader clock of THIS process's card while the timed steps run (hwmon of the card whose render node the
    process may open; a second thread, 20 ms period, sysfs reads only).  The c3 sweep runs at the socket power cap
    (profiles/r04/headline_power_wall.txt): these three numbers put that on record in the line itself.  Returns None when the
    box exposes no readable hwmon node."""

    def __init__(self, device_index=None):
        import glob

        self.hw = None
        if device_index is not None:  # the PCI address torch reports for this rank's device names the card directly
            try:
                import torch

                pr = torch.cuda.get_device_properties(device_index)
                addr = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
                cands = glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % addr)
                if cands and os.path.exists(os.path.join(cands[0], "power1_input")):
                    self.hw = cands[0]
            except Exception:  # older torch: no pci_* fields
                self.hw = None
        self.rows, self._stop, self._th = [], None, None
        if self.hw is not None:
            return
        for rn in glob.glob("/sys/class/drm/renderD*"):
            if not os.access("/dev/dri/" + os.path.basename(rn), os.R_OK | os.W_OK):
                continue
            dev = os.path.realpath(os.path.join(rn, "device"))
            cands = glob.glob(os.path.join(dev, "hwmon", "hwmon*"))
            if cands and os.path.exists(os.path.join(cands[0], "power1_input")):
                if self.hw is not None:
                    self.hw = None  # more than one card is ours: no way to say which one this rank drives
                    break
                self.hw = cands[0]

    def _read(self, name):
        try:
            with open(os.path.join(self.hw, name)) as f:
                return float(f.read())
        except (OSError, ValueError):
            return None

    def start(self):
        import threading

        if self.hw is None:
            return
        self._stop = threading.Event()

        def loop():
            while not self._stop.is_set():
                pw, fq = self._read("power1_input"), self._read("freq1_input")
                if pw is not None and fq is not None:
                    self.rows.append((pw / 1e6, fq / 1e6))
                time.sleep(0.02)

        self._th = threading.Thread(target=loop, daemon=True)
        self._th.start()

    def stop(self):
        if self._th is None:
            return None
        self._stop.set()
        self._th.join()
        rows = self.rows[len(self.rows) // 5:]  # the first fifth: ramp-up of the averaged power reading
        if len(rows) < 5:
            return None
        med = lambda v: sorted(v)[len(v) // 2]
        cap = self._read("power1_cap")
        return {"socket_power_w_median": med([r[0] for r in rows]), "socket_power_cap_w": cap / 1e6 if cap else None,
                "sclk_mhz_median": med([r[1] for r in rows]), "samples": len(rows),
                "source": "hwmon power1_input / freq1_input / power1_cap of this rank's card, sampled every 20 ms over the timed steps"}


def timed_region(wl, steps, warmup, world, coll, gather):
    """W warm-up steps, then exactly K steps between barrier + synchronize fences; max over ranks."""
    import torch

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            coll.barrier()
        torch.cuda.synchronize()

    out = None
    wl.settle_device_wait()  # untimed; a device-side wait that cannot work on this node is found (and switched off) here
    for _ in range(warmup):
        out = gather(wl.run())
    fence()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * steps)]
    power = PowerSampler(wl.Xd.device.index)  # every rank samples its own card; rank 0's goes into the line
    if power:
        power.start()
    t0 = time.perf_counter()
    for i in range(steps):
        ev[2 * i].record()      # HIP events on the launch stream, around the library call only
        mll = wl.run()
        ev[2 * i + 1].record()
        out = gather(mll)
    fence()
    elapsed = time.perf_counter() - t0
    timed_region.power = power.stop() if power else None
    if world > 1:
        elapsed = coll.max_over_ranks(elapsed, wl.Xd.device)
    call_ms = sum(ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(steps)) / steps
    return elapsed, call_ms, out


def require_rccl(args, world, requested) -> bool:
    """--require-rccl defaults to ON whenever there is an RCCL leg at all (more than one rank, nccl requested)."""
    if world <= 1 or requested != "nccl":
        return False
    return True if args.require_rccl is None else bool(args.require_rccl)


def init_collective(world, rank, dev_index, requested, must_be_rccl):
    """Collective object for the one exchange of the path (the MLL gather) -> (TorchCollective | AbiCollective | None,
    rccl_ranks_seen).  `requested`: nccl (default; torch.distributed over RCCL), gloo (dry runs), abi (RCCL through the C ABI).
    `nccl` IS RCCL on ROCm; one tiny all-reduce proves the communicator before any timed region, and its value — the
    number of ranks RCCL actually saw — goes into the JSON.  If RCCL cannot be brought up: with `must_be_rccl` the
    worker exits non-zero (the launcher then ends every rank); without it the gather falls back to gloo in the SAME
    process (no new process image), which changes nothing measurable for 8 bytes per sample but is stated in the JSON."""
    if world <= 1:
        return None, None
    import datetime

    import torch
    import torch.distributed as dist

    kw = dict(rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
    if requested == "abi":  # RCCL through the C ABI (bark_comm_* / bark_allgather_mll): no process group at all
        from bark_amd.distributed import RcclGroup

        group = RcclGroup(rank, world, dev_index, os.environ.get("MASTER_ADDR", "127.0.0.1"),
                          int(os.environ.get("MASTER_PORT", "29500")) + 29)
        probe = torch.ones(1, dtype=torch.float64, device=torch.device("cuda", dev_index))
        group.all_reduce(probe)
        torch.cuda.synchronize()
        seen = int(probe.item())
        if seen != world:
            raise SystemExit(f"RCCL (C ABI) all-reduce probe saw {seen} ranks, expected {world}")
        return AbiCollective(group), seen
    if requested != "nccl":
        dist.init_process_group(requested, **kw)
        return TorchCollective(dist, requested), None
    try:
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index), **kw)
        probe = torch.ones(1, device=torch.device("cuda", dev_index))
        dist.all_reduce(probe)
        torch.cuda.synchronize()
        seen = int(probe.item())
        if seen != world:
            raise RuntimeError(f"RCCL all-reduce probe saw {seen} ranks, expected {world}")
        return TorchCollective(dist, "nccl"), seen
    except Exception as exc:  # every rank sees the same failure
        if must_be_rccl:
            print(f"[bench rank {rank}] RCCL unavailable ({exc!r}) and --require-rccl is in force: giving up "
                  "(pass --no-require-rccl to let the MLL gather use gloo)", file=sys.stderr, flush=True)
            raise SystemExit(3)
        print(f"[bench rank {rank}] RCCL unavailable ({exc!r}); --no-require-rccl: falling back to gloo for the MLL gather",
              file=sys.stderr, flush=True)
        try:
            dist.destroy_process_group()
        except Exception:
            pass
        os.environ["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 17)
        dist.init_process_group("gloo", **kw)
        return TorchCollective(dist, "gloo"), None


# ---------------------------------------------------------------------------------------------
# worker
# ---------------------------------------------------------------------------------------------
def selftest_worker(args, world, rank):
    """Launcher / rendezvous / collective set-up (incl. the RCCL-unavailable branch: BARK_BENCH_BACKEND=nccl on a box
    without GPUs) / gather / timing protocol on CPU: everything a worker does around the GPU work, for the weak region
    and — in weak mode with more than one rank — the c4 (512 samples, strong) region."""
    import torch
    import torch.distributed as dist

    from bark_amd.distributed import gather_mll, shard_range

    requested = os.environ.get("BARK_BENCH_BACKEND", "gloo")
    coll, seen = init_collective(world, rank, 0, requested, require_rccl(args, world, requested))
    backend = coll.backend if coll else None

    def region(total, lo, hi):
        local = torch.arange(lo, hi, dtype=torch.float64)
        if world > 1:
            coll.barrier()
        t0 = time.perf_counter()
        full = coll.gather(local, total) if world > 1 else local
        elapsed = time.perf_counter() - t0
        if world > 1:
            elapsed = coll.max_over_ranks(elapsed, "cpu")
        return bool((full == torch.arange(total, dtype=torch.float64)).all()), elapsed

    total = args.total or args.batch * world
    lo, hi = shard_range(total, rank, world) if args.total else (rank * args.batch, (rank + 1) * args.batch)
    ok, _ = region(total, lo, hi)
    c4 = None
    if world > 1 and not args.total:  # the second timed region of a weak-scaling run
        l4, h4 = shard_range(512, rank, world)
        ok4, _ = region(512, l4, h4)
        c4 = {"total": 512, "local": h4 - l4, "gather_ok": ok4, "scaling": "strong"}
        ok = ok and ok4
    if rank == 0:
        print(json.dumps({"selftest": "launcher", "n_gpus": world, "ranks_reached": world, "gather_ok": ok,
                          "total": total, "local": hi - lo, "scaling": "strong" if args.total else "weak",
                          "collective_backend_requested": requested if world > 1 else None,
                          "collective_backend": backend, "rccl_ranks_seen": seen, "c4_strong": c4,
                          "launched_by": "bench.py" if os.environ.get("BARK_BENCH_WORKER") else "external"}), flush=True)
    if world > 1:
        coll.barrier()
        coll.close()
    return 0 if ok else 1


def worker(args) -> int:
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.selftest_launcher:
        return selftest_worker(args, world, rank)

    import numpy as np
    import torch
    import torch.distributed as dist

    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists)")
    backend = os.environ.get("BARK_BENCH_BACKEND", "nccl")  # nccl | gloo | abi (RCCL through the C ABI, no torch.distributed)
    if backend in ("nccl", "abi") and world > ndev:
        raise SystemExit(f"--gpus {world} but only {ndev} devices (set BARK_BENCH_BACKEND=gloo to share a GPU in a dry run)")
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    requested = backend
    coll, rccl_ranks_seen = init_collective(world, rank, dev_index, requested, require_rccl(args, world, requested))
    backend = coll.backend if coll else None

    from bark_amd import _lib
    from bark_amd.distributed import gather_mll, shard_range

    N, m, d = args.n, args.trees, args.dim
    strong = args.total > 0
    if strong:
        lo, hi = shard_range(args.total, rank, world)
        B, offset, total = hi - lo, lo, args.total
        if B < 1:
            raise SystemExit("--total smaller than the number of GPUs")
    else:
        B, offset, total = args.batch, rank * args.batch, args.batch * world

    def make_gather(total_):
        if world == 1:
            return lambda t: t
        return lambda t: coll.gather(t, total_)

    # ---- synthetic inputs: same X, y on every rank; rank r owns forests offset .. offset+B-1
    wl = Workload(N, d, m, B, seed_base=N, rank_offset=offset, chunk=args.chunk, noise_seed=rank)
    elapsed, call_ms, all_mll = timed_region(wl, args.steps, args.warmup, world, coll, make_gather(total))
    power_c3 = getattr(timed_region, "power", None)
    mll_host = wl.check()
    assert all_mll.shape[0] == total and bool(torch.isfinite(all_mll).all())

    # ---- c4 split (BASELINE configs[3]: 512 samples over the GPUs) as a second timed region in weak mode, N > 1
    c4 = None
    if world > 1 and not strong and N == 4096:
        lo, hi = shard_range(512, rank, world)
        wl4 = Workload(N, d, m, hi - lo, seed_base=N, rank_offset=lo, noise_seed=rank)
        e4, call4, out4 = timed_region(wl4, args.steps, args.warmup, world, coll, make_gather(512))
        wl4.check()
        c4 = {"workload": "c4: 512 forest samples sharded over %d GPUs (%d per GPU)" % (world, hi - lo), "scaling": "strong",
              "value": 512 * args.steps / e4, "unit": "evals/s", "ms_per_step": e4 / args.steps * 1e3,
              "call_ms_rank0": call4}
        del wl4
    if world > 1:
        coll.barrier()  # every rank leaves here; rank 0's untimed extras below run with no peer waiting in a collective
        coll.close()
    if rank != 0:
        return 0

    steps = args.steps
    value = total * steps / elapsed
    chol_flops = B * N**3 / 3.0  # algorithmic flops of one launch sequence (SURVEY §8d flops_chol x B)
    chol_tflops = chol_flops / (call_ms * 1e-3) / 1e12
    traffic = hbm_traffic_from_profile(N, B, m)
    lib = wl.lib

    # ---- one instrumented step (outside the timed region): per-kernel HIP-event spans inside the library
    timing = _lib.MllTiming()
    wl.run(timing)
    breakdown = {k: round(float(getattr(timing, k)), 3) for k in ("total_ms", "gram_ms", "chol_ms", "diag_ms", "panel_ms", "solve_ms")}
    kern = {}
    for name, ms_key, n_key, fl_key in (("row_kernel", "panel_ms", "n_panel_launches", "panel_flops"),
                                        ("solve_kernel", "solve_ms", "n_solve_launches", "solve_flops"),
                                        ("diag_kernel", "diag_ms", "n_diag_launches", None)):
        nl = int(getattr(timing, n_key))
        ms = float(getattr(timing, ms_key))
        kern[name] = {"launches_per_step": nl, "avg_ms": ms / max(nl, 1)}
        if fl_key:
            kern[name]["executed_tflops"] = float(getattr(timing, fl_key)) / (max(ms, 1e-9) * 1e-3) / 1e12

    result = {
        "metric": METRIC,
        "value": value,
        "unit": "evals/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": ("c4: N=%d d=%d m=%d, %d prior forest samples sharded over %d GPUs" % (N, d, m, total, world)) if strong else
                        ("c3: N=%d d=%d m=%d, %d prior forest samples per GPU (BASELINE configs[2]; sharding of "
                         "configs[3])" % (N, d, m, B)) + ", noise U[0.05,0.15), mcmc_record_mll convention",
            "N": N, "d": d, "trees": m, "forests_per_gpu": B, "forests_total": total, "chunk": wl.Bc,
            "parallelism": "samples/%d" % world, "collective_backend": backend,
            "collective_backend_requested": (requested if world > 1 else None), "rccl_ranks_seen": rccl_ranks_seen,
            "launched_by": "bench.py" if os.environ.get("BARK_BENCH_WORKER") else "external",
            "timed_path": "production (timing=NULL, no host sync inside the library)",
            "device_wait": device_wait_state(lib),
        },
        "roofline": {
            "bound": "mfma",
            "kernel": "Cholesky launch sequence of one step (the fp64 MFMA panel kernel dominates); `achieved` = algorithmic "
                      "B*N^3/3 flops / average duration of the library call, HIP events on the launch stream around every "
                      "timed step",
            "achieved": chol_tflops,
            "peak": F64_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": chol_tflops / F64_MFMA_PEAK_TFLOPS,
            "traffic": traffic["bytes_per_step"] if traffic else None,
            "traffic_unit": "bytes of HBM traffic per step (one Cholesky launch sequence)",
            "traffic_source": traffic if traffic else "no committed PMC profile matches this build (csrc digest %s) "
                                                      "and workload" % csrc_digest(),
            "power": power_c3,
            "algorithmic_flops_per_step": chol_flops,
            "call_ms": call_ms,
            "instrumented_step_ms": breakdown,
            "kernels": kern,
        },
    }
    if c4:
        result["c4_strong"] = c4

    if world == 1:
        extras(args, wl, result, mll_host)
    out_file = os.environ.get("BARK_BENCH_LINE_FILE")
    if out_file:  # the launcher adds the live HBM counters and prints the line (live_traffic)
        with open(out_file, "w") as f:
            f.write(json.dumps(result))
    else:
        print(json.dumps(result), flush=True)
    return 0


def numpy_api_probe(wl, reps=3):
    """The same c3 workload through the drop-in numpy API (bark_amd.fitting.batched_mll): host packing of the 26-byte
    records (or the content-hash cache hit), H2D of the packed forest / X / y / noise, the sweep, D2H of the (B,) result.
    PCIe-inclusive; never `value`."""
    import numpy as np

    import bark_amd.fitting as fit
    from bark_amd import forest as bforest

    def call(F):
        t = time.perf_counter()
        out = fit.batched_mll(F, wl.noise, None, wl.X, wl.y, wl.ft, include_scale=False, include_2pi=True, chunk=wl.Bc)
        return time.perf_counter() - t, out

    call(wl.forests)  # warm: workspace growth, first-use costs
    cold = []
    for i in range(reps):  # a changed batch every call (one record flipped back and forth): nothing to reuse
        F = wl.forests.copy()
        F[i, 0, 99]["depth"] = 7 + i  # an inactive, unreachable slot: same forests, different bytes
        cold.append(call(F)[0])
    h0 = dict(bforest.pack_cache_stats)
    warm = [call(wl.forests)[0] for _ in range(reps)]
    h1 = dict(bforest.pack_cache_stats)
    _, out = call(wl.forests)
    assert np.array_equal(out, wl.mll_d.cpu().numpy())
    B = wl.B
    return {"workload": "c3 through bark_amd.fitting.batched_mll on numpy inputs (pack + H2D + sweep + D2H)",
            "ms_new_forests": 1e3 * sorted(cold)[len(cold) // 2], "evals_per_s_new_forests": B / sorted(cold)[len(cold) // 2],
            "ms_same_forests_again": 1e3 * sorted(warm)[len(warm) // 2],
            "evals_per_s_same_forests_again": B / sorted(warm)[len(warm) // 2],
            "pack_cache_hits_in_repeat_calls": h1["hits"] - h0["hits"], "pack_cache_misses_in_repeat_calls": h1["misses"] - h0["misses"]}


def sampler_step_probe(args, wl):
    """SURVEY §8f-1 on record: one step of `_step_bark_sampler` (bark_sampler.py:217-284) at N = 4096, m = 50 — a sweep of
    50 tree proposals decided on the device (ChainBatch.sweep_trees) + the noise/scale proposal (leaf-space MLL) + the
    rebuild of the resident inverse on accept — for 1 and 4 chains; beside it the CPU oracle's Woodbury chain for the
    same proposals (bounded sample).  Wall time includes the host side of the step (packing the tree pairs)."""
    import numpy as np
    import torch

    import bark_amd.fitting as fit
    from bark_amd import synthetic

    N, m = wl.N, wl.m
    X, y, ft = wl.X, wl.y, wl.ft
    bounds = np.tile(np.array([[0.0, 1.0]]), (wl.d, 1))
    Xd = wl.Xd
    out = {"workload": "N=%d m=%d: 50 tree proposals (device-side Metropolis) + noise/scale proposal + rebuild" % (N, m)}
    # first: a host round trip per proposal is a latency measurement, and the CPU-oracle legs below leave LAPACK's worker threads
    # spinning on the box's cores for a while (measured: 4.3 ms per proposal right after them, 0.10-0.15 ms before)
    out["per_proposal_routes"] = proposal_routes_probe(wl)
    rng = np.random.default_rng(5)
    for nc in (1, 4):
        cur = synthetic.sample_prior_forests(nc, m, bounds, ft, seed=7000)
        prop = synthetic.sample_prior_forests(nc, m, bounds, ft, seed=8000)
        noise, scale = np.full(nc, 0.1), np.ones(nc)
        log_q, log_u = rng.normal(0.0, 0.5, size=(nc, m)), np.log(rng.uniform(size=(nc, m)))
        cb = fit.ChainBatch.from_forests(cur, noise, scale, Xd, y, ft)
        cb.sweep_trees(cur, prop, log_q, log_u, Xd, ft, scale, m)  # warm-up (workspaces, packer)
        torch.cuda.synchronize()
        sweeps = []
        for _ in range(3):
            cb = fit.ChainBatch.from_forests(cur, noise, scale, Xd, y, ft)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            mask = cb.sweep_trees(cur, prop, log_q, log_u, Xd, ft, scale, m)
            e1.record()
            torch.cuda.synchronize()
            sweeps.append((time.perf_counter() - t0, e0.elapsed_time(e1)))
        wall, dev = sorted(sweeps)[1]
        final = cur.copy()
        final[mask] = prop[mask]
        # noise/scale half: proposal MLL in leaf space, rebuild of the resident inverses on accept
        def timed(fn, reps=5):
            fn()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / reps

        ns_eval = timed(lambda: fit.batched_mll(final, noise * 1.1, scale * 0.9, Xd, y, ft, include_scale=True,
                                                include_2pi=False, method="leafspace"))
        rebuild = timed(lambda: fit.batched_kernel_inverse(final, noise * 1.1, scale * 0.9, Xd, y, ft, no_null=False,
                                                           return_device=True, method="leafspace"), reps=3)
        dense_rebuild = timed(lambda: fit.batched_kernel_inverse(final, noise * 1.1, scale * 0.9, Xd, y, ft, no_null=False,
                                                                 return_device=True), reps=2)
        out["chains_%d" % nc] = {
            "tree_sweep_wall_ms": 1e3 * wall, "tree_sweep_device_ms": dev, "ms_per_tree_proposal_per_chain": 1e3 * wall / (m * nc),
            "accepted": int(mask.sum()), "noise_scale_proposal_ms": 1e3 * ns_eval, "rebuild_on_accept_ms_leafspace": 1e3 * rebuild,
            "rebuild_on_accept_ms_dense": 1e3 * dense_rebuild,
            "step_ms_if_noise_scale_accepted": 1e3 * (wall + ns_eval + rebuild),
            "step_ms_if_noise_scale_rejected": 1e3 * (wall + ns_eval)}
        if nc == 1 and args.cpu_sample > 0:  # the oracle's restated chain (quick_inverse.py:13-38) on the same proposals
            from oracle import oracle as orc

            cpu, n_cpu, reps = oracle_chain_ms(orc, cur[0], prop[0], X, y, ft, m, n_props=3, reps=5)
            out["cpu_oracle"] = {"ms_per_tree_proposal": cpu, "cores": host_cores(), "kind": "port",
                                 "sample": "median of %d repetitions (after one un-timed warm-up pass) of %d tree proposals of "
                                           "chain 0 through the oracle's subtract/add Woodbury chain (numpy, N x N copies as "
                                           "the reference makes)" % (reps, n_cpu)}
        del cb
        torch.cuda.empty_cache()
    # the regime BARK itself samples in (tens to hundreds of points, one chain): a proposal is a chain of five small dependent
    # launches there, not bytes — the tree sweep alone, beside the oracle's chain on the same proposals
    small = {}
    for Ns in (128, 512):
        Xs, ys, bounds_s, fts = synthetic.unit_cube_problem(Ns, wl.d, seed=Ns)
        cur = synthetic.sample_prior_forests(1, m, bounds_s, fts, seed=7000)
        prop = synthetic.sample_prior_forests(1, m, bounds_s, fts, seed=8000)
        noise, scale = np.full(1, 0.1), np.ones(1)
        log_q, log_u = rng.normal(0.0, 0.5, size=(1, m)), np.log(rng.uniform(size=(1, m)))
        Xsd = torch.from_numpy(Xs).to(Xd.device)
        walls = []
        for it in range(4):
            cb = fit.ChainBatch.from_forests(cur, noise, scale, Xsd, ys, fts)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            cb.sweep_trees(cur, prop, log_q, log_u, Xsd, fts, scale, m)
            torch.cuda.synchronize()
            walls.append(time.perf_counter() - t0)
        row = {"tree_sweep_wall_ms": 1e3 * sorted(walls[1:])[1], "ms_per_tree_proposal": 1e3 * sorted(walls[1:])[1] / m}
        if args.cpu_sample > 0:
            from oracle import oracle as orc

            # warmed and repeated (ten un-warmed iterations right after GPU work once read 14.2 ms at
            # N = 128 against 0.80 ms at N = 512 — the first LAPACK calls of the process, not the chain)
            row["cpu_oracle_ms_per_tree_proposal"], n_cpu, reps = oracle_chain_ms(orc, cur[0], prop[0], Xs, ys, fts, m, n_props=10, reps=7)
            row["cpu_oracle_sample"] = "median of %d repetitions of %d proposals, one warm-up pass" % (reps, n_cpu)
        small["N=%d" % Ns] = row
        del cb
    out["small_n_one_chain"] = small
    return out


def oracle_chain_ms(orc, cur, prop, X, y, ft, m, n_props, reps):
    """The oracle's restatement of one tree proposal of `_step_bark_sampler` (bark_sampler.py:233-257 with
    quick_inverse.py:13-38: two get_leaf_vectors, subtract / add low_rank_inv_update + low_rank_det_update, mll) on the host:
    ms per proposal = MEDIAN over `reps` timed passes of the first `n_props` trees, after one un-timed pass that pays for
    the first LAPACK calls, thread-pool start-up and page faults of the N x N temporaries.  -> (ms, n_props, reps)"""
    import numpy as np

    N = X.shape[0]
    K = orc.forest_gram_matrix(cur, X, X, ft)
    K[np.diag_indices(N)] += 1e-6 + 0.1
    K_inv, logdet = np.linalg.inv(K), np.linalg.slogdet(K)[1]
    s_sqrtm = np.sqrt(1.0 / m)

    def one_pass():
        t = time.perf_counter()
        for ti in range(n_props):
            U_old = s_sqrtm * orc.get_leaf_vectors(cur[ti], X, ft)
            U_new = s_sqrtm * orc.get_leaf_vectors(prop[ti], X, ft)
            K1 = orc.low_rank_inv_update(K_inv, U_old, subtract=True)
            d1 = orc.low_rank_det_update(K_inv, U_old, logdet, subtract=True)
            K2 = orc.low_rank_inv_update(K1, U_new)
            orc.mll(K2, orc.low_rank_det_update(K1, U_new, d1), y)
        return (time.perf_counter() - t) / n_props

    one_pass()  # warm-up, un-timed
    times = sorted(one_pass() for _ in range(reps))
    return 1e3 * times[len(times) // 2], n_props, reps


def proposal_routes_probe(wl, n_props=30):
    """One tree proposal of `_step_bark_sampler` (bark_sampler.py:233-257) with a host round trip per proposal, as the
    reference's loop is written, through the two routes of INTEGRATION.md §4: (b) `ChainState.propose_tree` (Python + torch
    tensors: the de-jitted step) and (a) `bark_tree_swap_eval_host_pair` (integer arguments only: what the `@njit` step can
    call through a ctypes function pointer).  Wall time per proposal, one chain, rejected proposals (no rewrite)."""
    import ctypes as C

    import numpy as np
    import torch

    import bark_amd.fitting as fit
    from bark_amd import _lib, synthetic

    N, m, d = wl.N, wl.m, wl.d
    bounds = np.tile(np.array([[0.0, 1.0]]), (d, 1))
    cur = synthetic.sample_prior_forests(1, m, bounds, wl.ft, seed=7000)[0]
    prop = synthetic.sample_prior_forests(1, m, bounds, wl.ft, seed=8000)[0]
    st = fit.ChainState.from_forest(cur, 0.1, 1.0, wl.Xd, wl.y, wl.ft)
    n_props = min(n_props, m)
    vals_b = []
    st.propose_tree(cur[0], prop[0], wl.Xd, wl.ft, 1.0, m)  # warm-up
    t = time.perf_counter()
    for ti in range(n_props):
        vals_b.append(st.propose_tree(cur[ti], prop[ti], wl.Xd, wl.ft, 1.0, m))
    route_b = (time.perf_counter() - t) / n_props
    # route (a): everything below is int / float / array-address arguments
    lib = _lib.lib()
    ft = np.ascontiguousarray(wl.ft, dtype=np.int64)
    ws = torch.empty(int(lib.bark_tree_swap_workspace_bytes(N, 64)), dtype=torch.uint8, device=wl.Xd.device)
    scalars, r_out = np.empty(2), np.zeros(1, dtype=np.int64)
    pair = np.empty((2, cur.shape[1]), dtype=cur.dtype)
    k_inv, x_dev, y_dev, ws_dev, ctx = st.K_inv.data_ptr(), wl.Xd.data_ptr(), st.y.data_ptr(), ws.data_ptr(), _lib.ctx()
    stream = _lib.stream_ptr()
    s_sqrtm = float(np.sqrt(1.0 / m))

    def host_pair(ti):
        pair[0], pair[1] = cur[ti], prop[ti]
        _lib.check(lib.bark_tree_swap_eval_host_pair(ctx, k_inv, N, pair.ctypes.data, pair.shape[1], ft.ctypes.data, d, x_dev, s_sqrtm,
                                                     y_dev, scalars.ctypes.data, r_out.ctypes.data, ws_dev, ws.numel(), stream))
        return 0.5 * (-(st.quad - scalars[0]) - (st.logdet + scalars[1]))

    host_pair(0)
    vals_a = []
    t = time.perf_counter()
    for ti in range(n_props):
        vals_a.append(host_pair(ti))
    route_a = (time.perf_counter() - t) / n_props
    assert np.allclose(vals_a, vals_b, rtol=1e-12, atol=1e-9), "the two routes disagree"
    # the pure host share of route (a): packing the pair (no device work)
    info = _lib.PackInfo()
    t = time.perf_counter()
    for ti in range(n_props):
        pair[0], pair[1] = cur[ti], prop[ti]
        lib.bark_forest_pack_info(pair.ctypes.data, 1, 2, pair.shape[1], ft.ctypes.data, d, C.byref(info))
    pack = (time.perf_counter() - t) / n_props
    return {"workload": "N=%d m=%d, %d proposals, one chain, one host round trip per proposal" % (N, m, n_props),
            "chainstate_propose_tree_ms": 1e3 * route_b, "host_pair_entry_point_ms": 1e3 * route_a,
            "host_share_of_the_entry_point_ms (pair copy + pack_info, Python call included)": 1e3 * pack,
            "routes_agree_rtol": 1e-12}


def fitting_loop_probe(calls=100):
    """SURVEY §8f-4 as far as the image allows (gpytorch / botorch absent: LeafGP itself is not buildable here):
    TreeAgreementKernel.forward on RESIDENT CUDA tensors `calls` times in a row, as botorch's fitting loop evaluates
    the kernel (src/bark/tree_kernels/tree_model_kernel.py:16-23; src/bofire_mixed/surrogates/leafgp.py:61-77):
    per-call latency, and the share of calls served by the packed-forest cache (no host re-validation / upload)."""
    import torch

    from bark_amd import forest as bforest
    from bark_amd import synthetic
    from bark_amd.tree_kernels.tree_model_kernel import TreeAgreementKernel

    rows = []
    for n in (200, 1000, 2000):
        X, _y, bounds, ft = synthetic.mixed_problem(n, seed=n)
        forest = synthetic.sample_prior_forests(1, 50, bounds, ft, seed=n)[0]
        kern = TreeAgreementKernel(forest, ft)
        Xd = torch.from_numpy(X).cuda()
        kern.forward(Xd, Xd)
        torch.cuda.synchronize()
        h0 = dict(bforest.pack_cache_stats)
        t = time.perf_counter()
        for _ in range(calls):
            K = kern.forward(Xd, Xd)
            dg = kern.forward(Xd, Xd, diag=True)
        torch.cuda.synchronize()
        per = (time.perf_counter() - t) / calls
        h1 = dict(bforest.pack_cache_stats)
        assert K.is_cuda and dg.is_cuda and bool((K.diagonal() == dg).all())
        hits, misses = h1["hits"] - h0["hits"], h1["misses"] - h0["misses"]
        rows.append({"N": n, "calls": calls, "ms_per_forward_plus_diag": 1e3 * per,
                     "packed_forest_cache_hit_rate": hits / max(hits + misses, 1)})
    return {"note": "gpytorch-free harness; LeafGP / botorch fitting itself stays blocked on gpytorch's absence", "rows": rows}


def extras(args, wl, result, mll_host):
    """Rank 0, N = 1 only, all outside the timed region: Gram-kernel roofline, the other BASELINE configs,
    informational probes, and the CPU baseline."""
    import numpy as np
    import torch

    from bark_amd import _lib

    lib, N, m, d, B = wl.lib, wl.N, wl.m, wl.d, wl.B
    stream = wl.stream

    # ---- the standalone Gram kernel (forest.py:78-98 API path), HBM-write bound: 16 forests, full N x N fp64 output
    Bg = min(16, B)
    sub = _lib.PackInfo.from_buffer_copy(wl.pf.info)
    sub.B = Bg
    leaves = torch.empty((Bg, int(lib.bark_leaf_words(ctypes.byref(sub))), int(lib.bark_leaf_npad(N))),
                         dtype=torch.int32, device=wl.Xd.device)
    Kg = torch.empty((Bg, N, N), dtype=torch.float64, device=wl.Xd.device)
    reps = 8

    def probe(fn):
        fn()
        a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        z.record()
        torch.cuda.synchronize()
        return a.elapsed_time(z) / reps

    walk_ms = probe(lambda: _lib.check(lib.bark_leaf_codes_hip(_lib.ctx(), _lib.ptr(wl.pf.packed), ctypes.byref(sub), _lib.ptr(wl.Xd),
                                                             N, d, _lib.ptr(leaves), stream)))
    g_ms = probe(lambda: _lib.check(lib.bark_gram_from_leaves_hip(_lib.ptr(leaves), N, _lib.ptr(leaves), N, ctypes.byref(sub), None,
                                                                  None, None, _lib.ptr(Kg), N, N * N, stream)))
    g_bytes = Bg * (8.0 * N * N + 4.0 * m * 2 * N)  # SURVEY §8d bytes_gram per matrix
    result["roofline"]["gram_kernel"] = {
        "bound": "hbm", "forests": Bg, "algorithmic_bytes": g_bytes, "ms": g_ms,
        "kernel": "gram_kernel alone (HIP events over 8 launches); the leaf walk that feeds it is timed separately",
        "leaf_walk_ms": walk_ms,
        "leaf_code": "one-hot bits" if lib.bark_leaf_encoding(ctypes.byref(sub)) == 1 else "packed bytes",
        "leaf_code_words": int(lib.bark_leaf_words(ctypes.byref(sub))),
        "achieved_GBs": g_bytes / (g_ms * 1e-3) / 1e9, "peak_GBs": HBM_PEAK_GBS,
        "frac": g_bytes / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
    del Kg, leaves

    # ---- the other BASELINE configs and the small-batch regime (device time of the production call, HIP events)
    if not args.no_configs and N == 4096:
        cfgs = []

        def entry(name, w, reps_, graph=False):
            avg, med = w.device_ms(reps_)
            w.check()
            flops = w.B * (w.N**3 / 3.0 + (float(w.N) * w.N * w.C if w.C else 0.0))
            tf = flops / (med * 1e-3) / 1e12
            row = {"config": name, "N": w.N, "B": w.B, "C": w.C, "device_ms": med, "device_ms_mean": avg,
                   "evals_per_s": w.B / (med * 1e-3), "tflops": tf, "frac_of_f64_mfma_peak": tf / F64_MFMA_PEAK_TFLOPS}
            if graph:  # launch-bound sizes: the same call replayed from a hipGraph
                row["graph_replay_ms"] = w.graph_ms(reps_)
                w.check()
            cfgs.append(row)

        entry("c2: N=1024 d=8 m=50, single forest", Workload(1024, 8, m, 1, 1024, 0), 20, graph=True)
        entry("c3 stress variant: N=4096, 256 forests of 50 complete depth-5 trees (32 leaves per tree)",
              Workload(4096, d, m, 256, N, 0, problem="stress"), 3)
        entry("c4 per-GPU share: N=4096, 64 forests", Workload(4096, d, m, 64, N, 0), 5)
        entry("small batch: N=4096, 16 forests", Workload(4096, d, m, 16, N, 0), 5)
        entry("small batch: N=4096, 8 forests", Workload(4096, d, m, 8, N, 0), 5)
        # batch sizes off the multiples of 8 (the number of XCDs: the workgroup -> tile map deals them out as virtual matrices)
        entry("small batch: N=4096, 12 forests", Workload(4096, d, m, 12, N, 0), 5)
        entry("small batch: N=4096, 5 forests", Workload(4096, d, m, 5, N, 0), 5)
        entry("lone matrix: N=4096, 1 forest", Workload(4096, d, m, 1, N, 0), 10, graph=True)
        # the regime the reference itself runs in (BO with tens to hundreds of points: BASELINE configs[0] is N = 64)
        for n_small in (64, 256, 512):
            entry("small N: N=%d d=8 m=50, 256 forests" % n_small, Workload(n_small, 8, m, 256, n_small, 0), 20)
        torch.cuda.empty_cache()
        w5 = Workload(16384, 12, m, 1, 16384, 0, problem="mixed", include_scale=True)
        entry("c5 MLL: N=16384 mixed cat+int+cont, single forest", w5, 3)
        del w5
        torch.cuda.empty_cache()
        w5p = Workload(16384, 12, m, 1, 16384, 0, problem="mixed", C=10000)
        entry("c5 posterior: N=16384 mixed, 10^4 candidates (mu, var), single forest", w5p, 3)
        del w5p
        torch.cuda.empty_cache()
        # Gram kernel alone at c2 (BASELINE.md §3 asks for Gram GB/s per config)
        w2 = Workload(1024, 8, m, 1, 1024, 0)
        lv = torch.empty((1, int(lib.bark_leaf_words(w2.pf.info_ref)), int(lib.bark_leaf_npad(1024))), dtype=torch.int32,
                         device=wl.Xd.device)
        K2 = torch.empty((1, 1024, 1024), dtype=torch.float64, device=wl.Xd.device)
        _lib.check(lib.bark_leaf_codes_hip(_lib.ctx(), _lib.ptr(w2.pf.packed), w2.pf.info_ref, _lib.ptr(w2.Xd), 1024, 8,
                                           _lib.ptr(lv), stream))  # real leaf codes of this forest, not uninitialised memory
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for it in range(21):
            if it == 1:
                a0.record()
            _lib.check(lib.bark_gram_from_leaves_hip(_lib.ptr(lv), 1024, _lib.ptr(lv), 1024, w2.pf.info_ref, None, None, None,
                                                     _lib.ptr(K2), 1024, 1024 * 1024, stream))
        a1.record()
        torch.cuda.synchronize()
        g2 = a0.elapsed_time(a1) / 20
        cfgs.append({"config": "c2 Gram kernel alone: N=1024, single forest (launch-latency bound: 8.4 MB of output)",
                     "device_ms": g2, "achieved_GBs": (8.0 * 1024 * 1024 + 4.0 * m * 2048) / (g2 * 1e-3) / 1e9})
        del w2, lv, K2
        result["configs"] = cfgs

    if not args.no_configs and N == 4096:
        result["numpy_api_end_to_end"] = numpy_api_probe(wl)
        result["sampler_step"] = sampler_step_probe(args, wl)
        result["fitting_loop_harness"] = fitting_loop_probe()

    # ---- yardstick, not a baseline the metric is quoted against: the vendor's batched fp64 Cholesky (torch.linalg.cholesky ->
    # rocSOLVER / MAGMA) on 64 of this workload's matrices — the factorisation ALONE (no Gram generation, no solve, no MLL) beside
    # the sweep's whole evaluation of the same 64 (the `configs` row "c4 per-GPU share").  Nothing in the product calls it.
    if not args.no_configs and N == 4096:
        try:
            import bark_amd.forest as bf

            nb = 64
            Kv = bf.batched_forest_gram_matrix(wl.forests[:nb], wl.Xd, wl.Xd, wl.ft)
            Kv += (1e-6 + wl.noise_d[:nb])[:, None, None] * torch.eye(N, dtype=torch.float64, device=wl.Xd.device)
            torch.linalg.cholesky(Kv)
            v0, v1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            v0.record()
            for _ in range(3):
                torch.linalg.cholesky(Kv)
            v1.record()
            torch.cuda.synchronize()
            v_ms = v0.elapsed_time(v1) / 3
            result["vendor_cholesky_yardstick"] = {
                "what": "torch.linalg.cholesky (vendor batched fp64 potrf) on %d N=%d matrices of this workload: factorisation only" % (nb, N),
                "ms": v_ms, "tflops": nb * N**3 / 3.0 / (v_ms * 1e-3) / 1e12,
                "backend": str(torch.backends.cuda.preferred_linalg_library())}
            del Kv
        except Exception as exc:  # pragma: no cover - a torch build without a batched potrf
            result["vendor_cholesky_yardstick"] = {"error": repr(exc)}

    # ---- informational only: the leaf-space evaluation of the SAME MLLs (R x R system over the leaves instead
    # of the N x N matrix).  It does not do the Gram + Cholesky work the metric counts and is not part of `value`.
    pf = wl.pf
    lws = torch.empty(int(lib.bark_mll_leafspace_workspace_bytes(N, int(pf.info.max_bits), m, B, 0)), dtype=torch.uint8,
                      device=wl.Xd.device)
    lmll = torch.empty(B, dtype=torch.float64, device=wl.Xd.device)
    linfo = torch.empty(B, dtype=torch.int32, device=wl.Xd.device)
    l0, l1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    lreps = 5
    for it in range(lreps + 1):
        if it == 1:
            l0.record()
        _lib.check(lib.bark_mll_leafspace_hip(_lib.ctx(), _lib.ptr(pf.packed), pf.info_ref, _lib.ptr(wl.Xd), N, d, _lib.ptr(wl.yd),
                                              _lib.ptr(wl.noise_d), None, wl.flags, None, 0, _lib.ptr(lmll), None, None,
                                              _lib.ptr(linfo), _lib.ptr(lws), lws.numel(), B, stream))
    l1.record()
    torch.cuda.synchronize()
    l_ms = l0.elapsed_time(l1) / lreps
    leaf_local = lmll.cpu().numpy()
    result["leafspace_probe"] = {
        "note": "exact Woodbury/determinant-lemma evaluation over the forest's leaves; NOT the benchmark metric "
                "(no N x N Gram, no N^3/3 Cholesky)",
        "leaves_per_forest_max": int(pf.info.max_bits), "ms_per_%d_evals" % B: l_ms, "evals_per_s": B / (l_ms * 1e-3),
        "max_rel_diff_vs_dense": float(np.max(np.abs(leaf_local - mll_host) / np.abs(mll_host)))}
    del lws

    # ---- the oracle (checker) timed on this box's host cores on a bounded sample of the same workload
    if args.cpu_sample > 0:
        from oracle import oracle as orc

        cores = host_cores()
        ns = min(args.cpu_sample, B)
        F, noise = wl.forests, wl.noise
        try:  # LAPACK threads = the cores this process may really use (a pool sized to every core of a shared host
            # oversubscribes the container's CPU quota and runs slower than one thread)
            from threadpoolctl import threadpool_limits

            limit = threadpool_limits(limits=cores)
        except Exception:  # pragma: no cover
            limit = None
        t1 = time.perf_counter()
        ref = orc.batched_mll(F[:ns], noise[:ns], None, wl.X, wl.y, wl.ft, include_scale=False, include_2pi=True)
        cpu_s = time.perf_counter() - t1
        t2 = time.perf_counter()
        orc.batched_mll(F[:ns], noise[:ns], None, wl.X, wl.y, wl.ft, include_scale=False, include_2pi=True, cholesky=True)
        chol_s = time.perf_counter() - t2
        if limit is not None:
            limit.restore_original_limits()
        rel = float(np.max(np.abs(mll_host[:ns] - ref) / np.abs(ref)))
        assert np.allclose(mll_host[:ns], ref, rtol=1e-9, atol=1e-8), (mll_host[:ns], ref)
        single = None
        try:  # the reference's own configuration is single-threaded numba + whatever LAPACK threads numpy has
            from threadpoolctl import threadpool_limits

            with threadpool_limits(limits=1):
                t3 = time.perf_counter()
                orc.batched_mll(F[:1], noise[:1], None, wl.X, wl.y, wl.ft, include_scale=False, include_2pi=True)
                one_s = time.perf_counter() - t3
            single = {"value": 1.0 / one_s, "unit": "evals/s", "cores": 1,
                      "sample": "1 forest sample, every stage on one thread (LAPACK limited by threadpoolctl), %.1f s" % one_s}
        except Exception as exc:  # pragma: no cover
            single = {"error": repr(exc)}
        result["cpu_baseline"] = {
            "value": ns / cpu_s,
            "unit": "evals/s",
            "cores": cores,
            "kind": "port",
            "sample": "%d of the %d forest samples of this workload (N=%d): C leaf walk + N*N*m compare-count Gram "
                      "(1 thread, as the reference) + numpy.linalg.inv + slogdet (LAPACK on the %d usable cores), %.1f s"
                      % (ns, B, N, cores, cpu_s),
            "cholesky_variant_evals_per_s": ns / chol_s,
            "single_thread": single,
            "gpu_vs_oracle_max_rel_err": rel,
        }


def main():
    args = parse()
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        sys.exit(worker(args))
    sys.exit(launch(args))


if __name__ == "__main__":
    main()
