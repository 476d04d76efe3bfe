"""world_size-2 gloo test of the multi-GPU layout (shard forests, gather MLL) on CPU.
The per-rank evaluation is stood in by the oracle here; on the GPU box each rank runs the HIP path."""
import os
import socket

import numpy as np
import pytest

from bark_amd import synthetic
from bark_amd.distributed import shard_range


def test_shard_range_partitions_exactly():
    for total in (0, 1, 7, 256, 512, 513):
        for world in (1, 2, 3, 8):
            blocks = [shard_range(total, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            sizes = [h - l for l, h in blocks]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(512, 3, 8) == (192, 256)  # config c4: 64 samples per GPU
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, q):
    import torch
    import torch.distributed as dist

    from bark_amd.distributed import gather_mll, shard_range as sr
    from oracle import oracle as orc

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X, y, bounds, ft = synthetic.mixed_problem(48, seed=3)
    F = synthetic.sample_prior_forests(total, 20, bounds, ft, seed=30)
    noise = np.linspace(0.05, 0.2, total)
    lo, hi = sr(total, rank, world)
    local = orc.batched_mll(F[lo:hi], noise[lo:hi], None, X, y, ft, include_scale=False, include_2pi=True)
    full = gather_mll(torch.from_numpy(local), total)
    # posterior mixture over the union of the shards (one all-reduce of 2*C partial sums)
    from bark_amd.distributed import reduce_mixture

    cand, _, _, _ = synthetic.mixed_problem(11, seed=4)
    scale = np.linspace(0.8, 1.2, total)
    mu, var = orc.forest_predict((F[lo:hi], noise[lo:hi], scale[lo:hi]), (X, y), cand, ft)
    mix_mu, mix_var = reduce_mixture(torch.from_numpy(mu), torch.from_numpy(np.ascontiguousarray(var)), total)
    q.put((rank, full.numpy(), mix_mu.numpy(), mix_var.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [6, 5])
def test_gather_mll_world2_gloo(total):
    import torch.multiprocessing as mp

    from oracle import oracle as orc

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    results = {r[0]: r[1] for r in got}
    mixes = {r[0]: (r[2], r[3]) for r in got}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    X, y, bounds, ft = synthetic.mixed_problem(48, seed=3)
    F = synthetic.sample_prior_forests(total, 20, bounds, ft, seed=30)
    want = orc.batched_mll(F, np.linspace(0.05, 0.2, total), None, X, y, ft, include_scale=False, include_2pi=True)
    for rank in (0, 1):
        assert np.array_equal(results[rank], want)
    cand, _, _, _ = synthetic.mixed_problem(11, seed=4)
    noise, scale = np.linspace(0.05, 0.2, total), np.linspace(0.8, 1.2, total)
    mu, var = orc.forest_predict((F, noise, scale), (X, y), cand, ft)
    want_mu, want_var = orc.mixture_of_gaussians_as_normal(mu, var)
    for rank in (0, 1):
        assert np.allclose(mixes[rank][0], want_mu, rtol=1e-12, atol=1e-14)
        assert np.allclose(mixes[rank][1], want_var, rtol=1e-10, atol=1e-13)


def _run_bench(argv, env_extra=None, timeout=180):
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, *argv], cwd=root, env=env, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout  # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


@pytest.mark.parametrize("gpus", [1, 2])
def test_bench_launches_its_own_workers_from_a_plain_shell(gpus):
    """`python bench.py --gpus N` with no launcher around it: the parent (no torch, no GPU) starts N fresh workers,
    they rendezvous on 127.0.0.1, gather the per-rank MLL blocks and MAX-reduce the clock; rank 0 prints the line."""
    r = _run_bench(["bench.py", "--gpus", str(gpus), "--selftest-launcher"])
    assert r["ranks_reached"] == gpus and r["gather_ok"] and r["launched_by"] == "bench.py"
    assert r["scaling"] == "weak" and r["total"] == 256 * gpus and r["local"] == 256
    if gpus > 1:  # the second timed region of a weak run: BASELINE configs[3], 512 samples sharded over the ranks
        assert r["c4_strong"] == {"total": 512, "local": 256, "gather_ok": True, "scaling": "strong"}
        assert r["collective_backend"] == "gloo" and r["collective_backend_requested"] == "gloo" and r["rccl_ranks_seen"] is None
    else:
        assert r["c4_strong"] is None and r["collective_backend"] is None


def test_bench_rccl_failure_is_fatal_unless_the_fallback_is_asked_for():
    """RCCL cannot come up on this CPU-only box.  Default (--require-rccl): every worker exits non-zero, no JSON line.
    --no-require-rccl: the gather falls back to gloo inside the same worker processes and the line says so."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["BARK_BENCH_BACKEND"] = "nccl"
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--selftest-launcher"], cwd=root, env=env,
                         capture_output=True, text=True, timeout=180)
    assert out.returncode != 0 and "--require-rccl is in force" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    r = _run_bench(["bench.py", "--gpus", "2", "--selftest-launcher", "--no-require-rccl"], {"BARK_BENCH_BACKEND": "nccl"})
    assert r["collective_backend_requested"] == "nccl" and r["collective_backend"] == "gloo" and r["rccl_ranks_seen"] is None
    assert r["gather_ok"] and r["c4_strong"]["gather_ok"]


def test_bench_strong_scaling_shards_c4():
    """BASELINE configs[3]: 512 forest samples over the GPUs, contiguous blocks (forest.py:92-98 loop order)."""
    r = _run_bench(["bench.py", "--gpus", "2", "--total", "512", "--selftest-launcher"])
    assert r["scaling"] == "strong" and r["total"] == 512 and r["local"] == 256 and r["gather_ok"]


def test_bench_runs_as_a_worker_under_torch_distributed_run():
    """The form the driver uses for N > 1: an external launcher provides RANK / WORLD_SIZE."""
    r = _run_bench(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                    "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--selftest-launcher"])
    assert r["ranks_reached"] == 2 and r["gather_ok"] and r["launched_by"] == "external"


def test_bench_parent_never_imports_torch():
    """The launcher must not initialise the GPU before the workers exist (a forked/exec'd GPU context kills the box)."""
    import ast

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tree = ast.parse(open(os.path.join(root, "bench.py")).read())
    top = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
    names = {a.name.split(".")[0] for n in top if isinstance(n, ast.Import) for a in n.names}
    names |= {n.module.split(".")[0] for n in top if isinstance(n, ast.ImportFrom) and n.module}
    assert "torch" not in names and "bark_amd" not in names and "numpy" not in names
    launch = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "launch")
    assert not any(isinstance(n, (ast.Import, ast.ImportFrom)) for n in ast.walk(launch))


def test_unique_id_exchange_over_tcp():
    """The 128-byte RCCL id of the C-ABI communicator (bark_comm_unique_id) travels from rank 0 to the other ranks over a
    plain socket (bark_amd.distributed.exchange_unique_id): three ranks as threads, late listener included."""
    import threading
    import time

    from bark_amd.distributed import exchange_unique_id

    port, world = _free_port(), 3
    want = bytes(range(128))
    got = {}

    def rank_fn(r):
        if r == 0:
            time.sleep(0.3)  # the others are already retrying when rank 0 starts to listen
        got[r] = exchange_unique_id(r, world, "127.0.0.1", port, lambda: want, timeout=30.0)

    ts = [threading.Thread(target=rank_fn, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=60)
    assert got == {0: want, 1: want, 2: want}


def test_unique_id_is_not_handed_to_a_peer_without_the_job_token():
    """rank 0 used to send the id to whoever connected first.  A connection that does not present the job's
    token (a stray local client — here one that sends nothing and one that sends 32 wrong bytes) gets no byte of the id and
    does not use up a peer slot: the real rank 1, arriving later, is still served."""
    import socket
    import threading
    import time

    from bark_amd.distributed import exchange_unique_id

    port, world = _free_port(), 2
    want = bytes(range(128, 256))
    got, stray = {}, []

    def rank_fn(r):
        if r == 1:
            time.sleep(1.0)  # the strays come first
        got[r] = exchange_unique_id(r, world, "127.0.0.1", port, lambda: want, timeout=30.0)

    def stray_fn(payload):
        deadline = time.monotonic() + 10.0
        while time.monotonic() < deadline:
            try:
                with socket.create_connection(("127.0.0.1", port), timeout=2.0) as c:
                    if payload:
                        c.sendall(payload)
                    c.shutdown(socket.SHUT_WR)
                    c.settimeout(8.0)
                    stray.append(c.recv(128))
                    return
            except (ConnectionRefusedError, OSError):
                time.sleep(0.05)

    ts = [threading.Thread(target=rank_fn, args=(0,)), threading.Thread(target=stray_fn, args=(b"",)),
          threading.Thread(target=stray_fn, args=(b"x" * 32,)), threading.Thread(target=rank_fn, args=(1,))]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=60)
    assert got == {0: want, 1: want}
    assert stray == [b"", b""], stray
