"""Two processes, one GPU: each rank evaluates its shard of forest samples with the HIP path and the
shards are gathered (gloo on CPU tensors here; RCCL on device tensors in bench.py).  Covers the N > 1
flow of SURVEY §8e on the single-GPU test box."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, q):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import bark_amd.fitting as fit
    from bark_amd import synthetic
    from bark_amd.distributed import gather_mll, shard_range

    X, y, bounds, ft = synthetic.mixed_problem(700, seed=5)
    F = synthetic.sample_prior_forests(total, 50, bounds, ft, seed=50)
    noise = np.linspace(0.05, 0.2, total)
    lo, hi = shard_range(total, rank, world)
    local = fit.batched_mll(F[lo:hi], noise[lo:hi], None, X, y, ft, include_scale=False, include_2pi=True,
                            return_device=True)
    full = gather_mll(local.cpu(), total)
    q.put((rank, full.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_share_one_gpu():
    import torch.multiprocessing as mp

    from bark_amd import synthetic
    from oracle import oracle as orc

    total = 7  # ragged split: 4 + 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    X, y, bounds, ft = synthetic.mixed_problem(700, seed=5)
    F = synthetic.sample_prior_forests(total, 50, bounds, ft, seed=50)
    want = orc.batched_mll(F, np.linspace(0.05, 0.2, total), None, X, y, ft, include_scale=False, include_2pi=True)
    assert np.array_equal(results[0], results[1])
    assert np.allclose(results[0], want, rtol=1e-9, atol=1e-8)


def test_rccl_through_the_c_abi_world_of_one():
    """bark_comm_* / bark_allgather_mll / bark_allreduce_f64 (include/bark_hip.h): RCCL resolved at run time, a communicator
    built from the C ABI alone.  One GPU per box here, so the world has one rank (two ranks cannot share a device under
    RCCL): the gather returns the block, sum and max leave the buffer as it is, ragged totals keep their order."""
    import torch

    from bark_amd.distributed import RcclGroup

    g = RcclGroup(0, 1, 0)
    local = torch.arange(7, dtype=torch.float64, device="cuda") * 1.5
    out = g.gather_mll(local, 7)
    torch.cuda.synchronize()
    assert torch.equal(out, local)
    buf = torch.tensor([2.0, -3.0, 5.5], dtype=torch.float64, device="cuda")
    g.all_reduce(buf)
    g.all_reduce(buf, "max")
    g.barrier()
    assert buf.tolist() == [2.0, -3.0, 5.5]
    # raw pointers cross the ABI: anything but a contiguous float64 tensor on the communicator's GPU is refused up front
    with pytest.raises(ValueError):
        g.all_reduce(buf.float())
    with pytest.raises(ValueError):
        g.all_reduce(buf.cpu())
    with pytest.raises(ValueError):
        g.all_reduce(torch.zeros((4, 2), dtype=torch.float64, device="cuda")[:, 0])
    assert torch.cuda.current_device() == 0
    g.close()
    with pytest.raises(RuntimeError):
        g.all_reduce(buf)


_ABI_VS_TORCH = r"""
import os, sys
sys.path.insert(0, {root!r})
import torch
import torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
dist.init_process_group("nccl", rank=rank, world_size=world)
from bark_amd.distributed import RcclGroup, shard_range
g = RcclGroup(rank, world, rank)
total = 11  # ragged: 6 + 5
lo, hi = shard_range(total, rank, world)
local = (torch.arange(lo, hi, dtype=torch.float64, device="cuda") + 0.25) * (rank + 1)
via_abi = g.gather_mll(local, total)
even = torch.full((4,), float(rank), dtype=torch.float64, device="cuda")  # equal blocks: against dist.all_gather itself
pieces = [torch.empty(4, dtype=torch.float64, device="cuda") for _ in range(world)]
dist.all_gather(pieces, even)
assert torch.equal(g.gather_mll(even, 4 * world), torch.cat(pieces))
ref = torch.cat([(torch.arange(*shard_range(total, r, world), dtype=torch.float64, device="cuda") + 0.25) * (r + 1) for r in range(world)])
assert torch.equal(via_abi, ref), (via_abi, ref)
a = torch.full((5,), float(rank + 1), dtype=torch.float64, device="cuda")
b = a.clone()
g.all_reduce(a)
dist.all_reduce(b)
assert torch.equal(a, b)
g.all_reduce(a, "max")
g.barrier()
g.close()
dist.destroy_process_group()
print("abi-vs-torch ok", rank)
"""


def test_rccl_abi_matches_torch_distributed_two_ranks():
    """the hand-declared RCCL ABI (csrc/comm.cpp) against torch.distributed's nccl backend with TWO ranks on two
    GPUs — the first place the declarations can be wrong in a way a world of one does not show.  Needs a node with two
    GPUs: skipped on the one-GPU boxes this suite normally runs on (the RCCL leg has never run with more than one rank)."""
    import subprocess
    import sys

    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs on one node")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, "-c", _ABI_VS_TORCH.format(root=root)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-800:] for o in outs]
