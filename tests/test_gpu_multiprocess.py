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
    g.close()
