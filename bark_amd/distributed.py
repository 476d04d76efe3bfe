"""Multi-GPU layout of the hot path: independent forest samples shard across ranks.

Each MCMC forest sample is an independent evaluation, so the B samples are cut into
contiguous blocks, one per rank (one process per GPU); X, y and feat_types are replicated.  There
is no data-path collective: the only exchange is the final all-gather of the (B,) log-likelihoods
(8 B per sample) over RCCL/xGMI (`nccl` backend) — or gloo on CPU in tests.
"""

from __future__ import annotations


def shard_range(total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous block [lo, hi) of `total` samples owned by `rank`; sizes differ by at most one."""
    if not (0 <= rank < world) or total < 0:
        raise ValueError(f"bad shard request total={total} rank={rank} world={world}")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_mll(local, total: int, group=None):
    """All-gather per-rank MLL blocks (sizes from shard_range) into the full (total,) vector on
    every rank.  `local` is a 1-D float64 torch tensor on the backend's device."""
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized():
        if local.shape[0] != total:
            raise ValueError("not distributed: local block must be the whole vector")
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = [shard_range(total, r, world) for r in range(world)]
    lo, hi = sizes[rank]
    if local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} holds {local.shape[0]} values, expected {hi - lo}")
    widest = max(h - l for l, h in sizes)
    if all(h - l == widest for l, h in sizes):
        out = torch.empty(total, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    padded = torch.zeros(widest, dtype=local.dtype, device=local.device)
    padded[: hi - lo] = local
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[: h - l] for p, (l, h) in zip(parts, sizes)])


def reduce_mixture(mu_local, var_local, total: int, group=None):
    """Mixture-of-Gaussians moments over ALL forest samples when each rank holds the posterior (mu, var) of
    its own shard of samples (tree_gps.py:116-131 applied to the union of the shards):

        E[Y] = (1/total) sum_b mu_b ;   Var[Y] = (1/total) sum_b (var_b + mu_b^2) - E[Y]^2

    One all-reduce(sum) of 2*C float64 partial sums (160 KB at C = 10^4) instead of gathering (B, C) arrays.
    `mu_local`, `var_local`: (B_local, C) torch tensors on the backend's device.  On the GPU the partial sums and the
    final moments come from bark_mixture_partial_hip / bark_mixture_finish_hip; CPU tensors (gloo tests) take the
    same two steps in torch.  `group=False`: no collective (single process)."""
    import torch
    import torch.distributed as dist

    B, C = int(mu_local.shape[0]), int(mu_local.shape[1])
    on_gpu = mu_local.is_cuda
    if on_gpu:
        from . import _lib

        mu_c, var_c = mu_local.contiguous(), var_local.contiguous()
        partial = torch.empty((2, C), dtype=torch.float64, device=mu_local.device)
        _lib.check(_lib.lib().bark_mixture_partial_hip(_lib.ptr(mu_c), _lib.ptr(var_c), B, C, _lib.ptr(partial),
                                                       _lib.stream_ptr()))
    else:
        partial = torch.stack([mu_local.sum(dim=0), (var_local + mu_local**2).sum(dim=0)])
    if group is not False and dist.is_available() and dist.is_initialized():
        dist.all_reduce(partial, op=dist.ReduceOp.SUM, group=group)
    if on_gpu:
        mean = torch.empty(C, dtype=torch.float64, device=mu_local.device)
        var = torch.empty(C, dtype=torch.float64, device=mu_local.device)
        _lib.check(_lib.lib().bark_mixture_finish_hip(_lib.ptr(partial), float(total), C, _lib.ptr(mean), _lib.ptr(var),
                                                      _lib.stream_ptr()))
        return mean, var
    mean = partial[0] / total
    return mean, partial[1] / total - mean**2
