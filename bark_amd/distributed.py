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


# ----------------------------------------------------------------------------------------------------------------------
# The same two exchanges through the C ABI (include/bark_hip.h: bark_comm_* / bark_allgather_mll / bark_allreduce_f64):
# RCCL without torch.distributed.  Opt-in (`bench.py` with BARK_BENCH_BACKEND=abi); the default stays torch.distributed.
# ----------------------------------------------------------------------------------------------------------------------
def _id_token() -> bytes:
    """32 bytes every rank of ONE job derives alike and a stray local process does not know: $BARK_RCCL_ID_TOKEN (bench.py's
    launcher draws 128 random bits per job and exports them to its workers), else the job's rendezvous identity (torchrun's
    run id + MASTER_ADDR:MASTER_PORT — not secret, but not what an unrelated connection sends either)."""
    import hashlib
    import os

    tok = os.environ.get("BARK_RCCL_ID_TOKEN") or "|".join(
        os.environ.get(k, "") for k in ("TORCHELASTIC_RUN_ID", "MASTER_ADDR", "MASTER_PORT"))
    return hashlib.sha256(("bark-rccl-id:" + tok).encode()).digest()


def exchange_unique_id(rank: int, world: int, addr: str, port: int, make_id, timeout: float = 120.0) -> bytes:
    """Rank 0 makes the 128-byte RCCL unique id (`make_id()`) and hands it to the other ranks over TCP (they connect to
    addr:port, retrying until rank 0 listens).  Plain sockets: no process group is needed to build one.  A peer has to
    present the job's token (`_id_token`) first: a connection that does not — a port scanner, another job's rank on the same
    port — is closed without the id and does not count towards the world - 1 peers rank 0 serves."""
    import hmac
    import socket
    import time

    if world == 1:
        return make_id()
    token = _id_token()
    if rank == 0:
        uid = make_id()
        deadline = time.monotonic() + timeout
        with socket.socket() as srv:
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(world + 8)
            served = 0
            while served < world - 1:
                left = deadline - time.monotonic()
                if left <= 0:
                    raise TimeoutError("exchange_unique_id: %d of %d peers presented the job's token in %.0f s"
                                       % (served, world - 1, timeout))
                srv.settimeout(left)
                conn, _peer = srv.accept()
                with conn:
                    conn.settimeout(5.0)
                    try:
                        got = b""
                        while len(got) < len(token):
                            part = conn.recv(len(token) - len(got))
                            if not part:
                                break
                            got += part
                        if hmac.compare_digest(got, token):
                            conn.sendall(uid)
                            served += 1
                    except OSError:
                        pass  # a stray or stalled connection: dropped, the id stays here
        return uid
    deadline = time.monotonic() + timeout
    while True:
        try:
            with socket.create_connection((addr, port), timeout=5.0) as c:
                c.sendall(token)
                buf = b""
                while len(buf) < 128:
                    part = c.recv(128 - len(buf))
                    if not part:
                        raise ConnectionError("rank 0 closed the connection before the id was complete")
                    buf += part
                return buf
        except (ConnectionRefusedError, ConnectionError, OSError):
            if time.monotonic() > deadline:
                raise
            time.sleep(0.05)


class RcclGroup:
    """One RCCL communicator per process, built and used through the C ABI only (one process per GPU)."""

    def __init__(self, rank: int, world: int, device: int, addr: str = "127.0.0.1", port: int | None = None):
        import ctypes
        import os

        from . import _lib

        self._lib, self.rank, self.world = _lib, rank, world
        self._comm = None
        if port is None:
            # next to the launcher's rendezvous port, so that two jobs on one host do not meet on a fixed number (the id is
            # handed only to peers that present the job's token, exchange_unique_id; still keep addr on the loopback / a
            # private interface)
            port = int(os.environ.get("BARK_RCCL_ID_PORT", 0)) or (int(os.environ.get("MASTER_PORT", 29500)) + 33) % 65536 or 29533
        lib = _lib.lib()

        def make_id() -> bytes:
            buf = ctypes.create_string_buffer(128)
            _lib.check(lib.bark_comm_unique_id(buf))
            return buf.raw

        uid = exchange_unique_id(rank, world, addr, port, make_id)
        comm = ctypes.c_void_p()
        _lib.check(lib.bark_comm_create(ctypes.c_char_p(uid), rank, world, device, ctypes.byref(comm)))
        self._comm, self._device = comm, device

    def _checked(self, t, what: str):
        """float64, contiguous, on this communicator's GPU — raw pointers cross the ABI, so nothing else may."""
        import torch

        if not isinstance(t, torch.Tensor) or t.dtype != torch.float64 or not t.is_cuda or not t.is_contiguous():
            raise ValueError(f"{what}: a contiguous float64 CUDA tensor is required")
        if t.device.index != self._device:
            raise ValueError(f"{what}: tensor on cuda:{t.device.index}, communicator on cuda:{self._device}")
        if self._comm is None:
            raise RuntimeError("RcclGroup is closed")
        return t

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter teardown
            pass

    def gather_mll(self, local, total: int):
        """`gather_mll` above over this communicator: per-rank blocks (sizes from shard_range) -> (total,) on every rank."""
        import torch

        L, lib = self._lib, self._lib.lib()
        sizes = [shard_range(total, r, self.world) for r in range(self.world)]
        lo, hi = sizes[self.rank]
        if local.shape[0] != hi - lo:
            raise ValueError(f"rank {self.rank} holds {local.shape[0]} values, expected {hi - lo}")
        widest = max(h - l for l, h in sizes)
        send = self._checked(local.contiguous(), "gather_mll")
        if send.shape[0] != widest:
            send = torch.zeros(widest, dtype=torch.float64, device=local.device)
            send[: hi - lo] = local
        out = torch.empty(widest * self.world, dtype=torch.float64, device=local.device)
        L.check(lib.bark_allgather_mll(self._comm, L.ptr(send), widest, L.ptr(out), L.stream_ptr()))
        if all(h - l == widest for l, h in sizes):
            return out
        return torch.cat([out[r * widest: r * widest + (h - l)] for r, (l, h) in enumerate(sizes)])

    def all_reduce(self, t, op: str = "sum"):
        """In place over all ranks (float64 device tensor): 'sum' (mixture moments) or 'max' (the bench's clock)."""
        L = self._lib
        self._checked(t, "all_reduce")
        L.check(L.lib().bark_allreduce_f64(self._comm, L.ptr(t), t.numel(), 1 if op == "max" else 0, L.stream_ptr()))
        return t

    def barrier(self):
        import torch

        one = torch.zeros(1, dtype=torch.float64, device=torch.device("cuda", self._device))
        self.all_reduce(one)
        torch.cuda.synchronize()

    def close(self):
        if self._comm:
            self._lib.lib().bark_comm_destroy(self._comm)
            self._comm = None
