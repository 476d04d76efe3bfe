"""Device-resident state for the sampler's per-tree Metropolis step (reference:
src/bark/fitting/bark_sampler.py:217-264, `_step_bark_sampler`).

The reference keeps `cur_K_inv`, `cur_K_logdet`, `cur_mll` per chain and, for every tree proposal, chains

    low_rank_inv_update(subtract old leaf vectors) -> low_rank_det_update -> low_rank_inv_update(add new)
    -> low_rank_det_update -> mll                                     (bark_sampler.py:242-257)

i.e. about nine passes over the N x N inverse before the accept/reject.  `ChainState.propose` gets the same
`new_mll` from ONE pass (Y = K_inv [U_old U_new]) plus (r_old + r_new)^2 algebra, and `accept` rewrites
K_inv only for accepted proposals (one read + write pass).  The host keeps the control flow (proposal,
RNG, accept/reject), exactly as in the reference; only the linear algebra lives on the GPU.
This is an additive API (the drop-in functions of `quick_inverse` remain available)."""

from __future__ import annotations

import numpy as np

from .. import _lib
import ctypes

from ..forest import PackedForest, _as_nodes, _check_categorical, _feat_types, _is_torch, _points
from .mll import _run_leafspace, batched_kernel_inverse


def _dev64(a):
    import torch

    t = _lib.to_device(a.detach() if _is_torch(a) else np.asarray(a, dtype=np.float64))
    return t.to(torch.float64).contiguous()


class ChainState:
    """K_inv (N, N), log|K|, y'K_inv y and the targets y of one chain, resident in HBM."""

    def __init__(self, K_inv, K_logdet, y):
        import torch

        self.K_inv = _dev64(K_inv)
        self.y = _dev64(y).reshape(-1).contiguous()
        self.N = self.y.shape[0]
        if self.K_inv.shape != (self.N, self.N):
            raise ValueError(f"K_inv is {tuple(self.K_inv.shape)}, y has {self.N} rows")
        self.logdet = float(K_logdet)
        q = torch.empty(1, dtype=torch.float64, device=self.K_inv.device)
        _lib.check(_lib.lib().bark_quadform_hip(_lib.ptr(self.K_inv), _lib.ptr(self.y), self.N, _lib.ptr(q),
                                                _lib.stream_ptr()))
        self.quad = float(q.item())
        self._pending = None
        self._ws = {}  # rank -> workspace tensor, reused across proposals
        self._X_seen = None  # (caller's X object, validated device tensor) of the last propose_tree
        self._scalars = torch.empty(2, dtype=torch.float64, device=self.K_inv.device)

    def _workspace(self, r: int):
        import torch

        ws = self._ws.get(r)
        if ws is None:
            nbytes = int(_lib.lib().bark_tree_swap_workspace_bytes(self.N, r))  # >= the plain low-rank layout
            ws = self._ws[r] = torch.empty(nbytes, dtype=torch.uint8, device=self.K_inv.device)
        return ws

    @classmethod
    def from_forest(cls, forest, noise, scale, X, y, feat_types):
        """Initial state of a chain: bark_sampler.py:153-162 (scale * K + (1e-6 + noise) I, inverse, logdet)."""
        nodes = np.asarray(forest)
        K_inv, _, logdet = batched_kernel_inverse(nodes[None], [noise], [scale], X, y, feat_types, no_null=False,
                                                  return_device=True)
        return cls(K_inv[0], float(logdet[0].item()), y)

    def propose_noise_scale(self, forest, new_noise: float, new_scale: float, X, feat_types) -> float:
        """MLL the chain would have with the proposed (noise, scale) — the second half of `_step_bark_sampler`
        (bark_sampler.py:266-272), where the reference rebuilds inv + slogdet of the N x N matrix.  Evaluated in
        leaf space (R x R system, include/bark_hip.h) without touching K_inv; `accept()` rebuilds the resident
        inverse, also in leaf space.  `forest` is the chain's current (m, node_limit) forest."""
        nodes = np.asarray(forest)
        val = _run_leafspace(nodes[None], [new_noise], [new_scale], X, self.y, feat_types, _lib.MLL_INCLUDE_SCALE)
        self._pending = ("noise_scale", nodes, float(new_noise), float(new_scale), X, feat_types)
        return float(val[0].item())

    @property
    def mll(self) -> float:
        """quick_inverse.py:37-38."""
        return 0.5 * (-self.quad - self.logdet)

    def propose(self, cur_leaf_vectors, new_leaf_vectors) -> float:
        """MLL the chain would have after swapping the old tree's (scaled) leaf vectors for the new tree's
        (bark_sampler.py:238-256).  Does not modify the state; call `accept()` to commit."""
        import torch

        lib = _lib.lib()
        U_old, U_new = _dev64(cur_leaf_vectors), _dev64(new_leaf_vectors)
        if U_old.shape[0] != self.N or U_new.shape[0] != self.N:
            raise ValueError("leaf vectors must have N rows")
        r_old, r_new = U_old.shape[1], U_new.shape[1]
        if r_old + r_new > 64:
            raise ValueError(f"tree swap supports at most 64 leaf vectors in total (got {r_old + r_new})")
        U = torch.cat([U_old, U_new], dim=1).contiguous()
        r = r_old + r_new
        ws, scalars = self._workspace(r), self._scalars
        _lib.check(lib.bark_lowrank_swap_eval_hip(_lib.ptr(self.K_inv), self.N, _lib.ptr(U), r_old, r_new,
                                                  _lib.ptr(self.y), _lib.ptr(scalars), _lib.ptr(ws), ws.numel(),
                                                  _lib.stream_ptr()))
        dquad, dlogdet = (float(v) for v in scalars.cpu().numpy())
        self._pending = (ws, r, self.quad - dquad, self.logdet + dlogdet)
        return 0.5 * (-(self.quad - dquad) - (self.logdet + dlogdet))

    def accept(self) -> None:
        """Commit the last proposal: K_inv <- K_inv - Y (C+G)^-1 Y' (bark_sampler.py:259-264)."""
        if self._pending is None:
            raise RuntimeError("accept() without a pending propose()")
        if self._pending[0] == "noise_scale":
            _, nodes, noise, scale, X, feat_types = self._pending
            K_inv, K_inv_y, logdet = batched_kernel_inverse(nodes[None], [noise], [scale], X, self.y, feat_types,
                                                            no_null=False, return_device=True, method="leafspace")
            self.K_inv = K_inv[0]
            self.quad = float((K_inv_y[0] @ self.y).item())
            self.logdet = float(logdet[0].item())
            self._pending = None
            return
        ws, r, quad, logdet = self._pending
        _lib.check(_lib.lib().bark_lowrank_swap_apply_hip(_lib.ptr(self.K_inv), self.N, r, _lib.ptr(ws),
                                                          _lib.ptr(self.K_inv), _lib.stream_ptr()))
        self.quad, self.logdet = quad, logdet
        self._pending = None

    def propose_tree(self, old_nodes, new_nodes, X, feat_types, scale: float, m: int) -> float:
        """bark_sampler.py:233-256 from the two trees themselves: both are walked on the GPU and their one-hot
        leaf code, scaled by s_sqrtm = sqrt(scale / m), is the [U_old U_new] of `propose` (one column per leaf;
        leaves no point reaches give zero columns, which change nothing).  `accept()` commits as usual."""
        lib = _lib.lib()
        ft = _feat_types(feat_types)
        if self._X_seen is None or self._X_seen[0] is not X:
            Xd, _ = _points(X, ft.shape[0])
            _check_categorical(Xd, ft)
            self._X_seen = (X, Xd)
        Xd = self._X_seen[1]
        if Xd.shape[0] != self.N:
            raise ValueError(f"X has {Xd.shape[0]} rows, the chain has {self.N} points")
        old, new = _as_nodes(old_nodes, 1), _as_nodes(new_nodes, 1)
        if old.ndim != 1 or new.ndim != 1 or old.shape != new.shape:
            raise ValueError(f"trees must be (node_limit,) records of one container, got {old.shape} and {new.shape}")
        info_old = _lib.PackInfo()
        _lib.check(lib.bark_forest_pack_info(_lib.ptr(old), 1, 1, old.shape[0], _lib.ptr(ft), ft.shape[0],
                                             ctypes.byref(info_old)))
        pf = PackedForest(np.stack([old, new])[None], ft)
        r, r_old = int(pf.info.max_bits), int(info_old.max_bits)
        if r > 64:
            raise ValueError(f"tree swap supports at most 64 leaves in total (got {r})")
        ws, scalars = self._workspace(r), self._scalars
        _lib.check(lib.bark_tree_swap_eval_hip(_lib.ptr(self.K_inv), self.N, _lib.ptr(pf.packed), pf.info_ref, _lib.ptr(Xd),
                                               Xd.shape[1], r_old, float(np.sqrt(scale / m)), _lib.ptr(self.y),
                                               _lib.ptr(scalars), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()))
        dquad, dlogdet = (float(v) for v in scalars.cpu().numpy())
        self._pending = (ws, r, self.quad - dquad, self.logdet + dlogdet)
        return 0.5 * (-(self.quad - dquad) - (self.logdet + dlogdet))


class ChainBatch:
    """`ChainState` for several independent chains of the sampler (bark_sampler.py:147): K_inv (nc, N, N) resident,
    the per-tree proposals of all chains evaluated by ONE library call — the chain index is a grid dimension of
    every kernel, so the call costs one launch sequence — and one host synchronisation.  The chains share X and y (as in the reference)."""

    def __init__(self, K_inv, K_logdet, y):
        import torch

        self.K_inv = _dev64(K_inv)
        if self.K_inv.ndim != 3 or self.K_inv.shape[1] != self.K_inv.shape[2]:
            raise ValueError(f"K_inv must be (chains, N, N), got {tuple(self.K_inv.shape)}")
        self.nc, self.N = int(self.K_inv.shape[0]), int(self.K_inv.shape[1])
        if not 1 <= self.nc <= 64:
            raise ValueError("1 to 64 chains")
        self.y = _dev64(y).reshape(-1).contiguous()
        if self.y.shape[0] != self.N:
            raise ValueError(f"y has {self.y.shape[0]} rows, K_inv is {self.N} x {self.N}")
        self.logdet = np.asarray(K_logdet, dtype=np.float64).reshape(-1).copy()
        if self.logdet.shape[0] != self.nc:
            raise ValueError("one log-determinant per chain")
        self.quad = (torch.einsum("i,bij,j->b", self.y, self.K_inv, self.y)).cpu().numpy()
        self._scalars = torch.empty((self.nc, 2), dtype=torch.float64, device=self.K_inv.device)
        self._ws = {}
        self._pending = None
        self._X_seen = None

    @classmethod
    def from_forests(cls, forests, noise, scale, X, y, feat_types, method: str = "dense"):
        """Initial state of every chain (bark_sampler.py:153-162); forests (chains, m, node_limit)."""
        nodes = np.asarray(forests)
        K_inv, _, logdet = batched_kernel_inverse(nodes, noise, scale, X, y, feat_types, no_null=False,
                                                  return_device=True, method=method)
        return cls(K_inv, logdet.cpu().numpy(), y)

    @property
    def mll(self) -> np.ndarray:
        """quick_inverse.py:37-38 for every chain."""
        return 0.5 * (-self.quad - self.logdet)

    def propose_trees(self, old_trees, new_trees, X, feat_types, scale, m: int) -> np.ndarray:
        """bark_sampler.py:233-256 for one tree per chain: old_trees / new_trees (chains, node_limit) records,
        scale (chains,) -> the (chains,) MLL values the chains would have.  `accept(mask)` commits."""
        import torch

        lib = _lib.lib()
        ft = _feat_types(feat_types)
        if self._X_seen is None or self._X_seen[0] is not X:
            Xd, _ = _points(X, ft.shape[0])
            _check_categorical(Xd, ft)
            self._X_seen = (X, Xd)
        Xd = self._X_seen[1]
        if Xd.shape[0] != self.N:
            raise ValueError(f"X has {Xd.shape[0]} rows, the chains have {self.N} points")
        old, new = _as_nodes(old_trees, 2), _as_nodes(new_trees, 2)
        if old.shape != new.shape or old.ndim != 2 or old.shape[0] != self.nc:
            raise ValueError(f"trees must be (chains, node_limit) records, got {old.shape} and {new.shape}")
        scale = np.broadcast_to(np.asarray(scale, dtype=np.float64).reshape(-1), (self.nc,))
        r_old = np.empty(self.nc, dtype=np.int64)
        info_one = _lib.PackInfo()
        for b in range(self.nc):  # leaves of each old tree alone = the split between removed and added columns
            _lib.check(lib.bark_forest_pack_info(_lib.ptr(np.ascontiguousarray(old[b])), 1, 1, old.shape[1], _lib.ptr(ft),
                                                 ft.shape[0], ctypes.byref(info_one)))
            r_old[b] = info_one.max_bits
        pf = PackedForest(np.stack([old, new], axis=1), ft)  # (chains, 2, L): one [old, new] pair per chain
        r = int(pf.info.max_bits)
        if r > 64:
            raise ValueError(f"tree swap supports at most 64 leaves in total (got {r})")
        ws = self._ws.get(r)
        if ws is None:
            nbytes = int(lib.bark_tree_swap_chains_workspace_bytes(self.N, r, self.nc, None))
            ws = self._ws[r] = torch.empty(nbytes, dtype=torch.uint8, device=self.K_inv.device)
        s_sqrtm = np.ascontiguousarray(np.sqrt(scale / m))
        _lib.check(lib.bark_tree_swap_eval_chains_hip(_lib.ptr(self.K_inv), self.N, self.nc, _lib.ptr(pf.packed), pf.info_ref,
                                                      _lib.ptr(Xd), Xd.shape[1], _lib.ptr(r_old), _lib.ptr(s_sqrtm),
                                                      _lib.ptr(self.y), _lib.ptr(self._scalars), _lib.ptr(ws), ws.numel(),
                                                      _lib.stream_ptr()))
        sc = self._scalars.cpu().numpy()
        quad, logdet = self.quad - sc[:, 0], self.logdet + sc[:, 1]
        self._pending = (ws, r, quad, logdet)
        return 0.5 * (-quad - logdet)

    def accept(self, mask) -> None:
        """Commit the pending proposals of the chains where `mask` is true (bark_sampler.py:259-264)."""
        if self._pending is None:
            raise RuntimeError("accept() without a pending propose_trees()")
        ws, r, quad, logdet = self._pending
        mask = np.ascontiguousarray(np.broadcast_to(np.asarray(mask).reshape(-1), (self.nc,)), dtype=np.int32)
        _lib.check(_lib.lib().bark_lowrank_swap_apply_chains_hip(_lib.ptr(self.K_inv), self.N, self.nc, r, _lib.ptr(mask),
                                                                 _lib.ptr(ws), ws.numel(), _lib.stream_ptr()))
        keep = mask != 0
        self.quad = np.where(keep, quad, self.quad)
        self.logdet = np.where(keep, logdet, self.logdet)
        self._pending = None
