"""Device-resident state for the sampler's per-tree Metropolis step (reference:
src/bark/fitting/bark_sampler.py:217-264, `_step_bark_sampler`).

The reference keeps `cur_K_inv`, `cur_K_logdet`, `cur_mll` per chain and, for every tree proposal, chains

    low_rank_inv_update(subtract old leaf vectors) -> low_rank_det_update -> low_rank_inv_update(add new)
    -> low_rank_det_update -> mll                                     (bark_sampler.py:242-257)

i.e. about nine passes over the N x N inverse before the accept/reject.  `ChainState.propose` gets the same
`new_mll` from ONE pass (Y = K_inv [U_old U_new]) plus (r_old + r_new)^2 algebra, and `accept` rewrites
K_inv only for accepted proposals (one read + write pass).  `ChainBatch.sweep_trees` runs a whole sweep over the
trees of several chains with the accept/reject decided on the device (one read-back per sweep).  Proposals, RNG and
the forest container stay on the host, exactly as in the reference; only the linear algebra lives on the GPU.
This is an additive API (the drop-in functions of `quick_inverse` remain available)."""

from __future__ import annotations

import ctypes

import numpy as np

from .. import _lib
from ..forest import _as_nodes, _feat_types, _is_torch, _points, _raise_on_categorical_fault, leaf_vectors_device, packed_forest
from .mll import _run_leafspace, batched_kernel_inverse

MAX_RANK = 64  # leaf vectors one fused update can carry (lowrank.hip LR_MAX)


def _dev64(a):
    import torch

    t = _lib.to_device(a.detach() if _is_torch(a) else np.asarray(a, dtype=np.float64))
    return t.to(torch.float64).contiguous()


def _quadform(K_inv, y) -> float:
    """y' K_inv y (quick_inverse.py:38) -> host float."""
    import torch

    q = torch.empty(1, dtype=torch.float64, device=K_inv.device)
    _lib.check(_lib.lib().bark_quadform_hip(_lib.ptr(K_inv), _lib.ptr(y), int(y.shape[0]), _lib.ptr(q), _lib.stream_ptr()))
    return float(q.item())


def _raise_if_singular(ws, N: int, r: int):
    flag = ctypes.c_int32(0)
    _lib.check(_lib.lib().bark_lowrank_status_hip(_lib.ptr(ws), N, r, ctypes.byref(flag), _lib.stream_ptr()))
    if flag.value:
        raise np.linalg.LinAlgError(f"Singular matrix in the rank-{r} update (pivot {flag.value})")  # quick_inverse.py:19,31


def _tree_leaves(nodes1, Xd, ft):
    """(N,) device leaf indices of one tree (forest.py:50-55)."""
    import torch

    pf = packed_forest(nodes1[None, None], ft)
    out = torch.empty((1, Xd.shape[0], 1), dtype=torch.int32, device=Xd.device)
    _lib.check(_lib.lib().bark_leaf_indices_hip(_lib.ctx(), _lib.ptr(pf.packed), pf.info_ref, _lib.ptr(Xd), Xd.shape[0],
                                                Xd.shape[1], _lib.ptr(out), _lib.stream_ptr()))
    return out.reshape(-1)


def _reached_leaf_vectors(old, new, Xd, ft, s: float):
    """The scaled leaf vectors of bark_sampler.py:233-236 with one column per REACHED leaf (get_leaf_vectors,
    forest.py:70-75): -> (U, None, r_old, r_new) with U = [U_old U_new] in one (N, r_old + r_new) buffer when that fits
    one fused update, else (U_old, U_new, r_old, r_new) as two matrices."""
    import torch

    lo, ln = _tree_leaves(old, Xd, ft), _tree_leaves(new, Xd, ft)
    _raise_on_categorical_fault(ft)
    r_old = int(np.unique(lo.cpu().numpy()).shape[0])
    r_new = int(np.unique(ln.cpu().numpy()).shape[0])
    if r_old + r_new <= MAX_RANK:
        U = torch.empty((Xd.shape[0], r_old + r_new), dtype=torch.float64, device=Xd.device)
        leaf_vectors_device(lo, s, out=U, col0=0)
        leaf_vectors_device(ln, s, out=U, col0=r_old)
        return U, None, r_old, r_new
    return leaf_vectors_device(lo, s), leaf_vectors_device(ln, s), r_old, r_new


def _copy2d(dst, col0: int, src):
    """dst[:, col0 : col0 + src.shape[1]] = src for row-major float64 device matrices (strided DMA copy)."""
    _lib.check(_lib.lib().bark_copy2d_hip(ctypes.c_void_p(dst.data_ptr() + 8 * col0), int(dst.stride(0)), _lib.ptr(src),
                                          int(src.stride(0)), int(src.shape[0]), int(src.shape[1]), _lib.stream_ptr()))


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
        self.quad = _quadform(self.K_inv, self.y)
        self._pending = None
        self._ws = {}  # rank -> workspace tensor, reused across proposals
        self._X_seen = None  # (caller's X object, device tensor) of the last propose_tree
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

    def _eval_explicit(self, U, r_old: int, r_new: int) -> float:
        lib = _lib.lib()
        r = r_old + r_new
        ws, scalars = self._workspace(r), self._scalars
        _lib.check(lib.bark_lowrank_swap_eval_hip(_lib.ptr(self.K_inv), self.N, _lib.ptr(U), r_old, r_new,
                                                  _lib.ptr(self.y), _lib.ptr(scalars), _lib.ptr(ws), ws.numel(),
                                                  _lib.stream_ptr()))
        return self._finish_eval(ws, r)

    def _finish_eval(self, ws, r: int) -> float:
        dquad, dlogdet = (float(v) for v in self._scalars.cpu().numpy())
        if not (np.isfinite(dquad) and np.isfinite(dlogdet)):
            _raise_if_singular(ws, self.N, r)
        self._pending = ("swap", ws, r, self.quad - dquad, self.logdet + dlogdet)
        return 0.5 * (-(self.quad - dquad) - (self.logdet + dlogdet))

    def propose(self, cur_leaf_vectors, new_leaf_vectors) -> float:
        """MLL the chain would have after swapping the old tree's (scaled) leaf vectors for the new tree's
        (bark_sampler.py:238-256).  Does not modify the state; call `accept()` to commit."""
        import torch

        U_old, U_new = _dev64(cur_leaf_vectors), _dev64(new_leaf_vectors)
        if U_old.shape[0] != self.N or U_new.shape[0] != self.N:
            raise ValueError("leaf vectors must have N rows")
        r_old, r_new = U_old.shape[1], U_new.shape[1]
        if max(r_old, r_new) > MAX_RANK:
            raise ValueError(f"a tree contributes at most {MAX_RANK} leaf vectors (got {r_old} and {r_new})")
        if r_old + r_new > MAX_RANK:
            return self._propose_sequential(U_old, U_new)
        U = torch.empty((self.N, r_old + r_new), dtype=torch.float64, device=self.K_inv.device)
        _copy2d(U, 0, U_old)
        _copy2d(U, r_old, U_new)
        return self._eval_explicit(U, r_old, r_new)

    def _propose_sequential(self, U_old, U_new) -> float:
        """More than 64 leaf vectors in total (two bushy trees: the default container allows 50 leaves each): the
        reference's own chain, subtract then add (bark_sampler.py:242-255), on a copy of K_inv."""
        import torch

        lib = _lib.lib()
        K_new = torch.empty_like(self.K_inv)
        dets = torch.empty(2, dtype=torch.float64, device=self.K_inv.device)
        src = self.K_inv
        for i, (U, subtract) in enumerate(((U_old, 1), (U_new, 0))):
            r = int(U.shape[1])
            ws = self._workspace(r)
            _lib.check(lib.bark_lowrank_update_hip(_lib.ptr(src), self.N, _lib.ptr(U), r, subtract, 1, _lib.ptr(K_new),
                                                   ctypes.c_void_p(dets.data_ptr() + 8 * i), _lib.ptr(ws), ws.numel(),
                                                   _lib.stream_ptr()))
            _raise_if_singular(ws, self.N, r)
            src = K_new
        d1, d2 = (float(v) for v in dets.cpu().numpy())
        quad = _quadform(K_new, self.y)
        logdet = self.logdet + d1 + d2
        self._pending = ("replace", K_new, quad, logdet)
        return 0.5 * (-quad - logdet)

    def accept(self) -> None:
        """Commit the last proposal: K_inv <- K_inv - Y (C+G)^-1 Y' (bark_sampler.py:259-264)."""
        import torch

        if self._pending is None:
            raise RuntimeError("accept() without a pending propose()")
        kind = self._pending[0]
        if kind == "noise_scale":
            _, nodes, noise, scale, X, feat_types = self._pending
            K_inv, K_inv_y, logdet = batched_kernel_inverse(nodes[None], [noise], [scale], X, self.y, feat_types,
                                                            no_null=False, return_device=True, method="leafspace")
            self.K_inv = K_inv[0]
            q = torch.empty(1, dtype=torch.float64, device=self.K_inv.device)
            _lib.check(_lib.lib().bark_rowdot_hip(_lib.ptr(K_inv_y), 1, self.N, self.N, _lib.ptr(self.y), 1.0, None, 0.0,
                                                  _lib.ptr(q), _lib.stream_ptr()))
            self.quad = float(q.item())
            self.logdet = float(logdet[0].item())
        elif kind == "replace":
            _, self.K_inv, self.quad, self.logdet = self._pending
        else:
            _, ws, r, quad, logdet = self._pending
            _lib.check(_lib.lib().bark_lowrank_swap_apply_hip(_lib.ptr(self.K_inv), self.N, r, _lib.ptr(ws),
                                                              _lib.ptr(self.K_inv), _lib.stream_ptr()))
            self.quad, self.logdet = quad, logdet
        self._pending = None

    def _points_of(self, X, ft):
        if self._X_seen is None or self._X_seen[0] is not X:
            Xd, _ = _points(X, ft.shape[0])
            self._X_seen = (X, Xd)
        Xd = self._X_seen[1]
        if Xd.shape[0] != self.N:
            raise ValueError(f"X has {Xd.shape[0]} rows, the chain has {self.N} points")
        return Xd

    def propose_tree(self, old_nodes, new_nodes, X, feat_types, scale: float, m: int) -> float:
        """bark_sampler.py:233-256 from the two trees themselves: both are walked on the GPU and their one-hot
        leaf code, scaled by s_sqrtm = sqrt(scale / m), is the [U_old U_new] of `propose` (one column per leaf;
        leaves no point reaches give zero columns, which change nothing).  Pairs with more than 64 leaves in total
        go through the reached-leaf vectors (and, if those still exceed 64, the subtract-then-add chain of the
        reference).  `accept()` commits as usual."""
        lib = _lib.lib()
        ft = _feat_types(feat_types)
        Xd = self._points_of(X, ft)
        old, new = _as_nodes(old_nodes, 1), _as_nodes(new_nodes, 1)
        if old.ndim != 1 or new.ndim != 1 or old.shape != new.shape:
            raise ValueError(f"trees must be (node_limit,) records of one container, got {old.shape} and {new.shape}")
        s = float(np.sqrt(scale / m))
        info_old = _lib.PackInfo()
        _lib.check(lib.bark_forest_pack_info(_lib.ptr(old), 1, 1, old.shape[0], _lib.ptr(ft), ft.shape[0],
                                             ctypes.byref(info_old)))
        pf = packed_forest(np.stack([old, new])[None], ft)
        r, r_old = int(pf.info.max_bits), int(info_old.max_bits)
        if r > MAX_RANK:
            U, U_new, r_old, r_new = _reached_leaf_vectors(old, new, Xd, ft, s)
            if U_new is None:
                return self._eval_explicit(U, r_old, r_new)
            return self._propose_sequential(U, U_new)
        ws, scalars = self._workspace(r), self._scalars
        _lib.check(lib.bark_tree_swap_eval_hip(_lib.ctx(), _lib.ptr(self.K_inv), self.N, _lib.ptr(pf.packed), pf.info_ref,
                                               _lib.ptr(Xd), Xd.shape[1], r_old, s, _lib.ptr(self.y),
                                               _lib.ptr(scalars), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()))
        val = self._finish_eval(ws, r)
        _raise_on_categorical_fault(ft)
        return val


class ChainBatch:
    """`ChainState` for several independent chains of the sampler (bark_sampler.py:147): K_inv (nc, N, N) resident,
    the per-tree proposals of all chains evaluated by ONE library call — the chain index is a grid dimension of
    every kernel, so the call costs one launch sequence.  `sweep_trees` runs a whole sweep over the trees with the
    Metropolis decision on the device: one read-back per sweep instead of one per tree.  The chains share X and y
    (as in the reference)."""

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
        self.quad = np.array([_quadform(self.K_inv[b], self.y) for b in range(self.nc)])
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

    def _points_of(self, X, ft):
        if self._X_seen is None or self._X_seen[0] is not X:
            Xd, _ = _points(X, ft.shape[0])
            self._X_seen = (X, Xd)
        Xd = self._X_seen[1]
        if Xd.shape[0] != self.N:
            raise ValueError(f"X has {Xd.shape[0]} rows, the chains have {self.N} points")
        return Xd

    def _workspace(self, r: int, extra: int = 0):
        import torch

        key = (r, extra)
        ws = self._ws.get(key)
        if ws is None:
            nbytes = int(_lib.lib().bark_tree_swap_chains_workspace_bytes(self.N, r, self.nc, None)) + extra
            ws = self._ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=self.K_inv.device)
        return ws

    def _old_leaf_counts(self, old, ft):
        lib = _lib.lib()
        r_old = np.empty(old.shape[0], dtype=np.int64)
        info_one = _lib.PackInfo()
        for b in range(old.shape[0]):  # leaves of each old tree alone = the split between removed and added columns
            _lib.check(lib.bark_forest_pack_info(_lib.ptr(np.ascontiguousarray(old[b])), 1, 1, old.shape[1], _lib.ptr(ft),
                                                 ft.shape[0], ctypes.byref(info_one)))
            r_old[b] = info_one.max_bits
        return r_old

    def propose_trees(self, old_trees, new_trees, X, feat_types, scale, m: int) -> np.ndarray:
        """bark_sampler.py:233-256 for one tree per chain: old_trees / new_trees (chains, node_limit) records,
        scale (chains,) -> the (chains,) MLL values the chains would have.  `accept(mask)` commits."""
        lib = _lib.lib()
        ft = _feat_types(feat_types)
        Xd = self._points_of(X, ft)
        old, new = _as_nodes(old_trees, 2), _as_nodes(new_trees, 2)
        if old.shape != new.shape or old.ndim != 2 or old.shape[0] != self.nc:
            raise ValueError(f"trees must be (chains, node_limit) records, got {old.shape} and {new.shape}")
        scale = np.broadcast_to(np.asarray(scale, dtype=np.float64).reshape(-1), (self.nc,))
        r_old = self._old_leaf_counts(old, ft)
        pf = packed_forest(np.stack([old, new], axis=1), ft)  # (chains, 2, L): one [old, new] pair per chain
        r = int(pf.info.max_bits)
        if r > MAX_RANK:
            return self._propose_trees_one_by_one(old, new, Xd, ft, scale, m)
        ws = self._workspace(r)
        s_sqrtm = np.ascontiguousarray(np.sqrt(scale / m))
        _lib.check(lib.bark_tree_swap_eval_chains_hip(_lib.ctx(), _lib.ptr(self.K_inv), self.N, self.nc, _lib.ptr(pf.packed),
                                                      pf.info_ref, _lib.ptr(Xd), Xd.shape[1], _lib.ptr(r_old),
                                                      _lib.ptr(s_sqrtm), _lib.ptr(self.y), _lib.ptr(self._scalars), _lib.ptr(ws),
                                                      ws.numel(), _lib.stream_ptr()))
        sc = self._scalars.cpu().numpy()
        _raise_on_categorical_fault(ft)
        if not np.isfinite(sc).all():
            raise np.linalg.LinAlgError("Singular matrix in a tree-swap update")
        quad, logdet = self.quad - sc[:, 0], self.logdet + sc[:, 1]
        self._pending = ("swap", ws, r, quad, logdet)
        return 0.5 * (-quad - logdet)

    def _propose_trees_one_by_one(self, old, new, Xd, ft, scale, m):
        """A pair with more than 64 leaves in total somewhere: every chain goes through `ChainState`'s fallback
        (reached-leaf vectors, then the reference's subtract-then-add chain) on its own slice of K_inv."""
        states, vals = [], np.empty(self.nc)
        for b in range(self.nc):
            st = ChainState.__new__(ChainState)
            st.K_inv, st.y, st.N = self.K_inv[b], self.y, self.N
            st.quad, st.logdet = float(self.quad[b]), float(self.logdet[b])
            st._pending, st._ws, st._X_seen, st._scalars = None, {}, (Xd, Xd), self._scalars[b]
            vals[b] = st.propose_tree(old[b], new[b], Xd, ft, float(scale[b]), m)
            states.append(st)
        self._pending = ("states", states)
        return vals

    def accept(self, mask) -> None:
        """Commit the pending proposals of the chains where `mask` is true (bark_sampler.py:259-264)."""
        if self._pending is None:
            raise RuntimeError("accept() without a pending propose_trees()")
        mask = np.ascontiguousarray(np.broadcast_to(np.asarray(mask).reshape(-1), (self.nc,)), dtype=np.int32)
        if self._pending[0] == "states":
            for b, st in enumerate(self._pending[1]):
                if mask[b]:
                    st.accept()
                    if st.K_inv.data_ptr() != self.K_inv[b].data_ptr():  # the sequential chain built a new matrix
                        self.K_inv[b].copy_(st.K_inv)
                    self.quad[b], self.logdet[b] = st.quad, st.logdet
            self._pending = None
            return
        _, ws, r, quad, logdet = self._pending
        _lib.check(_lib.lib().bark_lowrank_swap_apply_chains_hip(_lib.ptr(self.K_inv), self.N, self.nc, r, _lib.ptr(mask),
                                                                 _lib.ptr(ws), ws.numel(), _lib.stream_ptr()))
        keep = mask != 0
        self.quad = np.where(keep, quad, self.quad)
        self.logdet = np.where(keep, logdet, self.logdet)
        self._pending = None

    def sweep_trees(self, old_trees, new_trees, log_q_prior, log_u, X, feat_types, scale, m: int) -> np.ndarray:
        """One sweep of the per-tree loop of `_step_bark_sampler` (bark_sampler.py:233-264) for every chain, decided on
        the device: old_trees / new_trees (chains, steps, node_limit) — the tree of step t and its proposal, which the
        host can draw up front because a proposal only depends on the tree it replaces (tree_proposals.py) —,
        log_q_prior and log_u (chains, steps): the proposal ratio and log of the uniform draw of bark_sampler.py:258.
        Per step the device evaluates all chains, accepts where log_u <= min(log_q_prior + new_mll - cur_mll, 0) and
        rewrites those chains' K_inv.  Returns the (chains, steps) boolean accept mask — the caller copies the accepted
        trees into its forest, as bark_sampler.py:264 does — after ONE read-back for the whole sweep."""
        import torch

        lib = _lib.lib()
        ft = _feat_types(feat_types)
        Xd = self._points_of(X, ft)
        old, new = _as_nodes(old_trees, 3), _as_nodes(new_trees, 3)
        if old.shape != new.shape or old.ndim != 3 or old.shape[0] != self.nc:
            raise ValueError(f"trees must be (chains, steps, node_limit) records, got {old.shape} and {new.shape}")
        steps = old.shape[1]
        lq = np.ascontiguousarray(np.asarray(log_q_prior, dtype=np.float64).reshape(self.nc, steps).T)  # (steps, chains)
        lu = np.ascontiguousarray(np.asarray(log_u, dtype=np.float64).reshape(self.nc, steps).T)
        scale = np.broadcast_to(np.asarray(scale, dtype=np.float64).reshape(-1), (self.nc,))
        infos = (_lib.PackInfo * steps)()
        r_old = np.empty((steps, self.nc), dtype=np.int64)
        sizes, pairs = [], []
        for t in range(steps):
            pair = np.ascontiguousarray(np.stack([old[:, t], new[:, t]], axis=1))  # (chains, 2, L)
            pairs.append(pair)
            _lib.check(lib.bark_forest_pack_info(_lib.ptr(pair), self.nc, 2, pair.shape[2], _lib.ptr(ft), ft.shape[0],
                                                 ctypes.byref(infos[t])))
            sizes.append(int(infos[t].packed_bytes))
            r_old[t] = self._old_leaf_counts(old[:, t], ft)
        r_max = max(int(infos[t].max_bits) for t in range(steps))
        if r_max > MAX_RANK:
            raise ValueError(f"sweep_trees supports at most {MAX_RANK} leaves per [old, new] pair (got {r_max}); "
                             "use propose_trees / accept for this sweep")
        offsets = np.zeros(steps, dtype=np.int64)
        offsets[1:] = np.cumsum([(sz + 255) // 256 * 256 for sz in sizes[:-1]])
        host = torch.empty(int(offsets[-1]) + sizes[-1], dtype=torch.uint8)
        for t in range(steps):
            _lib.check(lib.bark_forest_pack(_lib.ptr(pairs[t]), _lib.ptr(ft), ft.shape[0], ctypes.byref(infos[t]),
                                            ctypes.c_void_p(host.data_ptr() + int(offsets[t]))))
        packed = host.to(self.K_inv.device)
        lq_d, lu_d = _lib.to_device(lq), _lib.to_device(lu)
        state = _lib.to_device(np.ascontiguousarray(np.stack([self.quad, self.logdet], axis=1)))
        accept = torch.empty((steps, self.nc), dtype=torch.int32, device=self.K_inv.device)
        ws = self._workspace(r_max, extra=16 * self.nc)
        s_sqrtm = np.ascontiguousarray(np.sqrt(scale / m))
        _lib.check(lib.bark_tree_sweep_chains_hip(_lib.ctx(), _lib.ptr(self.K_inv), self.N, self.nc, steps, _lib.ptr(packed),
                                                  _lib.ptr(offsets), ctypes.cast(infos, ctypes.c_void_p), _lib.ptr(Xd), Xd.shape[1],
                                                  _lib.ptr(r_old), _lib.ptr(s_sqrtm), _lib.ptr(self.y), _lib.ptr(lq_d),
                                                  _lib.ptr(lu_d), _lib.ptr(state), _lib.ptr(accept), _lib.ptr(ws), ws.numel(),
                                                  _lib.stream_ptr()))
        acc = accept.cpu().numpy()  # the one synchronisation of the sweep
        st = state.cpu().numpy()
        # the device has already rewritten K_inv for every accepted step: take the running quad / logdet that belong to
        # it BEFORE raising, so a caller that catches the error keeps a consistent batch (a singular chain is latched
        # at -1 by decide_kernel and stays at the state after its last accepted step)
        self.quad, self.logdet = st[:, 0].copy(), st[:, 1].copy()
        self._pending = None
        self.last_accept = acc.T.copy()
        _raise_on_categorical_fault(ft)
        if (acc < 0).any():
            raise np.linalg.LinAlgError("Singular matrix in a tree-swap update")
        return (acc.T > 0)
