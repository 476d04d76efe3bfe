"""Batched GP marginal log-likelihood of forest samples on MI355X.

Reference call sites:
  * examples/mcmc/mcmc_record_mll.py:57-74  `mll(model, data, domain)` — no scale, with n*log(2pi)
  * src/bark/fitting/bark_sampler.py:153-162,267-272 — with scale, `quick_inverse.mll` (no 2pi term)

Both are served by one fused device sweep per chunk of forests (leaf walk -> Gram ->
blocked Cholesky -> solves), `bark_mll_batched_hip` in include/bark_hip.h.  The reference uses
LU `inv` + `slogdet`; results agree to rtol 1e-9 / atol 1e-8 (fp64), see DESIGN.md.
"""

from __future__ import annotations

import ctypes
import os

import numpy as np

from .. import _lib
from ..forest import _as_nodes, _feat_types, _is_torch, _points, packed_forest


def _feat_types_of(domain_or_feat_types):
    """The reference passes a bofire `Domain` only to call get_feature_types_array(domain)
    (mcmc_record_mll.py:62, tree_gps.py:96).  Accept that array directly, or any object exposing
    `feat_types`, or a bofire Domain when bofire_mixed is importable."""
    obj = domain_or_feat_types
    if hasattr(obj, "feat_types"):
        return np.asarray(obj.feat_types)
    if isinstance(obj, (np.ndarray, list, tuple)):
        return np.asarray(obj)
    try:  # pragma: no cover - bofire is not installed in the build image
        from bofire_mixed.domain import get_feature_types_array

        return get_feature_types_array(obj)
    except ImportError as exc:
        raise TypeError("pass feat_types (int array: 0=Cat, 1=Int, 2=Cont) or a bofire Domain") from exc


def _fit_chunk(B: int, need, budget_bytes: int | None = None) -> int:
    """Largest chunk of forests whose workspace `need(chunk)` (monotone) fits the HBM budget: default 70 % of the
    free device memory (counting the cached workspace as free), or $BARK_WORKSPACE_GB."""
    import torch

    if budget_bytes is None and not os.environ.get("BARK_WORKSPACE_GB") and need(B) <= (256 << 20):
        return int(B)  # small enough not to ask the driver (a latency-sensitive caller may be in a sampler loop);
        # an explicit budget (argument or $BARK_WORKSPACE_GB) is always honoured
    if budget_bytes is None:
        env = os.environ.get("BARK_WORKSPACE_GB")
        if env:
            budget_bytes = int(float(env) * (1 << 30))
        else:
            free, _total = torch.cuda.mem_get_info()
            budget_bytes = int(0.7 * (free + _lib.workspace_bytes()))  # this thread's scratch counts as free
    if need(B) <= budget_bytes:
        return int(B)
    lo, hi = 1, int(B)
    while lo < hi:
        mid = (lo + hi + 1) // 2
        if need(mid) <= budget_bytes:
            lo = mid
        else:
            hi = mid - 1
    return lo


def choose_chunk(B: int, N: int, C: int, m: int, budget_bytes: int | None = None) -> int:
    """How many forests to factorise concurrently in the dense sweep: as many as fit the HBM budget."""
    lib = _lib.lib()
    return _fit_chunk(B, lambda k: int(lib.bark_mll_workspace_bytes(N, C, m, k)), budget_bytes)


def _run(forest, noise, scale, X, y, feat_types, flags, cand=None, timing=None, chunk=None, shift=None,
         want_cov=False):
    import torch

    lib = _lib.lib()
    ft = _feat_types(feat_types)
    nodes = _as_nodes(forest, 2)
    nodes3 = nodes.reshape(-1, *nodes.shape[-2:])
    B = nodes3.shape[0]
    Xd, _ = _points(X, ft.shape[0])
    N, d = Xd.shape
    yd = _lib.to_device(y.detach() if _is_torch(y) else np.asarray(y, dtype=np.float64))
    yd = yd.to(torch.float64).reshape(-1).contiguous()
    if yd.shape[0] != N:
        raise ValueError(f"y has {yd.shape[0]} rows, X has {N}")
    noise_d = _lib.to_device(np.ascontiguousarray(np.asarray(noise, dtype=np.float64).reshape(-1)))
    if noise_d.shape[0] != B:
        raise ValueError(f"noise has {noise_d.shape[0]} entries for {B} forests")
    scale_d = None
    if scale is not None:
        scale_d = _lib.to_device(np.ascontiguousarray(np.asarray(scale, dtype=np.float64).reshape(-1)))
        if scale_d.shape[0] != B:
            raise ValueError(f"scale has {scale_d.shape[0]} entries for {B} forests")
    shift_d = None
    if shift is not None:
        shift_d = _lib.to_device(np.ascontiguousarray(np.asarray(shift, dtype=np.float64).reshape(-1)))
        if shift_d.shape[0] != B:
            raise ValueError(f"shift has {shift_d.shape[0]} entries for {B} forests")
    C = 0
    cand_d = mu = var = cov = None
    if flags & _lib.MLL_RHS_IDENTITY:
        C = N
    elif cand is not None:
        cand_d, _ = _points(cand, ft.shape[0])
        C = cand_d.shape[0]
    if C:
        mu = torch.empty((B, C), dtype=torch.float64, device=Xd.device)
        var = torch.empty((B, C), dtype=torch.float64, device=Xd.device)
        if want_cov:
            cov = torch.empty((B, C, C), dtype=torch.float64, device=Xd.device)
    pf = packed_forest(nodes3, ft)
    Bc = chunk or choose_chunk(B, N, C, pf.m)
    nbytes = int(lib.bark_mll_workspace_bytes(N, C, pf.m, Bc))
    ws = _lib.workspace(nbytes)
    out = torch.empty(B, dtype=torch.float64, device=Xd.device)
    info = torch.empty(B, dtype=torch.int32, device=Xd.device)
    tref = ctypes.byref(timing) if timing is not None else None
    def call():
        _lib.check(lib.bark_mll_batched_hip(
            _lib.ctx(), _lib.ptr(pf.packed), pf.info_ref, _lib.ptr(Xd), N, d, _lib.ptr(yd), _lib.ptr(noise_d), _lib.ptr(scale_d),
            _lib.ptr(shift_d), flags, _lib.ptr(cand_d), C, _lib.ptr(out), _lib.ptr(mu), _lib.ptr(var), _lib.ptr(cov),
            _lib.ptr(info),
            _lib.ptr(ws), ws.numel(), Bc, tref, _lib.stream_ptr()))

    call()
    bad = info.cpu().numpy()  # the one read-back of the call's status
    if (bad == -3).any():
        # the helper streams did not run beside this one (include/bark_hip.h, bark_device_wait): event joins from now on
        import warnings

        warnings.warn("bark_amd: device-side wait timed out; falling back to event joins for this process", RuntimeWarning)
        # the timed-out call has joined its helper streams into this one (chol.hip, rejoin_helpers), but the retry rewrites
        # the same workspace: nothing of the failed call may still be running anywhere on the device
        torch.cuda.synchronize()
        lib.bark_device_wait(0)
        call()
        bad = info.cpu().numpy()
        if (bad == -3).any():
            torch.cuda.synchronize()  # before the error propagates and the workspace is handed to someone else
    _raise_on_info(bad, "kernel matrix")
    return (out, mu, var, cov) if want_cov else (out, mu, var)


def _raise_on_info(info, what: str):
    """info_out of the sweep entry points: -1 = invalid categorical value met by a leaf walk (ValueError, as the
    reference's `1 << int(x)`), k > 0 = first non-positive pivot (LinAlgError, as np.linalg.inv on a singular matrix)."""
    bad = info if isinstance(info, np.ndarray) else info.cpu().numpy()
    if not bad.any():
        return
    if (bad == -3).any():
        raise RuntimeError("bark_hip: a device-side wait timed out (bark_device_wait / $BARK_NO_DEVICE_WAIT)")
    if (bad < 0).any():
        _clear_fault()  # read-and-reset, so the next call starts clean
        raise ValueError("categorical feature value is negative, NaN or inf")
    b = int(np.flatnonzero(bad)[0])
    raise np.linalg.LinAlgError(f"{what} of forest sample {b} is not positive definite (pivot {int(bad[b])} <= 0)")


def _clear_fault():
    try:
        _lib.check_categorical_fault()
    except ValueError:
        pass


def batched_kernel_inverse(forest, noise, scale, X, y, feat_types, *, no_null: bool = True, return_device=False,
                           chunk: int | None = None, method: str = "dense"):
    """Explicit inverse of the GP kernel matrix of each forest sample, as the acquisition builder needs it
    (src/bark/optimizer/opt_model.py:54-59,101):

        K_s   = scale_b * batched_forest_gram_matrix[_no_null](forest)[b] + (1e-6 + noise_b) I
        K_inv = inv(K_s)            (B, N, N)
        K_inv_y = K_inv @ y         (B, N)
        logdet  = log|K_s|          (B,)

    Computed from the Cholesky factor (K_inv = U^-1 U^-T: the same sweep with an identity right-hand
    side, then one MFMA V'V product) instead of the reference's LU `np.linalg.inv`.

    method="leafspace" (no_null=False only) builds the same three results from the R x R leaf-space system,
    K_inv = (I - c Z M^-1 Z') / sigma2 (include/bark_hip.h) — what a single chain's noise/scale step wants,
    where one N x N factorisation cannot fill the GPU."""
    nodes = _as_nodes(forest, 2)
    nodes3 = nodes.reshape(-1, *nodes.shape[-2:])
    scale = np.asarray(scale, dtype=np.float64).reshape(-1)
    if method == "leafspace":
        if no_null:
            raise ValueError("method='leafspace' supports no_null=False only")
        mll_noconst, K_inv_y, K_inv = _run_leafspace(nodes3, noise, scale, X, y, feat_types, _lib.MLL_INCLUDE_SCALE,
                                                     chunk=chunk, want_inverse=True)
        return _inverse_results(mll_noconst, K_inv_y, K_inv, X, y, return_device)
    if method != "dense":
        raise ValueError(f"unknown method {method!r} (use 'dense' or 'leafspace')")
    shift = None
    if no_null:  # forest.py:102-111 folded into (shift, scale)
        num_trees = nodes3.shape[-2]
        num_null = np.sum(nodes3[:, :, 0]["is_leaf"], axis=-1).astype(np.int64)
        shift = num_null / num_trees
        scale = scale * (num_trees / np.maximum(num_trees - num_null, 1))
    flags = _lib.MLL_INCLUDE_SCALE | _lib.MLL_RHS_IDENTITY
    mll_noconst, K_inv_y, _, K_inv = _run(nodes3, noise, scale, X, y, feat_types, flags, chunk=chunk, shift=shift,
                                          want_cov=True)
    return _inverse_results(mll_noconst, K_inv_y, K_inv, X, y, return_device)


def _inverse_results(mll_noconst, K_inv_y, K_inv, X, y, return_device):
    """mll = 0.5(-y'K^-1 y - logdet)  =>  logdet = -2 mll - y'K^-1 y."""
    import torch

    yd = _lib.to_device(y.detach() if _is_torch(y) else np.asarray(y, dtype=np.float64)).to(torch.float64).reshape(-1)
    yd = yd.contiguous()
    B, N = K_inv_y.shape
    logdet = torch.empty(B, dtype=torch.float64, device=K_inv_y.device)  # -(y' K^-1 y) - 2 mll per forest sample
    _lib.check(_lib.lib().bark_rowdot_hip(_lib.ptr(K_inv_y), B, N, N, _lib.ptr(yd), -1.0, _lib.ptr(mll_noconst.contiguous()),
                                          -2.0, _lib.ptr(logdet), _lib.stream_ptr()))
    if return_device or _is_torch(X):
        return K_inv, K_inv_y, logdet
    return K_inv.cpu().numpy(), K_inv_y.cpu().numpy(), logdet.cpu().numpy()


def _run_leafspace(forest, noise, scale, X, y, feat_types, flags, chunk=None, cand=None, want_inverse=False):
    """Leaf-space evaluation (bark_mll_leafspace_hip): R x R system instead of N x N.
    Returns the (B,) MLL tensor, (mll, mu, var) when candidates are given, or (mll, K_inv_y, K_inv) with
    `want_inverse` (bark_kernel_inverse_leafspace_hip)."""
    import torch

    lib = _lib.lib()
    ft = _feat_types(feat_types)
    nodes = _as_nodes(forest, 2)
    nodes3 = nodes.reshape(-1, *nodes.shape[-2:])
    B = nodes3.shape[0]
    Xd, _ = _points(X, ft.shape[0])
    N, d = Xd.shape
    yd = _lib.to_device(y.detach() if _is_torch(y) else np.asarray(y, dtype=np.float64))
    yd = yd.to(torch.float64).reshape(-1).contiguous()
    if yd.shape[0] != N:
        raise ValueError(f"y has {yd.shape[0]} rows, X has {N}")
    noise_d = _lib.to_device(np.ascontiguousarray(np.asarray(noise, dtype=np.float64).reshape(-1)))
    scale_d = None if scale is None else _lib.to_device(np.ascontiguousarray(np.asarray(scale, dtype=np.float64).reshape(-1)))
    if noise_d.shape[0] != B or (scale_d is not None and scale_d.shape[0] != B):
        raise ValueError(f"noise/scale must have one entry per forest ({B})")
    C = 0
    cand_d = mu = var = None
    if cand is not None:
        cand_d, _ = _points(cand, ft.shape[0])
        C = cand_d.shape[0]
        mu = torch.empty((B, C), dtype=torch.float64, device=Xd.device)
        var = torch.empty((B, C), dtype=torch.float64, device=Xd.device)
    pf = packed_forest(nodes3, ft)
    R = int(pf.info.max_bits)
    out = torch.empty(B, dtype=torch.float64, device=Xd.device)
    info = torch.empty(B, dtype=torch.int32, device=Xd.device)
    K_inv = K_inv_y = None
    if want_inverse:
        if cand is not None:
            raise ValueError("candidates and want_inverse are separate calls")
        K_inv = torch.empty((B, N, N), dtype=torch.float64, device=Xd.device)
        K_inv_y = torch.empty((B, N), dtype=torch.float64, device=Xd.device)
        Bc = int(chunk) if chunk else _fit_chunk(
            min(B, max(1, (1 << 30) // max(1, 8 * N * R))),  # keep the (Bc, N, R) row-sum scratch modest
            lambda k: int(lib.bark_kernel_inverse_leafspace_workspace_bytes(N, R, pf.m, k)))
        ws = _lib.workspace(int(lib.bark_kernel_inverse_leafspace_workspace_bytes(N, R, pf.m, Bc)))
        _lib.check(lib.bark_kernel_inverse_leafspace_hip(_lib.ctx(), _lib.ptr(pf.packed), pf.info_ref, _lib.ptr(Xd), N, d, _lib.ptr(yd),
                                                         _lib.ptr(noise_d), _lib.ptr(scale_d), flags, _lib.ptr(out),
                                                         _lib.ptr(K_inv), _lib.ptr(K_inv_y), _lib.ptr(info), _lib.ptr(ws),
                                                         ws.numel(), Bc, _lib.stream_ptr()))
    else:
        Bc = int(chunk) if chunk else _fit_chunk(B, lambda k: int(lib.bark_mll_leafspace_workspace_bytes(N, R, pf.m, k, C)))
        ws = _lib.workspace(int(lib.bark_mll_leafspace_workspace_bytes(N, R, pf.m, Bc, C)))
        _lib.check(lib.bark_mll_leafspace_hip(_lib.ctx(), _lib.ptr(pf.packed), pf.info_ref, _lib.ptr(Xd), N, d, _lib.ptr(yd),
                                              _lib.ptr(noise_d), _lib.ptr(scale_d), flags, _lib.ptr(cand_d), C,
                                              _lib.ptr(out), _lib.ptr(mu), _lib.ptr(var), _lib.ptr(info),
                                              _lib.ptr(ws), ws.numel(), Bc, _lib.stream_ptr()))
    _raise_on_info(info, "leaf-space system")
    if want_inverse:
        return out, K_inv_y, K_inv
    return out if cand is None else (out, mu, var)


def batched_mll(forest, noise, scale, X, y, feat_types, *, include_scale: bool, include_2pi: bool,
                return_device: bool = False, chunk: int | None = None, method: str = "dense"):
    """MLL of each forest sample -> (B,) float64.

    include_scale=False, include_2pi=True  reproduces examples/mcmc/mcmc_record_mll.py:57-74;
    include_scale=True,  include_2pi=False reproduces bark_sampler.py:153-162 (quick_inverse.mll).

    method="dense" (default): leaf walk -> N x N Gram -> blocked Cholesky, the reference's computation on the GPU.
    method="leafspace": the same value from the R x R system I + c Z'Z over the forest's leaves (exact
    Woodbury / determinant-lemma identity, O(N R^2/64 + R^3)); opt-in, see include/bark_hip.h.
    """
    flags = (_lib.MLL_INCLUDE_SCALE if include_scale else 0) | (_lib.MLL_INCLUDE_2PI if include_2pi else 0)
    if include_scale and scale is None:
        raise ValueError("include_scale=True needs scale")
    if method == "leafspace":
        out = _run_leafspace(forest, noise, scale if include_scale else None, X, y, feat_types, flags, chunk=chunk)
    elif method == "dense":
        out, _, _ = _run(forest, noise, scale if include_scale else None, X, y, feat_types, flags, chunk=chunk)
    else:
        raise ValueError(f"unknown method {method!r} (use 'dense' or 'leafspace')")
    return out if return_device else out.cpu().numpy()


def schedule_plan(N: int, B: int, *, C: int = 0, chunk: int | None = None, m: int = 50, leaf_words: int = 5,
                  timing: bool = False) -> dict:
    """Which launch schedule the dense sweep takes for B forests of m trees on N points (+ C candidates), `chunk` resident at
    a time (default: all) — `bark_mll_plan_query`, the function the entry point itself configures its sweep from (DESIGN.md
    section 4 has the table).  Nothing on the reference's side corresponds to it (`inv` + `slogdet`,
    examples/mcmc/mcmc_record_mll.py:63-73, have one schedule).  leaf_words: `bark_leaf_words` of the forests (5 for 50 prior
    trees).  No GPU needed."""
    import ctypes

    plan = _lib.MllPlan()
    _lib.check(_lib.lib().bark_mll_plan_query(N, C, m, B, chunk or B, leaf_words, int(timing), ctypes.byref(plan)))
    d = {n: int(getattr(plan, n)) for n, _ in _lib.MllPlan._fields_}
    d["schedule"], d["last_schedule"] = _lib.SCHEDULES[d["schedule"]], _lib.SCHEDULES[d["last_schedule"]]
    return d


def mll(model, data, domain):
    """examples/mcmc/mcmc_record_mll.py:57-74 — same signature; `domain` may be the feat_types array."""
    forest, noise, _scale = model
    train_x, y = data
    return batched_mll(forest, np.asarray(noise).reshape(-1), None, train_x, y, _feat_types_of(domain),
                       include_scale=False, include_2pi=True)
