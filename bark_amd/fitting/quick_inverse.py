"""Drop-in for `bark.fitting.quick_inverse` (reference: src/bark/fitting/quick_inverse.py) on MI355X.

Same three functions, same signatures:

    low_rank_inv_update(K_inv, U, subtract=False)            quick_inverse.py:13-21  (Woodbury)
    low_rank_det_update(K_inv, U, K_logdet, subtract=False)  quick_inverse.py:24-33  (determinant lemma)
    mll(K_inv, K_logdet, y)                                  quick_inverse.py:37-38

numpy in -> numpy out; pass CUDA `torch` tensors to keep K_inv resident in HBM between the many
updates of an MCMC step (bark_sampler.py:233-257) — then K_inv comes back as a device tensor; the scalar results
(log-determinant, MLL) are host floats, as in the reference (the sampler branches on them).
The kernels (csrc/lowrank.hip) are two streaming passes over K_inv around r x r algebra.
"""

from __future__ import annotations

import ctypes

import numpy as np

from .. import _lib
from ..forest import _is_torch


def _dev(a):
    import torch

    t = _lib.to_device(a.detach() if _is_torch(a) else np.asarray(a, dtype=np.float64))
    return t.to(torch.float64).contiguous()


def _update(K_inv, U, subtract, want_inv, want_det, symmetric=False):
    import torch

    lib = _lib.lib()
    Kd, Ud = _dev(K_inv), _dev(U)
    if Ud.ndim != 2 or Kd.ndim != 2 or Kd.shape[0] != Kd.shape[1] or Ud.shape[0] != Kd.shape[0]:
        raise ValueError(f"K_inv must be (N, N) and U (N, r); got {tuple(Kd.shape)} and {tuple(Ud.shape)}")
    N, r = Ud.shape
    if r < 1 or r > 64:
        raise ValueError(f"low-rank update supports 1 <= r <= 64 columns (got {r})")
    ws = torch.empty(int(lib.bark_lowrank_workspace_bytes(N, r)), dtype=torch.uint8, device=Kd.device)
    out = torch.empty_like(Kd) if want_inv else None
    det = torch.empty(1, dtype=torch.float64, device=Kd.device) if want_det else None
    _lib.check(lib.bark_lowrank_update_hip(_lib.ptr(Kd), N, _lib.ptr(Ud), r, 1 if subtract else 0, 1 if symmetric else 0,
                                           _lib.ptr(out), _lib.ptr(det), _lib.ptr(ws), ws.numel(),
                                           _lib.stream_ptr()))
    flag = ctypes.c_int32(0)  # np.linalg.solve / slogdet raise on a singular r x r system (quick_inverse.py:19,31)
    _lib.check(lib.bark_lowrank_status_hip(_lib.ptr(ws), N, r, ctypes.byref(flag), _lib.stream_ptr()))
    if flag.value:
        raise np.linalg.LinAlgError("Singular matrix")
    return out, det


def low_rank_inv_update(K_inv, U, subtract: bool = False, *, assume_symmetric: bool = False):
    """quick_inverse.py:13-21: K_inv - K_inv U (mul I + U' K_inv U)^-1 U' K_inv, mul = -1 if subtract.

    `assume_symmetric=True` (additive keyword) promises K_inv == K_inv' — true for every call of the
    sampler (bark_sampler.py:242-255) — and reuses K_inv U as (U' K_inv)', saving one pass over K_inv."""
    out, _ = _update(K_inv, U, subtract, True, False, symmetric=assume_symmetric)
    return out if _is_torch(K_inv) else out.cpu().numpy()


def low_rank_det_update(K_inv, U, K_logdet, subtract: bool = False):
    """quick_inverse.py:24-33: K_logdet + log|det(I + mul U' K_inv U)|."""
    _, det = _update(K_inv, U, subtract, False, True)
    return float(K_logdet) + float(det.item())  # a host scalar, as in the reference (its caller branches on it)


def mll(K_inv, K_logdet, y) -> float:
    """quick_inverse.py:37-38: 0.5 * (-y' K_inv y - K_logdet)."""
    import torch

    Kd = _dev(K_inv)
    yd = _dev(y).reshape(-1).contiguous()
    N = yd.shape[0]
    if Kd.shape != (N, N):
        raise ValueError(f"K_inv is {tuple(Kd.shape)}, y has {N} rows")
    out = torch.empty(1, dtype=torch.float64, device=Kd.device)
    _lib.check(_lib.lib().bark_quadform_hip(_lib.ptr(Kd), _lib.ptr(yd), N, _lib.ptr(out), _lib.stream_ptr()))
    return 0.5 * (-float(out.item()) - float(K_logdet))
