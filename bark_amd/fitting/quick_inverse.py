"""Drop-in for `bark.fitting.quick_inverse` (reference: src/bark/fitting/quick_inverse.py).

`mll(K_inv, K_logdet, y)` keeps the reference signature and evaluates the quadratic form on the
GPU (bark_quadform_hip).  The Woodbury / determinant-lemma updates (quick_inverse.py:13-33) belong
to the incremental per-tree MCMC step, which is the next row of the scope table and
is not built yet: calling them raises instead of silently running on the CPU.
"""

from __future__ import annotations

import numpy as np

from .. import _lib


def mll(K_inv, K_logdet, y) -> float:
    """quick_inverse.py:37-38: 0.5 * (-y' K_inv y - K_logdet)."""
    import torch

    Kd = _lib.to_device(K_inv if not isinstance(K_inv, np.ndarray) else np.asarray(K_inv, dtype=np.float64))
    yd = _lib.to_device(y if not isinstance(y, np.ndarray) else np.asarray(y, dtype=np.float64)).reshape(-1)
    Kd, yd = Kd.to(torch.float64).contiguous(), yd.to(torch.float64).contiguous()
    N = yd.shape[0]
    if Kd.shape != (N, N):
        raise ValueError(f"K_inv is {tuple(Kd.shape)}, y has {N} rows")
    out = torch.empty(1, dtype=torch.float64, device=Kd.device)
    _lib.check(_lib.lib().bark_quadform_hip(_lib.ptr(Kd), _lib.ptr(yd), N, _lib.ptr(out), _lib.stream_ptr()))
    return 0.5 * (-float(out.item()) - float(K_logdet))


def low_rank_inv_update(K_inv, U, subtract: bool = False):
    raise NotImplementedError("Woodbury update on device is the next scope row (SURVEY §8f-1); not built yet")


def low_rank_det_update(K_inv, U, K_logdet, subtract: bool = False):
    raise NotImplementedError("determinant-lemma update on device is the next scope row (SURVEY §8f-1); not built yet")
