"""Drop-in for the numpy part of `bark.tree_kernels.tree_gps`
(reference: src/bark/tree_kernels/tree_gps.py:80-131).

`forest_predict` runs one fused device sweep per chunk of forest samples: Gram(X,X), Gram(X,cand),
blocked Cholesky with the candidate block appended as extra columns (V = U^-T K_Xx), then
mu = V'z and var = scale - colsumsq(V).  The reference's `diag=False` full (B, C, C) covariance is
formed only on request (scale - V'V, one more MFMA product), from the same V.  The gpytorch `LeafGP`/`LeafMOGP` classes of
the reference are out of scope.
"""

from __future__ import annotations

from typing import NamedTuple

import numpy as np

from .. import _lib
from ..fitting.mll import _feat_types_of, _run, _run_leafspace
from ..forest import _is_torch


class BARKModel(NamedTuple):
    """tree_gps.py:14-17 — the (forest, noise, scale) triple `forest_predict` and `mll` take; leading dims
    (chains, samples) of all three are flattened by the callees."""

    forest: np.ndarray
    noise: np.ndarray
    scale: np.ndarray


def forest_predict(model, data, candidates, domain, diag: bool = True, method: str = "dense"):
    """tree_gps.py:80-113 -> (mu (B, C), var (B, C) or (B, C, C) if not diag); `domain` may be feat_types.

    method="dense" (default) follows the reference's computation (Gram, factorisation of the N x N matrix);
    method="leafspace" (diag only, <= 64 trees) evaluates the same posterior in closed form over the forest's
    leaves: mu = c sum_{a in L(x)} w_a, var = (scale/m) sum_{a,b in L(x)} (M^-1)_ab — see include/bark_hip.h."""
    forest, noise, scale = model
    train_x, train_y = data
    forest = np.asarray(forest)
    noise = np.asarray(noise, dtype=np.float64).reshape(-1)   # tree_gps.py:88-90 flatten
    scale = np.asarray(scale, dtype=np.float64).reshape(-1)
    flags = _lib.MLL_INCLUDE_SCALE
    if method == "leafspace":
        if not diag:
            raise ValueError("method='leafspace' provides the diagonal posterior only")
        _, mu, var = _run_leafspace(forest, noise, scale, train_x, train_y, _feat_types_of(domain), flags, cand=candidates)
    elif method != "dense":
        raise ValueError(f"unknown method {method!r} (use 'dense' or 'leafspace')")
    elif diag:
        _, mu, var = _run(forest, noise, scale, train_x, train_y, _feat_types_of(domain), flags, cand=candidates)
    else:  # tree_gps.py:108: full (B, C, C) covariance scale - K_xX K^-1 K_Xx (one extra MFMA V'V product)
        _, mu, _, var = _run(forest, noise, scale, train_x, train_y, _feat_types_of(domain), flags, cand=candidates,
                             want_cov=True)
    if _is_torch(candidates):
        return mu, var
    return mu.cpu().numpy(), var.cpu().numpy()


def mixture_of_gaussians_as_normal(mu, var):
    """tree_gps.py:116-131: moments of the equal-weight mixture over forest samples.
    (B x C elementwise host arithmetic, as in the reference; works on numpy or torch.)"""
    if _is_torch(mu):
        import torch

        # the HIP reduction reads raw contiguous (B, C) float64 pairs; anything else — float32, a full (B, C, C)
        # covariance from diag=False, broadcasting shapes — takes the reference's formula in torch, by dtype and shape
        if (mu.is_cuda and _is_torch(var) and var.is_cuda and mu.dtype == torch.float64 and var.dtype == torch.float64
                and mu.ndim == 2 and var.shape == mu.shape):
            from ..distributed import reduce_mixture

            return reduce_mixture(mu, var, int(mu.shape[0]), group=False)
        mu_y = mu.mean(dim=0)
        var_y = (var + mu**2).mean(dim=0) - mu_y**2
        return mu_y, var_y
    mu_y = np.mean(mu, axis=0)
    var_y = np.mean(var + mu**2, axis=0) - mu_y**2
    return mu_y, var_y
