"""Drop-in for `bark.tree_kernels.tree_model_kernel.TreeAgreementKernel`
(reference: src/bark/tree_kernels/tree_model_kernel.py:8-23).

The reference subclasses `gpytorch.kernels.Kernel`; gpytorch is optional here (absent from the
build image): with it installed the class is a real gpytorch kernel, without it the same
`forward` is available on a plain object.  CUDA tensors stay on the device (no numpy round trip).
"""

from __future__ import annotations

import numpy as np
import torch

from ..forest import forest_gram_matrix

try:  # pragma: no cover - gpytorch is not installed in the build image
    import gpytorch as _gpy

    _Base = _gpy.kernels.Kernel
except ImportError:
    class _Base:  # minimal stand-in so the adaptor stays importable
        def __init__(self, *args, **kwargs):
            pass


class TreeAgreementKernel(_Base):
    is_stationary = False

    def __init__(self, forest: np.ndarray, feat_types: np.ndarray):
        super().__init__()
        self.forest = forest
        self.feat_types = feat_types

    def forward(self, x1: torch.Tensor, x2: torch.Tensor, diag=False, **params):
        if diag:  # tree_model_kernel.py:17-18
            return torch.ones(x1.shape[0])
        if x1.is_cuda:
            a = x1.detach()
            b = a if x2 is x1 else x2.detach()
            return forest_gram_matrix(self.forest, a, b, self.feat_types)
        x1n = x1.detach().numpy()
        x2n = x1n if x2 is x1 else x2.detach().numpy()
        return torch.as_tensor(forest_gram_matrix(self.forest, x1n, x2n, self.feat_types))
