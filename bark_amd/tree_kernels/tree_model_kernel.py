"""GPU-backed tree-agreement kernel for gpytorch-style callers.

Interface parity with `bark.tree_kernels.tree_model_kernel.TreeAgreementKernel`
(reference: src/bark/tree_kernels/tree_model_kernel.py:8-23): constructed from a forest of node
records and the feature-type array; `forward(x1, x2, diag=False)` returns the (N, M) leaf-coincidence
Gram matrix as a torch tensor, or a vector of ones for `diag=True` (a point always shares its own leaf).

Differences from the reference, all additive:
  * the Gram matrix comes from the HIP kernels (`bark_amd.forest.forest_gram_matrix`);
  * CUDA inputs are consumed and returned on the device — no `.numpy()` round trip inside botorch's
    fitting loop, `diag=True` included (float64 ones on x1's device); CPU tensors are accepted too and
    give CPU tensors back (`diag=True`: the reference's `torch.ones(N)`);
  * gpytorch is optional (it is absent from the build image): when importable the class derives from
    `gpytorch.kernels.Kernel`, otherwise from a bare stand-in, with the same `forward`.
"""

from __future__ import annotations

import torch

from ..forest import forest_gram_matrix


def _kernel_base():
    try:  # pragma: no cover - gpytorch is not installed in the build image
        from gpytorch.kernels import Kernel

        return Kernel
    except ImportError:
        class _PlainKernel:
            """Minimal base so the adaptor stays importable and callable without gpytorch."""

            def __init__(self, *args, **kwargs):
                del args, kwargs

            def __call__(self, x1, x2=None, **kwargs):
                return self.forward(x1, x1 if x2 is None else x2, **kwargs)

        return _PlainKernel


class TreeAgreementKernel(_kernel_base()):
    #: the leaf-coincidence kernel depends on absolute positions, not on x1 - x2
    is_stationary = False

    def __init__(self, forest, feat_types):
        super().__init__()
        self.forest, self.feat_types = forest, feat_types

    def _gram(self, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        on_device = a.is_cuda
        if on_device:
            left = a.detach()
            right = left if b is a else b.detach()
            return forest_gram_matrix(self.forest, left, right, self.feat_types)
        left = a.detach().cpu().numpy()
        right = left if b is a else b.detach().cpu().numpy()
        return torch.from_numpy(forest_gram_matrix(self.forest, left, right, self.feat_types))

    def forward(self, x1: torch.Tensor, x2: torch.Tensor, diag: bool = False, **params):
        del params  # accepted for gpytorch's calling convention, unused (as in the reference)
        if diag:  # K(x, x) = 1: every tree puts a point in the same leaf as itself
            if x1.is_cuda:  # device inputs get a device result in the Gram's dtype (no host round trip in a fitting loop)
                return torch.ones(x1.shape[0], dtype=torch.float64, device=x1.device)
            return torch.ones(x1.shape[0])  # CPU inputs: the reference's literal `torch.ones(x1.shape[0])`
        return self._gram(x1, x2)
