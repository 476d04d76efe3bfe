"""bark_amd — MI355X-native forest-kernel Gram + GP marginal-likelihood / posterior engine.

Drop-in for the hot path of TobyBoyne/bark behind the reference's own module layout:

    bark_amd.forest                         <-> bark.forest
    bark_amd.fitting.quick_inverse          <-> bark.fitting.quick_inverse
    bark_amd.fitting.mll                    <-> examples/mcmc/mcmc_record_mll.py::mll (+ fused batched_mll)
    bark_amd.tree_kernels.tree_gps          <-> bark.tree_kernels.tree_gps (forest_predict, mixture)
    bark_amd.tree_kernels.tree_model_kernel <-> bark.tree_kernels.tree_model_kernel

All compute runs in hand-written HIP kernels (bark_amd/csrc, C ABI in include/bark_hip.h).
"""

__version__ = "0.1.0"

from . import forest  # noqa: F401
