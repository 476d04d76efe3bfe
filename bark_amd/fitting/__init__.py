from .mll import batched_kernel_inverse, batched_mll, mll, schedule_plan  # noqa: F401
from . import quick_inverse  # noqa: F401
from .incremental import ChainBatch, ChainState  # noqa: F401,E402
