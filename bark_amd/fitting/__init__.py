from .mll import batched_mll, mll  # noqa: F401
from . import quick_inverse  # noqa: F401
