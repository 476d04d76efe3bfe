"""Import harness for the *Python* reference (runs ONLY in the build container).

The reference (`/root/reference/src/bark`) is numpy + numba.  numba is not
installed here, so the reference is imported in its own documented debug mode
(`NUMBA_DISABLE_JIT=1`, hinted at `tests/tree_models/test_forest.py:3` of the
reference): `njit` becomes the identity decorator and `prange` becomes `range`,
i.e. the reference's own Python source runs unmodified as plain numpy.

Nothing from the reference is copied: this file only installs import stubs and
returns the reference's modules.  It is used by `make_golden.py` to produce the
committed `.npz` fixtures; it is never imported by tests, bench or product code
and cannot run on the GPU box (no `/root/reference` there).
"""

import sys
import types

REFERENCE_SRC = "/root/reference/src"


def _identity_decorator(*args, **kwargs):
    # supports both `@njit` and `@njit(parallel=False)` / `@jitclass(spec)`
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]

    def wrap(obj):
        return obj

    return wrap


class _TypePlaceholder:
    def __getattr__(self, name):
        return self

    def __call__(self, *a, **k):
        return self

    def __getitem__(self, item):
        return self


def install():
    """Install stubs and make the reference importable.  Idempotent."""
    # never write bytecode into the read-only reference tree (SURVEY §8c rule 1)
    sys.dont_write_bytecode = True

    if "numba" not in sys.modules:
        numba = types.ModuleType("numba")
        numba.njit = _identity_decorator
        numba.jit = _identity_decorator
        numba.prange = range
        placeholder = _TypePlaceholder()
        for name in (
            "int64", "int32", "uint32", "uint8", "float32", "float64", "boolean", "bool_",
            "types", "typed", "typeof",
        ):
            setattr(numba, name, placeholder)
        def _any_type(name):  # any other type name used in a jitclass spec — but no dunder (inspect asks modules for __file__)
            if name.startswith("__"):
                raise AttributeError(name)
            return placeholder

        numba.__getattr__ = _any_type
        experimental = types.ModuleType("numba.experimental")
        experimental.jitclass = _identity_decorator
        numba.experimental = experimental
        sys.modules["numba"] = numba
        sys.modules["numba.experimental"] = experimental

    if "jaxtyping" not in sys.modules:
        jaxtyping = types.ModuleType("jaxtyping")

        class _Ann:
            def __class_getitem__(cls, item):
                return cls

        jaxtyping.Float = _Ann
        jaxtyping.Int = _Ann
        jaxtyping.Shaped = _Ann
        sys.modules["jaxtyping"] = jaxtyping

    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)


def reference_modules():
    install()
    import numpy as np

    import bark.forest as ref_forest
    import bark.fitting.quick_inverse as ref_qi
    import bark.fitting.tree_proposals as ref_tp
    import bark.fitting.tree_traversal as ref_tt
    import bark.fitting.bark_prior_sampler as ref_prior

    # numpy>=2 refuses `-1 -> uint32` in the reference ctor (forest.py:116, it pins
    # numpy 1.26); give the prior sampler a ctor that stores 0xFFFFFFFF explicitly
    # (SURVEY §8c rule 2).  Same record, same values as numpy-1 wraparound.
    def _empty_forest(m, node_limit=100):
        forest = np.zeros((m, node_limit), dtype=ref_forest.NODE_RECORD_DTYPE)
        forest[:, 0] = (1, 0, 0, 0, 0, np.uint32(0xFFFFFFFF), 0, 1)
        return forest

    ref_prior.create_empty_forest = _empty_forest
    return types.SimpleNamespace(
        forest=ref_forest,
        quick_inverse=ref_qi,
        tree_proposals=ref_tp,
        tree_traversal=ref_tt,
        prior=ref_prior,
        empty_forest=_empty_forest,
    )
