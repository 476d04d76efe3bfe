"""ctypes binding of libbarkhip.so (include/bark_hip.h) + device plumbing.

torch is used here only as the device-memory container (allocation, H2D/D2H copies, the
current HIP stream); every computation goes through the C ABI.  There is no CPU fallback:
if the library is missing or no GPU is visible the product path raises.
"""

from __future__ import annotations

import ctypes
import os
import threading
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# $BARK_LIB_PATH points the binding at another build of the same ABI (A/B of tuning variants: tools/ab/) without
# ever overwriting the in-tree product library.
LIB_PATH = os.environ.get("BARK_LIB_PATH") or os.path.join(_HERE, "csrc", "libbarkhip.so")

BARK_OK = 0
BARK_ERR_ARG, BARK_ERR_TREE, BARK_ERR_CATEGORICAL, BARK_ERR_HIP, BARK_ERR_WORKSPACE = 1, 2, 3, 4, 5
MLL_INCLUDE_SCALE, MLL_INCLUDE_2PI, MLL_RHS_IDENTITY = 1, 2, 4

i64, vp, ci = ctypes.c_int64, ctypes.c_void_p, ctypes.c_int


class PackInfo(ctypes.Structure):
    _fields_ = [
        ("B", i64), ("m", i64), ("L", i64), ("stride", i64),
        ("max_leaves", i64), ("max_depth", i64), ("packed_bytes", i64), ("max_bits", i64),
    ]


class MllTiming(ctypes.Structure):
    _fields_ = [
        ("total_ms", ctypes.c_float), ("gram_ms", ctypes.c_float), ("chol_ms", ctypes.c_float),
        ("diag_ms", ctypes.c_float), ("panel_ms", ctypes.c_float), ("solve_ms", ctypes.c_float),
        ("n_diag_launches", i64), ("n_panel_launches", i64), ("n_solve_launches", i64),
        ("panel_flops", ctypes.c_double), ("solve_flops", ctypes.c_double),
    ]


class MllPlan(ctypes.Structure):
    """bark_mll_plan (include/bark_hip.h): which launch schedule bark_mll_batched_hip takes for a shape."""
    _fields_ = [(n, ctypes.c_int32) for n in (
        "n_chunks", "chunk", "last_chunk", "schedule", "last_schedule", "splitk_layout", "fused_gram", "dev_wait", "dev_gate",
        "pre_update", "lookahead_steps", "splitk_steps", "nrb", "ncb")]


SCHEDULES = ("one_block", "plain", "paired", "pipelined", "splitk", "splitk_lookahead", "two_block", "multi_block")  # BARK_SCHED_*


# every exported symbol of include/*.h: name -> (restype, argtypes)
SIGNATURES = {
    "bark_version": (ci, []),
    "bark_last_error": (ctypes.c_char_p, []),
    "bark_device_wait": (ci, [ci]),
    "bark_xcd_map_selftest": (ci, [ci, ci]),
    "bark_mll_plan_query": (ci, [i64, i64, i64, i64, i64, ci, ci, ctypes.POINTER(MllPlan)]),
    "bark_debug_fail_launch": (ctypes.c_long, [ctypes.c_long]),
    "bark_dev_alloc": (ci, [vp, ctypes.c_size_t, ctypes.POINTER(vp)]),
    "bark_dev_free": (ci, [vp, vp]),
    "bark_ctx_upload": (ci, [vp, vp, vp, ctypes.c_size_t, vp]),
    "bark_ctx_download": (ci, [vp, vp, vp, ctypes.c_size_t, vp]),
    "bark_stream_sync": (ci, [vp, vp]),
    "bark_tree_swap_eval_host_pair": (ci, [vp, vp, i64, vp, i64, vp, i64, vp, ctypes.c_double, vp, vp, vp, vp,
                                           ctypes.c_size_t, vp]),
    "bark_comm_unique_id": (ci, [vp]),
    "bark_comm_create": (ci, [vp, ci, ci, ci, ctypes.POINTER(vp)]),
    "bark_comm_destroy": (None, [vp]),
    "bark_allgather_mll": (ci, [vp, vp, i64, vp, vp]),
    "bark_allreduce_f64": (ci, [vp, vp, i64, ci, vp]),
    "bark_forest_pack_info": (ci, [vp, i64, i64, i64, vp, i64, ctypes.POINTER(PackInfo)]),
    "bark_forest_pack": (ci, [vp, vp, i64, ctypes.POINTER(PackInfo), vp]),
    "bark_ctx_create": (ci, [ci, ctypes.POINTER(vp)]),
    "bark_ctx_destroy": (None, [vp]),
    "bark_ctx_workspace": (ci, [vp, ctypes.c_size_t, ctypes.POINTER(vp)]),
    "bark_ctx_workspace_bytes": (ctypes.c_size_t, [vp]),
    "bark_ctx_status": (ci, [vp, vp, ctypes.POINTER(ctypes.c_int32)]),
    "bark_leaf_indices_hip": (ci, [vp, vp, ctypes.POINTER(PackInfo), vp, i64, i64, vp, vp]),
    "bark_onehot_match_hip": (ci, [vp, i64, i64, vp, i64, ctypes.c_double, vp, i64, vp]),
    "bark_rowdot_hip": (ci, [vp, i64, i64, i64, vp, ctypes.c_double, vp, ctypes.c_double, vp, vp]),
    "bark_mixture_partial_hip": (ci, [vp, vp, i64, i64, vp, vp]),
    "bark_mixture_finish_hip": (ci, [vp, ctypes.c_double, i64, vp, vp, vp]),
    "bark_copy2d_hip": (ci, [vp, i64, vp, i64, i64, i64, vp]),
    "bark_tree_sweep_chains_hip": (ci, [vp, vp, i64, i64, i64, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp,
                                        ctypes.c_size_t, vp]),
    "bark_lowrank_status_hip": (ci, [vp, i64, i64, ctypes.POINTER(ctypes.c_int32), vp]),
    "bark_leaf_npad": (i64, [i64]),
    "bark_leaf_encoding": (ci, [ctypes.POINTER(PackInfo)]),
    "bark_leaf_words": (i64, [ctypes.POINTER(PackInfo)]),
    "bark_leaf_codes_hip": (ci, [vp, vp, ctypes.POINTER(PackInfo), vp, i64, i64, vp, vp]),
    "bark_gram_from_leaves_hip": (ci, [vp, i64, vp, i64, ctypes.POINTER(PackInfo), vp, vp, vp, vp, i64, i64, vp]),
    "bark_mll_workspace_bytes": (ctypes.c_size_t, [i64, i64, i64, i64]),
    "bark_mll_batched_hip": (ci, [vp, vp, ctypes.POINTER(PackInfo), vp, i64, i64, vp, vp, vp, vp, ci, vp, i64, vp, vp, vp,
                                  vp, vp, vp, ctypes.c_size_t, i64, ctypes.POINTER(MllTiming), vp]),
    "bark_quadform_hip": (ci, [vp, vp, i64, vp, vp]),
    "bark_mll_leafspace_workspace_bytes": (ctypes.c_size_t, [i64, i64, i64, i64, i64]),
    "bark_mll_leafspace_hip": (ci, [vp, vp, ctypes.POINTER(PackInfo), vp, i64, i64, vp, vp, vp, ci, vp, i64, vp, vp, vp, vp,
                                    vp, ctypes.c_size_t, i64, vp]),
    "bark_kernel_inverse_leafspace_workspace_bytes": (ctypes.c_size_t, [i64, i64, i64, i64]),
    "bark_kernel_inverse_leafspace_hip": (ci, [vp, vp, ctypes.POINTER(PackInfo), vp, i64, i64, vp, vp, vp, ci, vp, vp, vp, vp,
                                               vp, ctypes.c_size_t, i64, vp]),
    "bark_lowrank_workspace_bytes": (ctypes.c_size_t, [i64, i64]),
    "bark_lowrank_update_hip": (ci, [vp, i64, vp, i64, ci, ci, vp, vp, vp, ctypes.c_size_t, vp]),
    "bark_lowrank_swap_eval_hip": (ci, [vp, i64, vp, i64, i64, vp, vp, vp, ctypes.c_size_t, vp]),
    "bark_lowrank_swap_apply_hip": (ci, [vp, i64, i64, vp, vp, vp]),
    "bark_tree_swap_workspace_bytes": (ctypes.c_size_t, [i64, i64]),
    "bark_tree_swap_chains_workspace_bytes": (ctypes.c_size_t, [i64, i64, i64, vp]),
    "bark_tree_swap_eval_chains_hip": (ci, [vp, vp, i64, i64, vp, ctypes.POINTER(PackInfo), vp, i64, vp, vp, vp, vp, vp,
                                            ctypes.c_size_t, vp]),
    "bark_lowrank_swap_apply_chains_hip": (ci, [vp, i64, i64, i64, vp, vp, ctypes.c_size_t, vp]),
    "bark_tree_swap_eval_hip": (ci, [vp, vp, i64, vp, ctypes.POINTER(PackInfo), vp, i64, i64, ctypes.c_double, vp, vp, vp,
                                     ctypes.c_size_t, vp]),
}

_lib = None
_lock = threading.Lock()


def lib():
    """Load libbarkhip.so (raises if it has not been built: no silent fallback)."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                # torch bundles its own HIP runtime; it must be in the process before libbarkhip.so
                # resolves libamdhip64, or two runtimes get loaded and the second sees no device.
                import torch  # noqa: F401

                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        f"{LIB_PATH} is missing: build it with `make -C bark_amd/csrc` "
                        "(or __graft_entry__.build()); bark_amd has no CPU fallback")
                handle = ctypes.CDLL(LIB_PATH)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(handle, name)
                    fn.restype, fn.argtypes = res, args
                _lib = handle
    return _lib


def check(rc: int):
    """Map a C status to the exception type the reference would raise."""
    if rc == BARK_OK:
        return
    msg = lib().bark_last_error().decode(errors="replace")
    if rc in (BARK_ERR_ARG, BARK_ERR_TREE, BARK_ERR_CATEGORICAL):
        raise ValueError(f"bark_hip: {msg}")
    if rc == BARK_ERR_WORKSPACE:
        raise MemoryError(f"bark_hip: {msg}")
    raise RuntimeError(f"bark_hip: {msg} (status {rc})")


def torch_device():
    import torch

    if not torch.cuda.is_available():
        raise RuntimeError("bark_amd needs an AMD Instinct GPU (HIP device) — no CPU fallback exists")
    return torch.device("cuda", torch.cuda.current_device())


def stream_ptr():
    import torch

    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device (or host numpy) pointer as c_void_p; None -> NULL."""
    if t is None:
        return ctypes.c_void_p(0)
    if isinstance(t, np.ndarray):
        return ctypes.c_void_p(t.ctypes.data)
    return ctypes.c_void_p(t.data_ptr())


def to_device(a, dtype=None):
    """numpy / torch (any device) -> contiguous torch tensor on the GPU."""
    import torch

    dev = torch_device()
    if isinstance(a, torch.Tensor):
        t = a.to(device=dev, dtype=dtype) if dtype is not None else a.to(dev)
        return t.contiguous()
    arr = np.ascontiguousarray(a)
    t = torch.from_numpy(arr)
    if dtype is not None:
        t = t.to(dtype)
    return t.to(dev, non_blocking=False)


# ---- per-thread, per-device context (include/bark_hip.h bark_ctx): helper streams, events, scratch, fault flag ----
_tls = threading.local()


def _destroy_ctx(handle_value: int, device: int):
    """Finalizer of a _Ctx: drain the device (the context's helper streams may still run) and free everything the
    context owns.  Runs when the owning thread's locals are collected, at release_ctx(), or at interpreter exit."""
    try:
        import torch

        if torch.cuda.is_available():
            torch.cuda.synchronize(device)
        lib().bark_ctx_destroy(vp(handle_value))
    except Exception:  # interpreter teardown: the HIP runtime may already be gone
        pass


class _Ctx:
    """Owner of one bark_ctx handle.  threading.local drops a dead thread's dict, which drops this object, whose
    finalizer destroys the context — a thread that exits without release_ctx() no longer leaks its helper streams,
    events, pinned page and grow-only scratch (tens of GB after a c3-sized call)."""

    __slots__ = ("handle", "device", "_fin", "__weakref__")

    def __init__(self, device: int):
        h = vp()
        check(lib().bark_ctx_create(device, ctypes.byref(h)))
        self.handle, self.device = h, device
        self._fin = weakref.finalize(self, _destroy_ctx, h.value, device)

    def close(self):
        self._fin()


def ctx():
    """The calling thread's bark_ctx for the current device (created on first use).  Host threads never share a
    context, so concurrent callers on their own streams share no mutable library state."""
    dev = torch_device().index
    handles = getattr(_tls, "handles", None)
    if handles is None:
        handles = _tls.handles = {}
    c = handles.get(dev)
    if c is None:
        c = handles[dev] = _Ctx(dev)
    return c.handle


def release_ctx():
    """Destroy the calling thread's contexts (frees their scratch buffers); they are re-created on demand."""
    handles = getattr(_tls, "handles", None) or {}
    for c in list(handles.values()):
        c.close()
    handles.clear()
    cache = getattr(_tls, "packed", None)
    if cache is not None:
        cache.clear()


class Scratch:
    """View of the context's grow-only device scratch.  INVALIDATED by the next `workspace()` call of this thread that
    asks for more bytes (the old buffer is freed after a device synchronisation) and by release_ctx(): fetch it
    immediately before the call that uses it and do not keep it across calls."""

    def __init__(self, ptr_value: int, nbytes: int):
        self._ptr, self._n = ptr_value, nbytes

    def data_ptr(self) -> int:
        return self._ptr

    def numel(self) -> int:
        return self._n


def workspace(nbytes: int) -> Scratch:
    """The calling thread's scratch, grown to at least `nbytes`.  The buffer is a raw hipMalloc outside torch's
    caching allocator, so a failed growth is retried once after handing torch's cached-but-free blocks back to the
    driver; a second failure raises MemoryError."""
    out = vp()
    rc = lib().bark_ctx_workspace(ctx(), int(nbytes), ctypes.byref(out))
    if rc == BARK_ERR_WORKSPACE:
        import torch

        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        rc = lib().bark_ctx_workspace(ctx(), int(nbytes), ctypes.byref(out))
    check(rc)
    assert out.value % 256 == 0
    return Scratch(out.value, int(lib().bark_ctx_workspace_bytes(ctx())))


def workspace_bytes() -> int:
    handles = getattr(_tls, "handles", None) or {}
    return sum(int(lib().bark_ctx_workspace_bytes(c.handle)) for c in handles.values())


def release_workspace():
    release_ctx()


def check_categorical_fault():
    """Raise what the reference raises inside `1 << int(x)` (forest.py:38) if a leaf walk enqueued by this thread met
    a NaN / inf / negative value at a categorical split.  One 4-byte read-back (synchronises the current stream)."""
    flag = ctypes.c_int32(0)
    check(lib().bark_ctx_status(ctx(), stream_ptr(), ctypes.byref(flag)))
    if flag.value:
        raise ValueError("categorical feature value is negative, NaN or inf")
