"""Explicit hand-off of a gradient buffer between two autograd nodes (no dynamic attributes on tensors)."""
from __future__ import annotations
# ---- hand-off of the in_proj gradient buffer between two autograd nodes -------------------------------------------------
# The gated LayerNorm's backward (layernorm.py) allocates ONE (..., 2C) buffer for the gradient of the in_proj output and
# writes dz into its z half; SplitXZFn.backward (ss2d_fused.py) completes the x half of the SAME buffer in place.  The
# buffer is handed over explicitly: registered here under the address of its z half, claimed (and removed) by the one
# consumer that presents a gradient living at that address with that geometry.  A gradient that took any other route
# (summed with a second consumer's, cloned by a hook) does not match and takes the copying path; stale entries are dropped
# after a few registrations.
_XZ_GRAD_BUFFERS = {}          # address of the z half -> weak reference to the full buffer
_XZ_GRAD_KEEP = 8


def offer_xz_grad_buffer(full, C: int) -> None:
    """`full`: contiguous (..., 2C) fp32 buffer whose z half [..., C:] is about to be written by the caller.

    Held WEAKLY: the gradient the caller returns is a view of `full` and keeps it alive (``view._base``) exactly as long
    as autograd holds that gradient; a buffer nobody claims (another consumer, a cloning hook) is released with its
    gradient instead of staying resident between steps or inside a graph-private pool (~100 MB each)."""
    import weakref
    for k in [k for k, r in _XZ_GRAD_BUFFERS.items() if r() is None]:
        _XZ_GRAD_BUFFERS.pop(k, None)
    while len(_XZ_GRAD_BUFFERS) >= _XZ_GRAD_KEEP:
        _XZ_GRAD_BUFFERS.pop(next(iter(_XZ_GRAD_BUFFERS)))
    _XZ_GRAD_BUFFERS[full.data_ptr() + 4 * C] = weakref.ref(full)


def claim_xz_grad_buffer(dz, shape2):
    """the registered buffer of shape `shape2` = (..., 2C) whose z half IS `dz` (same address, strides, dtype, device), or None"""
    if dz is None:
        return None
    ref = _XZ_GRAD_BUFFERS.pop(dz.data_ptr(), None)
    full = ref() if ref is not None else None
    if full is None:
        return None
    C = shape2[-1] // 2
    ok = (tuple(full.shape) == tuple(shape2) and full.is_contiguous() and full.dtype == dz.dtype and full.device == dz.device
          and tuple(dz.shape) == (*shape2[:-1], C) and dz.stride() == full[..., C:].stride())
    return full if ok else None
