"""Drop-in for the reference's compiled extension module ``selective_scan_cuda_core``.

Reference: models/encoders/selective_scan/csrc/selective_scan/selective_scan.cpp
(``fwd`` :165-249, ``bwd`` :251-362, pybind :364-367).  Same positional signatures, same
checks (RuntimeError where the reference TORCH_CHECKs), same outputs:

    fwd(u, delta, A, B, C, D_, delta_bias_, delta_softplus, nrows) -> [out, x]
    bwd(u, delta, A, B, C, D_, delta_bias_, dout, x_, delta_softplus, nrows)
        -> [du, ddelta, dA, dB, dC, dD, ddelta_bias]

The work is done by the hand-written HIP kernels in libsigma_hip.so through the C ABI
of include/sigma_scan.h; this file only validates, allocates outputs exactly like the
reference host code, and forwards raw device pointers + the current HIP stream.
There is no fallback: without the library (or a GPU tensor) these functions raise.
"""
from __future__ import annotations

import ctypes
from typing import List, Optional

import torch

from . import _capi

_DTYPES = {torch.float32: _capi.DTYPE_F32, torch.float16: _capi.DTYPE_F16, torch.bfloat16: _capi.DTYPE_BF16}


_launch_hook = None


def set_launch_hook(hook) -> None:
    """Benchmark instrumentation: ``hook(kind, key, launch)`` must call ``launch()`` and may
    bracket it with HIP events on the current stream.  ``key`` = (B, dim, L, N, G, elem_size).
    ``None`` removes the hook.  Not used by the model code."""
    global _launch_hook
    _launch_hook = hook


def _launch(kind, key, fn):
    return fn() if _launch_hook is None else _launch_hook(kind, key, fn)


def _check(cond: bool, msg: str) -> None:
    if not cond:
        raise RuntimeError(msg)


def _check_common(u, delta, A, B, C, D_, delta_bias_, nrows, u_gshift=0):
    _check(u.dtype in _DTYPES, "selective_scan: input type must be float32, float16 or bfloat16")
    _check(A.dtype == torch.float32, "selective_scan: A must be float32")
    _check(delta.dtype == u.dtype and B.dtype == u.dtype and C.dtype == u.dtype,
           "selective_scan: u, delta, B, C must share one dtype")
    for name, t in (("u", u), ("delta", delta), ("A", A), ("B", B), ("C", C)):
        _check(t.is_cuda, f"selective_scan: {name} must be a GPU tensor")
    _check(u.dim() == 3, "u must have shape (batch_size, dim, seqlen)")
    _check(u.stride(-1) == 1 or u.size(-1) <= 1, "u.stride(-1) must be 1")
    _check(delta.stride(-1) == 1 or delta.size(-1) <= 1, "delta.stride(-1) must be 1")
    batch, dim, seqlen = u.shape
    if u_gshift:
        _check(delta.dim() == 3 and B.dim() == 4 and B.size(1) % (1 << u_gshift) == 0
               and u.size(1) * (1 << u_gshift) == delta.size(1), "u must have dim >> u_gshift rows")
        dim = delta.size(1)
    _check(A.dim() == 2, "A must have shape (dim, dstate)")
    dstate = A.size(1)
    _check(B.dim() == 4, "B must have shape (batch_size, n_groups, dstate, seqlen)")
    n_groups = B.size(1)
    nrows = int(nrows)
    _check(nrows >= 1 and dim % (n_groups * nrows) == 0, "dims should be dividable by n_groups * nrows")
    _check(dstate <= _capi.SIGMA_SCAN_MAX_DSTATE // nrows,
           "selective_scan only supports state dimension <= 256 / nrows")
    _check(tuple(delta.shape) == (batch, dim, seqlen), "delta must have shape (batch_size, dim, seqlen)")
    _check(tuple(A.shape) == (dim, dstate), "A must have shape (dim, dstate)")
    _check(tuple(B.shape) == (batch, n_groups, dstate, seqlen),
           "B must have shape (batch_size, n_groups, dstate, seqlen)")
    _check(B.stride(-1) == 1 or seqlen <= 1, "B.stride(-1) must be 1")
    _check(tuple(C.shape) == (batch, n_groups, dstate, seqlen),
           "C must have shape (batch_size, n_groups, dstate, seqlen)")
    _check(C.stride(-1) == 1 or seqlen <= 1, "C.stride(-1) must be 1")
    for name, t in (("D", D_), ("delta_bias", delta_bias_)):
        if t is not None:
            _check(t.dtype == torch.float32, f"{name} must be float32")
            _check(t.is_cuda, f"{name} must be a GPU tensor")
            _check(tuple(t.shape) == (dim,), f"{name} must have shape (dim)")
            _check(t.stride(-1) == 1 or dim <= 1, f"{name}.stride(-1) must be 1")
    return batch, dim, seqlen, dstate, n_groups


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def quad_backward_ok(u, delta, B, C) -> bool:
    """True when the quad-row backward (csrc/scan_bwd4.hip, ckpt_pitch 160) can take these operands: f32 IO,
    dstate in {4, 8, 16}, rows per group divisible by 4, L % 4 == 0, 16-byte aligned u / delta / B / C with
    strides that are multiples of 4 elements (plan_bwd4 in csrc/capi.hip is the authority and fails loudly)."""
    if any(t.dtype != torch.float32 for t in (u, delta, B, C)):
        return False
    Bv = B if B.dim() == 4 else B.unsqueeze(1)
    Cv = C if C.dim() == 4 else C.unsqueeze(1)
    n_groups, dstate, seqlen = Bv.shape[1], Bv.shape[2], Bv.shape[3]
    if dstate not in (4, 8, 16) or seqlen % 4 != 0 or delta.shape[1] % (4 * n_groups) != 0:
        return False
    for t in (u, delta, Bv, Cv):
        if t.data_ptr() % 16 != 0 or t.stride(-1) != 1 or any(s % 4 != 0 for s in t.stride()[:-1]):
            return False
    return True


def rowlane_ok(u, delta, B, C, dout=None) -> bool:
    """True when the row-lane kernels (csrc/scan_fwdr.hip / scan_bwdr.hip, ckpt_pitch 16) can take these operands: f32 IO,
    dstate in {4, 8, 16}, rows per group divisible by 64, L % 4 == 0, 16-byte aligned u / delta / B / C (/ dout) with
    strides that are multiples of 4 elements (rowlane_legal in csrc/capi.hip is the authority and fails loudly)."""
    if any(t.dtype != torch.float32 for t in (u, delta, B, C)):
        return False
    Bv = B if B.dim() == 4 else B.unsqueeze(1)
    Cv = C if C.dim() == 4 else C.unsqueeze(1)
    n_groups, dstate, seqlen = Bv.shape[1], Bv.shape[2], Bv.shape[3]
    if dstate not in (4, 8, 16) or seqlen % 4 != 0 or seqlen == 0 or delta.shape[1] % (64 * n_groups) != 0:
        return False
    for t in (u, delta, Bv, Cv) + ((dout,) if dout is not None else ()):
        if t.data_ptr() % 16 != 0 or t.stride(-1) != 1 or any(s % 4 != 0 for s in t.stride()[:-1]):
            return False
    if dstate * max(Bv.stride(2), Cv.stride(2)) + seqlen >= (1 << 29):
        return False
    return True


_FINE_PITCHES = (_capi.SIGMA_SCAN_CKPT_PITCH_FINE, _capi.SIGMA_SCAN_CKPT_PITCH_320, _capi.SIGMA_SCAN_CKPT_PITCH_160,
                 _capi.SIGMA_SCAN_CKPT_PITCH_16)


def _ckpt_slots(seqlen: int, pitch: int) -> int:
    """checkpoints per row of a fine-checkpoint x: one per `pitch` positions (include/sigma_scan.h; the row-lane kernels of
    pitch 16 kept two per tile in rounds 4-5)"""
    return max((seqlen + pitch - 1) // pitch, 1)


def _fine_pitch(x, seqlen: int, dstate: int, ckpt_pitch: int) -> int:
    """Pitch of a fine-checkpoint x (B, dim, ceil(L/pitch) * N): the caller's ``ckpt_pitch`` if given, else
    inferred from the slot count -- which is only possible when exactly one pitch of {640, 320, 160} gives
    that count (a single slot fits all three for L <= 160): an ambiguous x must come with its pitch."""
    if ckpt_pitch:
        return int(ckpt_pitch)
    slots = x.size(2) // max(dstate, 1)
    match = [pitch for pitch in _FINE_PITCHES if slots == _ckpt_slots(seqlen, pitch)]
    if len(match) == 1:
        return match[0]
    if not match:
        raise RuntimeError("fine-checkpoint x must be (batch, dim, ceil(L/pitch)*dstate) with pitch 640, 320, 160 or 16")
    raise RuntimeError(f"fine-checkpoint x with {slots} slot(s) fits the pitches {match}: pass ckpt_pitch (the pitch given to fwd_ext)")


def _fill_fwd(fp: _capi.FwdParams, u, delta, A, B, C, D_, delta_bias_, out, x, delta_softplus, sizes,
              rev_mask=0, u_gshift=0, ckpt_pitch=0, param_swap=0):
    batch, dim, seqlen, dstate, n_groups = sizes
    fp.rev_group_mask, fp.u_group_shift = int(rev_mask), int(u_gshift)
    fp.param_group_swap = int(param_swap)
    if x is not None and x.dim() == 3:              # fine checkpoints: (B, dim, ceil(L/pitch) * N)
        fp.ckpt_pitch, fp.x_row_stride = _fine_pitch(x, seqlen, dstate, ckpt_pitch), x.stride(1)
        if fp.ckpt_pitch == _capi.SIGMA_SCAN_CKPT_PITCH_16 and not x.is_contiguous():
            raise RuntimeError("ckpt_pitch 16: x must be contiguous (its layout is private to the row-lane kernels)")
    elif x is None and ckpt_pitch:                  # inference: no checkpoints are written, the pitch still selects the kernel
        fp.ckpt_pitch = int(ckpt_pitch)
    fp.batch, fp.dim, fp.seqlen, fp.dstate, fp.n_groups = batch, dim, seqlen, dstate, n_groups
    fp.n_chunks = (seqlen + _capi.SIGMA_SCAN_CHUNK - 1) // _capi.SIGMA_SCAN_CHUNK
    fp.io_dtype = _DTYPES[u.dtype]
    fp.delta_softplus = 1 if delta_softplus else 0
    fp.u, fp.delta, fp.A, fp.B, fp.C = _ptr(u), _ptr(delta), _ptr(A), _ptr(B), _ptr(C)
    fp.D, fp.delta_bias = _ptr(D_), _ptr(delta_bias_)
    fp.out, fp.x = _ptr(out), _ptr(x)
    fp.u_batch_stride, fp.u_d_stride = u.stride(0), u.stride(1)
    fp.delta_batch_stride, fp.delta_d_stride = delta.stride(0), delta.stride(1)
    fp.A_d_stride, fp.A_dstate_stride = A.stride(0), A.stride(1)
    fp.B_batch_stride, fp.B_group_stride, fp.B_dstate_stride = B.stride(0), B.stride(1), B.stride(2)
    fp.C_batch_stride, fp.C_group_stride, fp.C_dstate_stride = C.stride(0), C.stride(1), C.stride(2)
    if out is not None:
        fp.out_batch_stride, fp.out_d_stride = out.stride(0), out.stride(1)


def fwd(u: torch.Tensor, delta: torch.Tensor, A: torch.Tensor, B: torch.Tensor, C: torch.Tensor,
        D_: Optional[torch.Tensor], delta_bias_: Optional[torch.Tensor], delta_softplus: bool,
        nrows: int) -> List[torch.Tensor]:
    """Selective scan forward (selective_scan.cpp:165-249).  ``nrows`` only takes part in
    the shape checks: the row-to-workgroup mapping is chosen by the library."""
    return fwd_ext(u, delta, A, B, C, D_, delta_bias_, delta_softplus, nrows=nrows)


_ROWLANE_TESTED = set()


def rowlane_selftest(device=None) -> None:
    """Once per process and device, before the first ckpt_pitch-16 launch: ``sigma_scan_rowlane_selftest`` (include/sigma_scan.h)
    runs the row-lane kernels on a small problem and compares forward + seven gradients with a host recurrence -- their
    memory waits are hand-counted, so a toolchain that schedules them differently must fail here, loudly.
    Allocates and synchronises: never inside a stream capture (graphed steps call this before they capture)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key in _ROWLANE_TESTED:
        return
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("sigma_amd: the first row-lane scan of this process on this device was issued inside a stream "
                           "capture; run one step (or selective_scan_cuda_core.rowlane_selftest(device)) eagerly first")
    with torch.cuda.device(dev):
        rc = _capi.load().sigma_scan_rowlane_selftest(ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"sigma_scan_rowlane_selftest failed on {dev} (status {rc}): {_capi.last_error()} -- rebuild "
                           f"libsigma_hip.so with the supported ROCm toolchain")
    _ROWLANE_TESTED.add(key)


def fwd_ext(u, delta, A, B, C, D_, delta_bias_, delta_softplus, nrows: int = 1, rev_mask: int = 0,
            u_gshift: int = 0, need_x: bool = True, fine_ckpt: bool = False, ckpt_pitch: int = 0,
            param_swap: int = 0) -> List[torch.Tensor]:
    """``fwd`` plus the two extensions of include/sigma_scan.h used by the fused SS2D path:
    ``rev_mask`` (bit g: group g scans backwards, by addressing) and ``u_gshift`` (group g reads
    the u rows of group g >> u_gshift; u has dim >> u_gshift rows).  ``fine_ckpt``: x is allocated
    as (B, dim, ceil(L/640) * N) with one state checkpoint per 640 elements (include/sigma_scan.h),
    which spares ``bwd_ext`` its forward sweep and lets it run the second-generation kernel
    (csrc/scan_bwd2.hip); ``ckpt_pitch`` = 640 / 320 selects the pitch explicitly (320: 320-element
    backward tiles), 160 the quad-row kernels (csrc/scan_fwd4.hip / scan_bwd4.hip).  Such an x is only valid for
    ``bwd_ext`` called with the same ``ckpt_pitch``.  ``need_x=False`` (inference): x comes back empty.  ``param_swap`` = 1 (four groups): A, D,
    delta_bias stay in the reference's direction order while the sequence operands use the kernel's group
    order (include/sigma_scan.h, param_group_swap)."""
    lib = _capi.load()
    sizes = _check_common(u, delta, A, B, C, D_, delta_bias_, nrows, u_gshift)
    batch, dim, seqlen, dstate, _ = sizes
    n_chunks = (seqlen + _capi.SIGMA_SCAN_CHUNK - 1) // _capi.SIGMA_SCAN_CHUNK
    out = torch.empty_like(delta)                                   # selective_scan.cpp:226
    if fine_ckpt and not ckpt_pitch:
        ckpt_pitch = _capi.SIGMA_SCAN_CKPT_PITCH_FINE
    _check(ckpt_pitch in (0,) + _FINE_PITCHES, "ckpt_pitch must be 0, 640, 320, 160 or 16")
    if not need_x:                                                  # eval / no_grad: nothing is checkpointed
        x = torch.empty((0,), device=u.device, dtype=torch.float32)
    elif ckpt_pitch:
        x = torch.empty((batch, dim, _ckpt_slots(seqlen, ckpt_pitch) * dstate), device=u.device, dtype=torch.float32)
    else:
        x = torch.empty((batch, dim, n_chunks, dstate * 2), device=u.device, dtype=torch.float32)  # :228
    if batch == 0 or seqlen == 0:
        return [out, x]
    fp = _capi.FwdParams()
    _fill_fwd(fp, u, delta, A, B, C, D_, delta_bias_, out, x if need_x else None, delta_softplus, sizes,
              rev_mask, u_gshift, ckpt_pitch, param_swap)
    workspace = None
    if ckpt_pitch == _capi.SIGMA_SCAN_CKPT_PITCH_16:                # few rows: forward summaries of the sequence segments
        rowlane_selftest(u.device)
        ws_bytes = int(lib.sigma_scan_fwd_workspace_bytes(ctypes.byref(fp)))
        _check(ws_bytes >= 0, "selective_scan_fwd: " + _capi.last_error())
        if ws_bytes > 0:
            workspace = torch.empty(ws_bytes, dtype=torch.uint8, device=u.device)
            fp.workspace, fp.workspace_bytes = _ptr(workspace), ws_bytes
    with torch.cuda.device(u.device):                               # CUDAGuard, :240
        stream = torch.cuda.current_stream(u.device).cuda_stream    # :241
        key = (batch, dim, seqlen, dstate, sizes[4], u.element_size())
        _launch("fwd", key, lambda: _capi.check(
            lib.sigma_selective_scan_fwd(ctypes.byref(fp), ctypes.c_void_p(stream)), "selective_scan_fwd"))
    return [out, x]


def bwd(u: torch.Tensor, delta: torch.Tensor, A: torch.Tensor, B: torch.Tensor, C: torch.Tensor,
        D_: Optional[torch.Tensor], delta_bias_: Optional[torch.Tensor], dout: torch.Tensor,
        x_: Optional[torch.Tensor], delta_softplus: bool, nrows: int) -> List[Optional[torch.Tensor]]:
    """Selective scan backward (selective_scan.cpp:251-362)."""
    return bwd_ext(u, delta, A, B, C, D_, delta_bias_, dout, x_, delta_softplus, nrows=nrows)


def bwd_ext(u, delta, A, B, C, D_, delta_bias_, dout, x_, delta_softplus, nrows: int = 1, rev_mask: int = 0,
            u_gshift: int = 0, dout_gshift: int = 0, dB_out: Optional[torch.Tensor] = None,
            dC_out: Optional[torch.Tensor] = None, ckpt_pitch: int = 0, param_swap: int = 0) -> List[Optional[torch.Tensor]]:
    """``bwd`` with the extensions of ``fwd_ext``.  du has one row per CHANNEL row (batch, dim, L)
    even when u_gshift folds several groups onto one copy of u; ``dout_gshift`` does the same for
    dout.  ``dB_out`` / ``dC_out``: optional fp32 (B, G, N, L) views (stride(-1) == 1) the kernel
    writes dB / dC into directly (e.g. slices of the gradient of the x_proj output)."""
    lib = _capi.load()
    sizes = _check_common(u, delta, A, B, C, D_, delta_bias_, nrows, u_gshift)
    batch, dim, seqlen, dstate, n_groups = sizes
    _check(dout.dtype == u.dtype, "dout must have the dtype of u")
    _check(dout.is_cuda, "dout must be a GPU tensor")
    _check(tuple(dout.shape) == (batch, dim >> dout_gshift, seqlen), "dout must have shape (batch_size, dim, seqlen)")
    _check(dout.stride(-1) == 1 or seqlen <= 1, "dout.stride(-1) must be 1")
    if ckpt_pitch == _capi.SIGMA_SCAN_CKPT_PITCH_16 and batch > 0 and seqlen > 0 and (
            dout.data_ptr() % 16 != 0 or any(s % 4 != 0 for s in dout.stride()[:-1])):
        # The row-lane backward loads 16 bytes per lane: a gradient that autograd hands over as a narrowed / offset view
        # is re-laid here (one copy) for EVERY caller instead of failing in the library (u / delta / B / C were checked
        # by rowlane_ok before the forward picked this pitch; dout is only known now).
        dout = dout.contiguous()
        if dout.data_ptr() % 16 != 0:
            dout = dout.clone()
    n_chunks = (seqlen + _capi.SIGMA_SCAN_CHUNK - 1) // _capi.SIGMA_SCAN_CHUNK
    if n_chunks > 1 or seqlen > _capi.SIGMA_SCAN_CKPT_PITCH:
        _check(x_ is not None, "x is required when seqlen > 2048")   # :320 (here: already above 1280)
    if x_ is not None:
        _check(x_.dtype == torch.float32 and x_.is_cuda and x_.is_contiguous(), "x must be a contiguous float32 GPU tensor")
        if x_.dim() == 3:        # fine checkpoints of fwd_ext(fine_ckpt=True / ckpt_pitch=...)
            pitch = _fine_pitch(x_, seqlen, dstate, ckpt_pitch)
            _check(tuple(x_.shape) == (batch, dim, _ckpt_slots(seqlen, pitch) * dstate), "fine-checkpoint x must be (batch, dim, ceil(L/pitch)*dstate)")
        else:
            _check(tuple(x_.shape) == (batch, dim, n_chunks, 2 * dstate),
                   "x must have shape (batch_size, dim, n_chunks, 2 * dstate)")
    du = torch.empty_like(delta)                                     # :329-337 (== empty_like(u) in the reference)
    ddelta = torch.empty_like(delta)
    # dA, dD, ddelta_bias are accumulated into (atomicAdd per row and tile): ONE zero fill for all three
    zeros = torch.zeros(dim * dstate + 2 * dim, dtype=torch.float32, device=A.device)
    dA = zeros[:dim * dstate].view(dim, dstate)
    # fully written by the library (deterministic two-stage sum), so no zero fill is needed
    if dB_out is not None:
        _check(dB_out.dtype == torch.float32 and tuple(dB_out.shape) == tuple(B.shape) and dB_out.stride(-1) == 1 and
               dC_out is not None and dC_out.dtype == torch.float32 and tuple(dC_out.shape) == tuple(C.shape) and
               dC_out.stride(-1) == 1, "dB_out / dC_out must be fp32 views shaped like B / C")
        dB, dC = dB_out, dC_out
    elif batch > 0 and seqlen > 0:
        dB = torch.empty(B.shape, dtype=torch.float32, device=B.device)
        dC = torch.empty(C.shape, dtype=torch.float32, device=C.device)
    else:
        dB, dC = torch.zeros_like(B, dtype=torch.float32), torch.zeros_like(C, dtype=torch.float32)
    dD = zeros[dim * dstate:dim * dstate + dim] if D_ is not None else None
    ddelta_bias = zeros[dim * dstate + dim:] if delta_bias_ is not None else None
    if batch > 0 and seqlen > 0:
        bp = _capi.BwdParams()
        _fill_fwd(bp.fwd, u, delta, A, B, C, D_, delta_bias_, None, x_, delta_softplus, sizes, rev_mask, u_gshift,
                  ckpt_pitch, param_swap)
        bp.dout_group_shift = int(dout_gshift)
        bp.dout, bp.du, bp.ddelta = _ptr(dout), _ptr(du), _ptr(ddelta)
        bp.dA, bp.dB, bp.dC, bp.dD, bp.ddelta_bias = _ptr(dA), _ptr(dB), _ptr(dC), _ptr(dD), _ptr(ddelta_bias)
        bp.dout_batch_stride, bp.dout_d_stride = dout.stride(0), dout.stride(1)
        bp.du_batch_stride, bp.du_d_stride = du.stride(0), du.stride(1)
        bp.ddelta_batch_stride, bp.ddelta_d_stride = ddelta.stride(0), ddelta.stride(1)
        bp.dA_d_stride, bp.dA_dstate_stride = dA.stride(0), dA.stride(1)
        bp.dB_batch_stride, bp.dB_group_stride, bp.dB_dstate_stride = dB.stride(0), dB.stride(1), dB.stride(2)
        bp.dC_batch_stride, bp.dC_group_stride, bp.dC_dstate_stride = dC.stride(0), dC.stride(1), dC.stride(2)
        ws_bytes = int(lib.sigma_scan_bwd_workspace_bytes(ctypes.byref(bp)))
        _check(ws_bytes >= 0, "selective_scan_bwd: " + _capi.last_error())
        workspace = None
        if ws_bytes > 0:                                 # per-workgroup dB/dC partials (caching allocator)
            workspace = torch.empty(ws_bytes, dtype=torch.uint8, device=u.device)
            bp.workspace, bp.workspace_bytes = _ptr(workspace), ws_bytes
        with torch.cuda.device(u.device):
            stream = torch.cuda.current_stream(u.device).cuda_stream
            key = (batch, dim, seqlen, dstate, n_groups, u.element_size())
            _launch("bwd", key, lambda: _capi.check(
                lib.sigma_selective_scan_bwd(ctypes.byref(bp), ctypes.c_void_p(stream)), "selective_scan_bwd"))
            if (ckpt_pitch == _capi.SIGMA_SCAN_CKPT_PITCH_16 and lib.sigma_scan_get_option(b"rl_chain") == 2
                    and not torch.cuda.is_current_stream_capturing()):
                # The chained walk (on request only) poisons a row block with NaN when a hand-over wait runs out; say WHY
                # here instead of leaving a NaN loss to be explained (costs a device synchronisation, in this mode only).
                n = lib.sigma_scan_get_option(b"rl_chain_timeouts")
                if n != 0:
                    raise RuntimeError(
                        f"selective_scan_bwd: {n if n > 0 else 'an unknown number of'} hand-over wait(s) of the chained "
                        "row-lane walk ran out (the producer workgroup was not resident: another stream's kernel held "
                        "its slot); the gradients of this call hold NaN. Unset the 'rl_chain' option or run the scan "
                        "alone on the device")
    if dB_out is None:
        dB, dC = dB.to(B.dtype), dC.to(C.dtype)                                  # :360
    return [du, ddelta, dA, dB, dC, dD, ddelta_bias]
