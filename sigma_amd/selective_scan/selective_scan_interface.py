"""Autograd front-end of the HIP selective scan, mirroring
models/encoders/selective_scan/selective_scan/selective_scan_interface.py:10-131.

``SelectiveScanFn`` / ``selective_scan_fn`` have the reference's signature and
normalisation rules (last-dim contiguity, 3-D B/C promoted to one group, D and
delta_bias promoted to float32, backward always issued with nrows = 1) and call
``sigma_amd.selective_scan_cuda_core`` -- i.e. the hand-written gfx950 kernels.

``selective_scan_ref`` is the textbook recurrence in plain torch ops, kept because the
reference package exports it (its unit test uses it as the oracle).  It is NOT a
fallback: nothing in sigma_amd calls it.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .. import selective_scan_cuda_core as _core
from ..ss2d_fused import ckpt_pitch_for


def _last_contig(t: torch.Tensor) -> torch.Tensor:
    return t if t.stride(-1) == 1 else t.contiguous()


class SelectiveScanFn(torch.autograd.Function):
    """out = selective_scan(u, delta, A, B, C, D, delta_bias, delta_softplus, nrows)."""

    @staticmethod
    def forward(ctx, u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=False, nrows=1):
        u, delta, B, C = _last_contig(u), _last_contig(delta), _last_contig(B), _last_contig(C)
        ctx.squeeze_B = B.dim() == 3
        ctx.squeeze_C = C.dim() == 3
        if ctx.squeeze_B:
            B = B.unsqueeze(1)
        if ctx.squeeze_C:
            C = C.unsqueeze(1)
        ctx.d_dtype = None if D is None else D.dtype
        ctx.bias_dtype = None if delta_bias is None else delta_bias.dtype
        if D is not None:
            D = D.contiguous().float()
        if delta_bias is not None:
            delta_bias = delta_bias.contiguous().float()
        if nrows not in (1, 2, 3, 4):
            raise AssertionError(f"nrows must be 1..4, got {nrows}")
        if u.shape[1] % (B.shape[1] * nrows) != 0:
            raise AssertionError(f"dim {u.shape[1]} not divisible by n_groups*nrows = {B.shape[1] * nrows}")
        # x never leaves this Function, so it is allocated with one checkpoint per backward tile
        # (include/sigma_scan.h, ckpt_pitch): the backward then runs csrc/scan_bwd2.hip.  The module-level
        # ``selective_scan_cuda_core.fwd`` keeps the reference-shaped x (selective_scan.cpp:225-228).
        ctx.pitch = ckpt_pitch_for(u.shape[-1], A.shape[1], delta.shape[0] * delta.shape[1],
                                   _core.quad_backward_ok(u, delta, B, C) and nrows == 1,
                                   _core.rowlane_ok(u, delta, B, C), B.shape[1])
        out, x = _core.fwd_ext(u, delta, A, B, C, D, delta_bias, delta_softplus, nrows=nrows, ckpt_pitch=ctx.pitch)
        ctx.delta_softplus = delta_softplus
        ctx.save_for_backward(u, delta, A, B, C, D, delta_bias, x)
        return out

    @staticmethod
    def backward(ctx, dout, *unused):
        u, delta, A, B, C, D, delta_bias, x = ctx.saved_tensors
        dout = _last_contig(dout)
        du, ddelta, dA, dB, dC, dD, ddelta_bias = _core.bwd_ext(
            u, delta, A, B, C, D, delta_bias, dout, x, ctx.delta_softplus, nrows=1, ckpt_pitch=ctx.pitch)
        if ctx.squeeze_B:
            dB = dB.squeeze(1)
        if ctx.squeeze_C:
            dC = dC.squeeze(1)
        if dD is not None and ctx.d_dtype is not None and dD.dtype != ctx.d_dtype:
            dD = dD.to(ctx.d_dtype)
        if ddelta_bias is not None and ctx.bias_dtype is not None and ddelta_bias.dtype != ctx.bias_dtype:
            ddelta_bias = ddelta_bias.to(ctx.bias_dtype)
        return du, ddelta, dA, dB, dC, dD, ddelta_bias, None, None


def selective_scan_fn(u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=False, nrows=1):
    return SelectiveScanFn.apply(u, delta, A, B, C, D, delta_bias, delta_softplus, nrows)


def selective_scan_ref(u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=False):
    """Plain-torch recurrence with the reference's semantics
    (selective_scan_interface.py:86-131): fp32 math, grouped B/C, output in u.dtype.
    Written step-by-step over L; runs on the device of its inputs."""
    dtype_in = u.dtype
    u32 = u.float()
    dl = delta.float()
    if delta_bias is not None:
        dl = dl + delta_bias.float().unsqueeze(-1)
    if delta_softplus:
        dl = F.softplus(dl)
    bsz, dim, L = u32.shape
    N = A.shape[1]
    Bf = B.float() if B.dim() == 4 else B.float().unsqueeze(1)
    Cf = C.float() if C.dim() == 4 else C.float().unsqueeze(1)
    G = Bf.shape[1]
    rep = dim // G
    Bf = Bf.repeat_interleave(rep, dim=1)               # (b, dim, N, L): row d uses group d // rep
    Cf = Cf.repeat_interleave(rep, dim=1)
    decay = torch.exp(dl.unsqueeze(2) * A.float().view(1, dim, N, 1))      # (b, dim, N, L)
    inject = (dl * u32).unsqueeze(2) * Bf                                  # (b, dim, N, L)
    state = u32.new_zeros(bsz, dim, N)
    ys = []
    for i in range(L):
        state = decay[..., i] * state + inject[..., i]
        ys.append((state * Cf[..., i]).sum(-1))
    y = torch.stack(ys, dim=-1) if L > 0 else u32.new_zeros(bsz, dim, 0)
    if D is not None:
        y = y + u32 * D.float().view(1, dim, 1)
    return y.to(dtype_in)
