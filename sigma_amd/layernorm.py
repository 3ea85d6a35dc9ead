"""nn.LayerNorm with the HIP forward / backward of sigma_amd/csrc/layernorm.hip.

``LayerNorm`` IS an ``nn.LayerNorm`` (same constructor, parameters, state-dict keys; isinstance
checks of the reference such as utils/init_func.py:45 and vmamba.py:2021 keep working).  The HIP
path serves what the model uses: fp32 CUDA tensors, affine, normalisation over the last dimension
with C % 4 == 0 and C <= 2048; anything else goes to ATen's layer_norm."""
from __future__ import annotations

import ctypes

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _capi
from ._handoff import offer_xz_grad_buffer


def _gate_view(z: torch.Tensor, C: int):
    """(tensor to keep alive, row stride) of a gate whose rows of C floats are evenly spaced"""
    if z.stride(-1) == 1 and z.dim() >= 2:
        zs = z.reshape(-1, C) if z.is_contiguous() else None
        if zs is not None:
            return zs, C
        # the chunk view xz[..., d:] of a contiguous (..., 2d) tensor: rows 2d apart
        st = z.stride(-2)
        ok = all(z.stride(i) == z.stride(i + 1) * z.shape[i + 1] for i in range(z.dim() - 2))
        if ok and st % 4 == 0 and st >= C and z.data_ptr() % 16 == 0:
            return z, st
    zc = z.contiguous()
    return zc, C


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps, gate=None, row_scale=None):
        lib = _capi.load()
        xc = x.contiguous()
        C = xc.shape[-1]
        rows = xc.numel() // C
        y = torch.empty_like(xc)
        need_bwd = any(ctx.needs_input_grad)
        mean = torch.empty(rows, device=x.device, dtype=torch.float32) if need_bwd else None
        rstd = torch.empty(rows, device=x.device, dtype=torch.float32) if need_bwd else None
        p = _capi.LayerNormParams()
        p.rows, p.channels, p.eps = rows, C, float(eps)
        p.x, p.gamma, p.beta, p.y = xc.data_ptr(), weight.data_ptr(), (bias.data_ptr() if bias is not None else None), y.data_ptr()
        if need_bwd:
            p.mean, p.rstd = mean.data_ptr(), rstd.data_ptr()
        zk = None
        if gate is not None:
            zk, zstride = _gate_view(gate, C)
            p.gate, p.gate_row_stride = zk.data_ptr(), zstride
        rsc = None
        if row_scale is not None:                 # one factor per sample (leading dimension), see include/sigma_ops.h
            rsc = row_scale.detach().reshape(-1).float().contiguous()
            if rsc.numel() == 0 or rows % rsc.numel() != 0 or rsc.numel() != xc.shape[0]:
                raise RuntimeError("LayerNorm row_scale: one factor per sample of the leading dimension")
            p.row_scale, p.rows_per_scale = rsc.data_ptr(), rows // rsc.numel()
        with torch.cuda.device(x.device):
            _capi.check(lib.sigma_layernorm_fwd(ctypes.byref(p), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
                        "layernorm_fwd")
        if need_bwd:
            ctx.save_for_backward(xc, weight, mean, rstd, bias if bias is not None else weight.new_empty(0),
                                  zk if zk is not None else weight.new_empty(0), rsc if rsc is not None else weight.new_empty(0))
            ctx.scaled = rsc is not None
            ctx.has_bias = bias is not None
            ctx.gated = gate is not None
            ctx.gate_shape = None if gate is None else tuple(gate.shape)
            ctx.zstride = zstride if gate is not None else 0
            # gate = second half of a contiguous (..., 2C) tensor (SS2D: z of the in_proj output): the backward then
            # writes dz into the z half of ONE (..., 2C) gradient buffer, which SplitXZFn.backward completes in place
            base = gate._base if gate is not None else None
            ctx.gate_half = bool(gate is not None and zk is gate and zstride == 2 * C and base is not None and base.is_contiguous()
                                 and base.shape[-1] == 2 * C and base.numel() == 2 * gate.numel()
                                 and gate.data_ptr() == base.data_ptr() + 4 * C)
            ctx.eps = float(eps)
        return y

    @staticmethod
    def backward(ctx, dy):
        return _ln_backward(ctx, dy, None)


def _ln_backward(ctx, dy, dx_add):
    """sigma_layernorm_bwd for a context saved by LayerNormFn.forward; ``dx_add``: gradient that reached x around the
    LayerNorm, added to dx inside the kernel"""
    lib = _capi.load()
    xc, weight, mean, rstd, bias, zk, rsc = ctx.saved_tensors
    C = xc.shape[-1]
    rows = xc.numel() // C
    dy = dy.contiguous()
    dx = torch.empty_like(xc)
    dgamma = torch.empty_like(weight)
    dbeta = torch.empty_like(weight) if ctx.has_bias else None
    nrows = int(lib.sigma_layernorm_bwd_partial_rows(rows, C))
    ws = torch.empty(max(nrows, 1) * 2 * C, device=xc.device, dtype=torch.float32)
    p = _capi.LayerNormParams()
    p.rows, p.channels, p.eps = rows, C, ctx.eps
    p.x, p.gamma, p.mean, p.rstd = xc.data_ptr(), weight.data_ptr(), mean.data_ptr(), rstd.data_ptr()
    p.dy, p.dx, p.dgamma, p.workspace = dy.data_ptr(), dx.data_ptr(), dgamma.data_ptr(), ws.data_ptr()
    p.dbeta = dbeta.data_ptr() if dbeta is not None else None
    if ctx.scaled:
        p.row_scale, p.rows_per_scale = rsc.data_ptr(), rows // rsc.numel()
    if dx_add is not None:
        if tuple(dx_add.shape) != tuple(xc.shape) or dx_add.dtype != torch.float32:
            raise RuntimeError("layernorm backward: the residual gradient must have the shape of x")
        dx_add = dx_add.contiguous()
        if dx_add.data_ptr() % 16:
            dx_add = dx_add.clone()
        p.dx_add = dx_add.data_ptr()
    dz = None
    if ctx.gated:
        if ctx.gate_half:
            full = torch.empty(*ctx.gate_shape[:-1], 2 * C, device=xc.device, dtype=torch.float32)
            offer_xz_grad_buffer(full, C)     # SplitXZFn.backward completes THIS buffer in place (_handoff.py)
            dz = full[..., C:]
            p.dgate_row_stride = 2 * C
        else:
            dz = torch.empty(ctx.gate_shape, device=xc.device, dtype=torch.float32)
        p.beta = bias.data_ptr() if ctx.has_bias else None
        p.gate, p.gate_row_stride, p.dgate = zk.data_ptr(), ctx.zstride, dz.data_ptr()
    with torch.cuda.device(xc.device):
        _capi.check(lib.sigma_layernorm_bwd(ctypes.byref(p), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
                    "layernorm_bwd")
    return dx, dgamma, dbeta, None, dz, None


class LayerNormResidualFn(torch.autograd.Function):
    """(LayerNorm(x), x): the block input is handed through, so that the gradient of the residual stream
    (x + op(norm(x)), vmamba.py:1716-1722) reaches THIS node together with the LayerNorm's and the two are joined
    inside the backward kernel (dx_add) instead of by an add pass of the autograd engine."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        y = LayerNormFn.forward(ctx, x, weight, bias, eps)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dpass):
        if dy is None:                                      # the normalised branch was not used
            return dpass, None, None, None
        return _ln_backward(ctx, dy, dpass)[:4]


def _aligned16(*tensors) -> bool:
    """the kernels issue 16-byte loads / stores on x, gamma, beta: a contiguous view with an odd storage
    offset, or a parameter inside a flat buffer, must not take the vector path"""
    return all(t is None or (t.data_ptr() % 16 == 0 and t.is_cuda and t.dtype == torch.float32) for t in tensors)


class LayerNorm(nn.LayerNorm):
    """Drop-in nn.LayerNorm; HIP kernels when the input qualifies (fp32 GPU tensors, C % 4 == 0, C <= 2048, 16-byte
    aligned operands), ``F.layer_norm`` otherwise."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        C = x.shape[-1] if x.dim() > 0 else 0
        if (x.is_cuda and x.dtype == torch.float32 and self.elementwise_affine and len(self.normalized_shape) == 1
                and C % 4 == 0 and 0 < C <= 2048 and x.numel() > 0
                and _aligned16(self.weight, self.bias) and (not x.is_contiguous() or x.data_ptr() % 16 == 0)):
            return LayerNormFn.apply(x, self.weight, self.bias, self.eps)
        return F.layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps)

    def forward_with_pass(self, x: torch.Tensor):
        """(LayerNorm(x), x) for a residual block: use the second result as the residual operand and its gradient is
        added to the LayerNorm's inside the backward kernel (one pass less per block)."""
        C = x.shape[-1] if x.dim() > 0 else 0
        if (x.is_cuda and x.dtype == torch.float32 and self.elementwise_affine and len(self.normalized_shape) == 1
                and C % 4 == 0 and 0 < C <= 2048 and x.numel() > 0 and x.requires_grad and torch.is_grad_enabled()
                and _aligned16(self.weight, self.bias) and x.is_contiguous() and x.data_ptr() % 16 == 0):
            return LayerNormResidualFn.apply(x, self.weight, self.bias, self.eps)
        return self.forward(x), x

    def forward_gated(self, x: torch.Tensor, z: torch.Tensor, row_scale=None) -> torch.Tensor:
        """LayerNorm(x) * silu(z) in one pass (SS2D.forward, vmamba.py:1077); z may be the strided
        second half of the in_proj output.  ``row_scale``: optional per-sample factor on the result (the stochastic-depth
        mask of the block, applied here for free instead of in a pass of its own after out_proj)."""
        C = x.shape[-1]
        if (x.is_cuda and x.dtype == torch.float32 and z.dtype == torch.float32 and self.elementwise_affine
                and len(self.normalized_shape) == 1 and C % 4 == 0 and 0 < C <= 2048 and x.numel() > 0
                and tuple(z.shape) == tuple(x.shape) and _aligned16(self.weight, self.bias)
                and (not x.is_contiguous() or x.data_ptr() % 16 == 0)):
            return LayerNormFn.apply(x, self.weight, self.bias, self.eps, z, row_scale)
        y = self.forward(x) * F.silu(z)
        return y if row_scale is None else y * row_scale.reshape(-1, *([1] * (y.dim() - 1)))
