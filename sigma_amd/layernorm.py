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


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
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
        with torch.cuda.device(x.device):
            _capi.check(lib.sigma_layernorm_fwd(ctypes.byref(p), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
                        "layernorm_fwd")
        if need_bwd:
            ctx.save_for_backward(xc, weight, mean, rstd)
            ctx.has_bias = bias is not None
            ctx.eps = float(eps)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _capi.load()
        xc, weight, mean, rstd = ctx.saved_tensors
        C = xc.shape[-1]
        rows = xc.numel() // C
        dy = dy.contiguous()
        dx = torch.empty_like(xc)
        dgamma = torch.empty_like(weight)
        dbeta = torch.empty_like(weight) if ctx.has_bias else None
        nrows = int(lib.sigma_layernorm_bwd_partial_rows(rows))
        ws = torch.empty(max(nrows, 1) * 2 * C, device=xc.device, dtype=torch.float32)
        p = _capi.LayerNormParams()
        p.rows, p.channels, p.eps = rows, C, ctx.eps
        p.x, p.gamma, p.mean, p.rstd = xc.data_ptr(), weight.data_ptr(), mean.data_ptr(), rstd.data_ptr()
        p.dy, p.dx, p.dgamma, p.workspace = dy.data_ptr(), dx.data_ptr(), dgamma.data_ptr(), ws.data_ptr()
        p.dbeta = dbeta.data_ptr() if dbeta is not None else None
        with torch.cuda.device(xc.device):
            _capi.check(lib.sigma_layernorm_bwd(ctypes.byref(p), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
                        "layernorm_bwd")
        return dx, dgamma, dbeta, None


class LayerNorm(nn.LayerNorm):
    """Drop-in nn.LayerNorm; HIP kernels when the input qualifies."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        C = x.shape[-1] if x.dim() > 0 else 0
        if (x.is_cuda and x.dtype == torch.float32 and self.elementwise_affine and len(self.normalized_shape) == 1
                and self.weight.dtype == torch.float32 and C % 4 == 0 and 0 < C <= 2048 and x.numel() > 0):
            return LayerNormFn.apply(x, self.weight, self.bias, self.eps)
        return F.layer_norm(x, self.normalized_shape, self.weight, self.bias, self.eps)
