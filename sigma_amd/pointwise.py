"""ChannelAttention's gate and the segmentation loss on the HIP streams of csrc/pointwise.hip (C ABI: include/sigma_ops.h).

``channel_gate``   y = x * sigmoid(fc(mean(x)) + fc(amax(x)))  for contiguous (B, C, H, W) fp32 activations
                   (ChannelAttention, vmamba.py:1725-1741): one pooling pass, the (2B, C) squeeze/excite MLP in torch,
                   one scaling pass; backward = one plane dot product, the MLP's autograd on (2B, C) tensors, one pass
                   that writes dx.
``cross_entropy``  nn.CrossEntropyLoss(reduction='mean', ignore_index) of models/builder.py:146-166 on the CHANNELS-LAST
                   logits the classifier GEMM produces (MambaDecoder.up_x4): log-sum-exp + loss in one pass, gradient in
                   one pass, no (B, classes, H, W) copy.

GPU tensors only (no fallback); the callers keep the torch formulation for everything these kernels do not take.
"""
from __future__ import annotations

import ctypes

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _capi


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _mlp(pooled, w1, w2, batch):
    """fc of ChannelAttention on the stacked [mean; max] rows: conv1x1 - SiLU - conv1x1 (no biases), summed halves"""
    g = F.linear(F.silu(F.linear(pooled, w1)), w2)
    return g[:batch] + g[batch:]


class ChannelGateFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, w2):
        lib = _capi.load()
        B, C, H, W = x.shape
        planes, hw = B * C, H * W
        w1m, w2m = w1.reshape(w1.shape[0], -1), w2.reshape(w2.shape[0], -1)
        pooled = torch.empty(2 * planes, device=x.device, dtype=torch.float32)       # [mean (B, C); max (B, C)]
        cnt = torch.empty(planes, device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _capi.check(lib.sigma_plane_pool(_p(x), planes, hw, _p(pooled), _p(pooled[planes:]), _p(cnt), _stream()), "plane_pool")
            s = torch.sigmoid(_mlp(pooled.view(2 * B, C), w1m, w2m, B)).contiguous()
            y = torch.empty_like(x)
            _capi.check(lib.sigma_plane_scale(_p(x), _p(s), _p(y), planes, hw, _stream()), "plane_scale")
        ctx.save_for_backward(x, w1, w2, pooled, cnt, s)
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _capi.load()
        x, w1, w2, pooled, cnt, s = ctx.saved_tensors
        B, C, H, W = x.shape
        planes, hw = B * C, H * W
        g = g.float().contiguous()
        with torch.cuda.device(x.device):
            ds = torch.empty(B, C, device=x.device, dtype=torch.float32)
            _capi.check(lib.sigma_plane_dot(_p(g), _p(x), _p(ds), planes, hw, _stream()), "plane_dot")
            # the (2B, C) squeeze/excite MLP and the sigmoid by hand on tiny tensors (no nested autograd: the step is
            # captured into HIP graphs):  h = P W1^T, a = silu(h), g = a W2^T, gate = g[:B] + g[B:], s = sigmoid(gate)
            w1m, w2m = w1.reshape(w1.shape[0], -1), w2.reshape(w2.shape[0], -1)
            P2 = pooled.view(2 * B, C)
            h = F.linear(P2, w1m)
            sh = torch.sigmoid(h)
            a = h * sh
            dgate = ds * s * (1.0 - s)
            dg = torch.cat([dgate, dgate], dim=0)                       # (2B, C)
            dw2 = (dg.t() @ a).view_as(w2)
            dh = (dg @ w2m) * (sh * (1.0 + h * (1.0 - sh)))
            dw1 = (dh.t() @ P2).view_as(w1)
            dpl = dh @ w1m
            dpl = dpl.contiguous()
            dx = torch.empty_like(x)
            p = _capi.GateBwdParams()
            p.planes, p.hw = planes, hw
            p.g, p.x, p.scale, p.dx = g.data_ptr(), x.data_ptr(), s.data_ptr(), dx.data_ptr()
            p.dmean, p.dmax = dpl.data_ptr(), dpl[B:].data_ptr()
            p.max, p.count = pooled[planes:].data_ptr(), cnt.data_ptr()
            _capi.check(lib.sigma_plane_gate_bwd(ctypes.byref(p), _stream()), "plane_gate_bwd")
        return dx, dw1, dw2


def channel_gate_ok(x: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor) -> bool:
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous() and x.numel() > 0
            and w1.dtype == torch.float32 and w2.dtype == torch.float32)


def channel_gate(x: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor) -> torch.Tensor:
    """x (B, C, H, W) contiguous; w1 (C/r, C, 1, 1), w2 (C, C/r, 1, 1): the two bias-free 1x1 convolutions of fc"""
    if not channel_gate_ok(x, w1, w2):
        raise RuntimeError("channel_gate: contiguous fp32 (B, C, H, W) GPU tensors only (no fallback)")
    return ChannelGateFn.apply(x, w1, w2)


class ScaleResidualFn(torch.autograd.Function):
    """a + x * scale with a per-channel scale on channels-last tensors -- the two residuals of the decoder block,
    ``x * scale1 + op(norm1(x))`` and ``x * scale2 + conv_blk(norm2(x))`` (vmamba.py:1800-1805).  Forward: torch's
    addcmul (one pass); backward: ONE pass over dy and x for dx = dy * scale and dscale = sum dy * x
    (sigma_colscale_bwd) instead of two multiplies and a column reduction, and dy itself for the branch."""

    @staticmethod
    def forward(ctx, a, x, scale):
        ctx.save_for_backward(x, scale)
        return torch.addcmul(a, x, scale)

    @staticmethod
    def backward(ctx, dy):
        x, scale = ctx.saved_tensors
        C = x.shape[-1]
        g = dy.contiguous()
        xc = x.contiguous()
        dx = torch.empty_like(xc)
        ds = torch.zeros(C, device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            _capi.check(_capi.load().sigma_colscale_bwd(_p(g), _p(xc), _p(scale), _p(dx), _p(ds), xc.numel() // C, C, _stream()),
                        "colscale_bwd")
        return dy, dx, ds


def scale_residual_ok(a: torch.Tensor, x: torch.Tensor, scale: torch.Tensor) -> bool:
    C = x.shape[-1] if x.dim() else 0
    return (a.is_cuda and x.is_cuda and a.dtype == torch.float32 and x.dtype == torch.float32 and scale.dtype == torch.float32
            and tuple(a.shape) == tuple(x.shape) and tuple(scale.shape) == (C,) and C % 4 == 0 and 0 < C <= 1024
            and x.is_contiguous() and x.data_ptr() % 16 == 0 and scale.data_ptr() % 16 == 0 and x.numel() > 0)


def scale_residual(a: torch.Tensor, x: torch.Tensor, scale: torch.Tensor) -> torch.Tensor:
    """a + x * scale (channel-wise) -- on the fused backward where it applies, torch.addcmul otherwise"""
    if scale_residual_ok(a, x, scale) and torch.is_grad_enabled() and (x.requires_grad or scale.requires_grad or a.requires_grad):
        return ScaleResidualFn.apply(a, x, scale)
    return torch.addcmul(a, x, scale)


class SoftmaxCEFn(torch.autograd.Function):
    """mean over the non-ignored pixels of -log softmax(logits)[label]; logits (rows, classes) contiguous"""

    @staticmethod
    def forward(ctx, logits2, labels, ignore_index):
        lib = _capi.load()
        rows, nc = logits2.shape
        lse = torch.empty(rows, device=logits2.device, dtype=torch.float32)
        partial = torch.empty(_capi.SIGMA_CE_BLOCKS, 2, device=logits2.device, dtype=torch.float32)
        with torch.cuda.device(logits2.device):
            _capi.check(lib.sigma_softmax_ce_fwd(_p(logits2), _p(labels), rows, nc, int(ignore_index), _p(lse), _p(partial), _stream()),
                        "softmax_ce_fwd")
        tot = partial.sum(0)
        ctx.save_for_backward(logits2, labels, lse, tot)
        ctx.ignore_index = int(ignore_index)
        return tot[0] / tot[1]

    @staticmethod
    def backward(ctx, g):
        lib = _capi.load()
        logits2, labels, lse, tot = ctx.saved_tensors
        rows, nc = logits2.shape
        scale = (g.float() / tot[1]).reshape(1).contiguous()
        dl = torch.empty_like(logits2)
        with torch.cuda.device(logits2.device):
            _capi.check(lib.sigma_softmax_ce_bwd(_p(logits2), _p(labels), _p(lse), _p(scale), rows, nc, ctx.ignore_index, _p(dl), _stream()),
                        "softmax_ce_bwd")
        return dl, None, None


def cross_entropy(criterion, logits: torch.Tensor, label: torch.Tensor):
    """criterion(logits, label) for a plain mean-reduced nn.CrossEntropyLoss on channels-last logits -- logits is the
    (B, classes, H, W) VIEW of a contiguous (B, H, W, classes) tensor -- or None when this path does not apply.
    Labels outside [0, classes) that are not ``ignore_index`` are treated as ignored (csrc/pointwise.hip), where
    torch's kernel device-asserts: the reference's datasets map every unlabeled pixel to 255 = ignore_index
    (dataloader/RGBXDataset.py), so such labels do not occur on this path."""
    if not (type(criterion) is nn.CrossEntropyLoss and criterion.reduction == "mean" and criterion.weight is None
            and getattr(criterion, "label_smoothing", 0.0) == 0.0):
        return None
    if not (logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 4 and label.dim() == 3 and label.is_cuda):
        return None
    nhwc = logits.permute(0, 2, 3, 1)
    nc = nhwc.shape[-1]
    if not nhwc.is_contiguous() or nc % 4 != 0 or nhwc.data_ptr() % 16 != 0 or tuple(label.shape) != tuple(nhwc.shape[:3]):
        return None
    lab = label.long().contiguous()
    return SoftmaxCEFn.apply(nhwc.reshape(-1, nc), lab.view(-1), criterion.ignore_index)
