"""Layout changes of the fusion blocks and the decoder as tiled transposes (csrc/merge.hip: transpose2d_kernel).

The reference moves activations between token-major (B, H, W, C) and channel-major (B, C, H, W) / (B, d, L) with
``permute(...).contiguous()`` and ``transpose(1, 2)`` in front of LayerNorm (vmamba.py:1265-1284 ConMB, :1622-1640 CroMB,
:1800-1805 CVSS, :1507-1545 the cross-SSM out_norms).  ATen's generic strided copy runs these at ~1 TB/s on MI355X
(profiles/r03_aten_tail_by_node.txt: 118 MB in 210 us) and autograd mirrors each one with another strided copy or a
strided add; the 32 x 32 LDS tile transpose of this library moves the same bytes at 3.7 TB/s, forward and backward.

    channels_first(x)   (B, H, W, C) -> (B, C, H, W) contiguous
    channels_last(x)    (B, C, H, W) -> (B, H, W, C) contiguous
    transpose_rows(x)   (B, R, C) with unit column stride (row / batch strides free, e.g. one half of a sequence)
                        -> (B, C, R) contiguous

CPU tensors (host-logic tests) take the torch ops: a layout change is not arithmetic.
"""
from __future__ import annotations

import torch

from .ss2d_fused import _transpose2d


class _TransposeRowsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        B, R, C = x.shape
        out = torch.empty(B, C, R, device=x.device, dtype=torch.float32)
        if x.numel():
            _transpose2d(x, out, B, R, C, x.stride(0), x.stride(1), C * R, R)
        return out

    @staticmethod
    def backward(ctx, g):
        B, C, R = g.shape
        if g.stride(2) != 1 or g.dtype != torch.float32:
            g = g.float().contiguous()
        gx = torch.empty(B, R, C, device=g.device, dtype=torch.float32)
        if g.numel():
            _transpose2d(g, gx, B, C, R, g.stride(0), g.stride(1), R * C, C)
        return gx


def transpose_rows(x: torch.Tensor) -> torch.Tensor:
    """(B, R, C) -> (B, C, R) contiguous"""
    if x.dim() != 3:
        raise RuntimeError("transpose_rows: 3-D tensors only")
    if not x.is_cuda:
        return x.transpose(1, 2).contiguous()
    if x.dtype != torch.float32 or x.stride(2) != 1 or (x.shape[0] > 1 and x.stride(0) < 0) or x.stride(1) < 0:
        x = x.float().contiguous()
    return _TransposeRowsFn.apply(x)


def channels_first(x: torch.Tensor) -> torch.Tensor:
    """(B, H, W, C) -> (B, C, H, W) contiguous"""
    B, H, W, C = x.shape
    if not x.is_cuda:
        return x.permute(0, 3, 1, 2).contiguous()
    return transpose_rows(x.reshape(B, H * W, C)).view(B, C, H, W)


def channels_last(x: torch.Tensor) -> torch.Tensor:
    """(B, C, H, W) -> (B, H, W, C) contiguous"""
    B, C, H, W = x.shape
    if not x.is_cuda:
        return x.permute(0, 2, 3, 1).contiguous()
    return transpose_rows(x.reshape(B, C, H * W)).view(B, H, W, C)
