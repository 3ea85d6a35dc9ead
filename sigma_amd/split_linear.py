"""Split-operand bf16 GEMMs for the nn.Linear layers of the model (opt-in: ``SIGMA_SPLIT_GEMM=1`` or
``enable_split_linears(model)``; the default path stays rocBLAS / hipBLASLt fp32).

The reference computes every ``nn.Linear`` (SS2D.in_proj / out_proj ``vmamba.py:1067-1089``, PatchMerging2D.reduction
``:612-636``, the fusion and decoder linears) as an fp32 GEMM.  On MI355X those run at 65-85 % of the fp32 MFMA peak
(157 TFLOP/s) and are a quarter of the training step.  bf16 MFMA is 16x faster, but bf16 operands miss the 1e-3 logit
bar (``profiles/r02_gemm_precision.jsonl``).  Here every fp32 operand is split exactly into two bf16 halves,
x = hi + lo + O(2^-17 |x|), and

    a @ b  ~=  a_hi @ b_hi + a_hi @ b_lo + a_lo @ b_hi          (dropped: a_lo @ b_lo ~ 2^-16 relative)

is ONE bf16 GEMM with fp32 accumulation and fp32 output on operands concatenated along the reduction dimension,
[a_hi | a_hi | a_lo] @ [b_hi ; b_lo ; b_hi].  Measured (``profiles/r02_split_gemm_probe.jsonl``): 4.4e-6 rms error
against an fp64 product (the fp32 GEMM itself: ~1e-6), 2.1-3.3x faster than the fp32 GEMM.  The operand images are
written by one HIP kernel (``csrc/split.hip``); the GEMMs are hipBLASLt through ``torch.mm(..., out_dtype=float32)``.

forward : y  = [x_h | x_h | x_l] @ [W_h | W_l | W_h]^T            (M, 3K) x (3K, N)
backward: dx = [g_h | g_h | g_l] @ [W_h ; W_l ; W_h]              (M, 3N) x (3N, K)
          dW = g^T x                                              fp32 GEMM as before: the reduction runs over the
          M = 19200 .. 614400 tokens, where three bf16 GEMMs (115 us each at (1536 x 19200) x (19200 x 384)) lose to
          the one fp32 GEMM (224 us) -- profiles/r02_step_profile_split.txt
"""
from __future__ import annotations

import ctypes
import os
import types

import torch
import torch.nn as nn

from . import _capi

MIN_FEATURES = 64          # smaller reductions are launch-bound either way


def worth_splitting(reduce_dim: int, out_dim: int) -> bool:
    """Splitting the token-side operand costs a pass over M x reduce_dim (4 B read + 6 B written per element); the GEMM
    it accelerates has M x reduce_dim x out_dim products.  Measured on sigma_small's shapes
    (profiles/r02_split_gemm_probe.jsonl, r02_step_profile_split.txt): a win from out_dim >= reduce_dim on (in_proj
    forward 177 -> 20 + 77 us, out_proj backward 110 -> 20 + 42 us), a wash or a loss below (in_proj backward
    168 -> 70 + 97 us)."""
    return out_dim >= reduce_dim and reduce_dim >= MIN_FEATURES


def _split(src2d: torch.Tensor, layout: str) -> torch.Tensor:
    """bf16 image of a 2-D fp32 tensor: 'hhl' -> (R, 3C) [hi | hi | lo]; 'hlh' -> (R, 3C) [hi | lo | hi];
    'h;l;h' -> (3R, C) [hi ; lo ; hi]."""
    if src2d.dtype != torch.float32 or not src2d.is_cuda:
        raise RuntimeError("split_linear: fp32 GPU tensors only (no fallback)")
    if src2d.stride(1) != 1:
        src2d = src2d.contiguous()
    R, C = src2d.shape
    if layout == "h;l;h":
        dst = torch.empty((3 * R, C), device=src2d.device, dtype=torch.bfloat16)
        dst_rs, hi2, lo = C, 2 * R * C, R * C
    else:
        dst = torch.empty((R, 3 * C), device=src2d.device, dtype=torch.bfloat16)
        dst_rs = 3 * C
        hi2, lo = (C, 2 * C) if layout == "hhl" else (2 * C, C)
    if R and C:
        stream = torch.cuda.current_stream(src2d.device).cuda_stream
        rc = _capi.load().sigma_split_bf16(ctypes.c_void_p(src2d.data_ptr()), R, C, src2d.stride(0), ctypes.c_void_p(dst.data_ptr()),
                                           dst_rs, hi2, lo, ctypes.c_void_p(stream))
        if rc != 0:
            raise RuntimeError(f"sigma_split_bf16 failed (status {rc})")
    return dst


class SplitLinearFn(torch.autograd.Function):
    """F.linear(x2, weight, bias) for a 2-D x2 with split-operand bf16 GEMMs (see the module docstring).  2-D in, 2-D
    out: the output must not be a view made inside the Function (in-place activations follow some linears)."""

    @staticmethod
    def forward(ctx, x2, weight, bias):
        K = x2.shape[-1]
        N = weight.shape[0]
        if worth_splitting(K, N):
            A = _split(x2, "hhl")                                    # (M, 3K)
            Wf = _split(weight, "hlh")                               # (N, 3K)
            y = torch.mm(A, Wf.t(), out_dtype=torch.float32)
            if bias is not None:
                y += bias
        else:
            y = nn.functional.linear(x2, weight, bias)
        ctx.save_for_backward(x2, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, weight = ctx.saved_tensors
        g2 = dy
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            N, K = weight.shape
            if worth_splitting(N, K):
                G = _split(g2, "hhl")                                # (M, 3N)
                Wb = _split(weight, "h;l;h")                         # (3N, K)
                dx = torch.mm(G, Wb, out_dtype=torch.float32)
            else:
                dx = torch.mm(g2, weight)
        if ctx.needs_input_grad[1]:
            dw = torch.mm(g2.t(), x2)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = g2.sum(0)
        return dx, dw, db


def split_linear(x: torch.Tensor, weight: torch.Tensor, bias=None) -> torch.Tensor:
    y2 = SplitLinearFn.apply(x.reshape(-1, x.shape[-1]), weight, bias)
    return y2.view(*x.shape[:-1], weight.shape[0])


def _forward(self, x):
    if x.is_cuda and x.dtype == torch.float32 and self.weight.dtype == torch.float32:
        return split_linear(x, self.weight, self.bias)
    return nn.functional.linear(x, self.weight, self.bias)


def enable_split_linears(model: nn.Module, min_features: int = MIN_FEATURES) -> int:
    """Route every nn.Linear of `model` with in_features and out_features >= min_features through SplitLinearFn
    (module classes, parameter names and state-dict keys unchanged).  Returns the number of layers switched."""
    n = 0
    for m in model.modules():
        if isinstance(m, nn.Linear) and m.in_features >= min_features and m.out_features >= min_features:
            m.forward = types.MethodType(_forward, m)
            n += 1
    return n


def split_gemm_requested() -> bool:
    return os.environ.get("SIGMA_SPLIT_GEMM", "0") == "1"
