"""fp32 GEMMs of the hot path on the hand-written split-operand bf16 MFMA kernels (csrc/gemm_split.hip, C ABI
include/sigma_gemm.h).

The reference computes every projection as an fp32 GEMM (nn.Linear: SS2D.in_proj / out_proj ``vmamba.py:1067-1089``,
PatchMerging2D.reduction ``:612-636``, the CroMB / ConMB and decoder linears).  On MI355X fp32 GEMMs are bound by the
fp32 MFMA rate (157 TFLOP/s); the kernels here split every fp32 operand element in registers into two bf16 halves and
issue three bf16 MFMAs per tile (fp32 accumulate): ~4e-6 rms error against fp64 (fp32 GEMM ~1e-6), the model meets the
reference fixtures at unchanged tolerances (tests/test_gemm_gpu.py, tests/test_model_gpu.py).

    y  = x W^T (+ b)     ``gemm_nt``   x (M, K), W (N, K)
    dx = dy W            ``gemm_nn``   dy (M, N), W (N, K) read in place (no transposed copy)
    dW = dy^T x          ``gemm_tn``   reduction over the M tokens, split over workgroups, fp32 atomics

There is no CPU fallback: CPU tensors raise.  Shapes the kernels do not take (K % 4 != 0, unaligned rows) go to the
vendor fp32 GEMM (``torch.mm``), which is the round-2 path.  ``SIGMA_GEMM=fp32`` switches the module patching off.

Precision (profiles/r03_grad_precision.jsonl, tools/grad_precision.py; reference fixtures tests/golden/model_*.npz):
two bf16 pieces per operand (three MFMAs) give logits within 3e-5 of the reference (north star: 1e-3) and gradient
digests within 1.3e-3 (test bound 5e-3); three pieces (six MFMAs, ``SIGMA_GEMM_FWD / _DGRAD / _WGRAD = 3``) reproduce the
fp32 GEMM (logits 2e-6).  Element-wise gradient agreement below ~5e-3 of a tensor's scale cannot be promised by ANY
change of GEMM rounding on the small fixtures: ChannelAttention's global max pool (vmamba.py:1725-1741) routes its
gradient to the arg-max position, so a near-tie flipped by a 1e-6 perturbation moves gradients discontinuously (the
three-piece forward, more accurate than the two-piece one, flips one on the 72x88 fixture and none on the 64x96 one;
the two-piece forward the other way round).
"""
from __future__ import annotations

import ctypes
import os
import types

import torch
import torch.nn as nn

from . import _capi

MIN_DIM = 32               # smaller projections are launch / HBM bound either way


def _params(M, N, K, A, Bt, C, bias, lda, ldb, ldc, accumulate=False, batch=1, sA=0, sB=0, sC=0, a_mod=0, pieces=2):
    p = _capi.GemmParams()
    p.M, p.N, p.K = int(M), int(N), int(K)
    p.A, p.Bt, p.C = A.data_ptr(), Bt.data_ptr(), C.data_ptr()
    p.bias = bias.data_ptr() if bias is not None else None
    p.lda, p.ldb, p.ldc = int(lda), int(ldb), int(ldc)
    p.accumulate, p.batch = int(bool(accumulate)), int(batch)
    p.strideA, p.strideB, p.strideC = int(sA), int(sB), int(sC)
    p.a_mod = int(a_mod)
    p.pieces = int(pieces)
    return p


def _run(name, p, dev):
    with torch.cuda.device(dev):
        rc = getattr(_capi.load(), name)(ctypes.byref(p), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        raise RuntimeError(f"{name} failed (sigma_ops status {rc}): M={p.M} N={p.N} K={p.K} lda={p.lda} ldb={p.ldb}")


def _check2d(*ts):
    for t in ts:
        if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2:
            raise RuntimeError("sigma_amd.gemm: 2-D fp32 GPU tensors only (no fallback)")


def _rows_ok(t: torch.Tensor) -> bool:
    """last dimension contiguous, 16-byte aligned rows"""
    return t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0


def nt_ok(a: torch.Tensor, bt: torch.Tensor) -> bool:
    return a.shape[1] % 4 == 0 and _rows_ok(a) and _rows_ok(bt)


def gemm_nt(a: torch.Tensor, bt: torch.Tensor, bias=None, out=None, accumulate=False, pieces: int = 2) -> torch.Tensor:
    """out (M, N) (+)= a (M, K) @ bt (N, K)^T (+ bias).  pieces: bf16 pieces per operand element -- 2 = three MFMAs per
    block (~4e-6 rms error), 3 = six MFMAs (~1e-6, the accuracy of an fp32 GEMM)"""
    _check2d(a, bt)
    M, K = a.shape
    N = bt.shape[0]
    if bt.shape[1] != K or not nt_ok(a, bt):
        raise RuntimeError("gemm_nt: operands must be (M, K) and (N, K) with K % 4 == 0 and 16-byte aligned rows")
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.float32)
    elif out.stride(1) != 1 or tuple(out.shape) != (M, N):
        raise RuntimeError("gemm_nt: out must be (M, N) with contiguous rows")
    if M and N:
        _run("sigma_gemm_nt_split3", _params(M, N, K, a, bt, out, bias, a.stride(0), bt.stride(0), out.stride(0), accumulate, pieces=pieces), a.device)
    return out


def nn_ok(a: torch.Tensor, b: torch.Tensor) -> bool:
    return a.shape[1] % 4 == 0 and b.shape[1] % 4 == 0 and _rows_ok(a) and _rows_ok(b)


def gemm_nn(a: torch.Tensor, b: torch.Tensor, out=None, accumulate=False, pieces: int = 2) -> torch.Tensor:
    """out (M, N) (+)= a (M, K) @ b (K, N), b read row-major in place"""
    _check2d(a, b)
    M, K = a.shape
    N = b.shape[1]
    if b.shape[0] != K or not nn_ok(a, b):
        raise RuntimeError("gemm_nn: operands must be (M, K) and (K, N) with K % 4 == 0, N % 4 == 0 and 16-byte aligned rows")
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.float32)
    elif out.stride(1) != 1 or tuple(out.shape) != (M, N):
        raise RuntimeError("gemm_nn: out must be (M, N) with contiguous rows")
    if M and N:
        _run("sigma_gemm_nn_split3", _params(M, N, K, a, b, out, None, a.stride(0), b.stride(0), out.stride(0), accumulate, pieces=pieces), a.device)
    return out


def tn_ok(a: torch.Tensor, b: torch.Tensor) -> bool:
    return a.shape[1] % 4 == 0 and b.shape[1] % 4 == 0 and _rows_ok(a) and _rows_ok(b)


def gemm_tn(a: torch.Tensor, b: torch.Tensor, out=None, accumulate=False, pieces: int = 2) -> torch.Tensor:
    """out (N, K) (+)= a (M, N)^T @ b (M, K): the reduction runs over the rows (tokens) of both operands"""
    _check2d(a, b)
    M, N = a.shape
    K = b.shape[1]
    if b.shape[0] != M or not tn_ok(a, b):
        raise RuntimeError("gemm_tn: operands must be (M, N) and (M, K) with N % 4 == 0, K % 4 == 0 and 16-byte aligned rows")
    if out is None:
        out = torch.zeros((N, K), device=a.device, dtype=torch.float32)      # slices are summed with atomics
    elif out.stride(1) != 1 or tuple(out.shape) != (N, K):
        raise RuntimeError("gemm_tn: out must be (N, K) with contiguous rows")
    elif not accumulate:
        out.zero_()
        accumulate = True
    if N and K:
        _run("sigma_gemm_tn_split3", _params(M, N, K, a, b, out, None, a.stride(0), b.stride(0), out.stride(0), accumulate, pieces=pieces), a.device)
    return out


def _pieces(var: str, default: str) -> int:
    """per-GEMM-kind precision knob: '2' / '3' = bf16 pieces on the MFMA kernels, 'fp32' (-> 0) = vendor fp32 GEMM"""
    v = os.environ.get(var, default)
    return 0 if v == "fp32" else int(v)


_FWD = _pieces("SIGMA_GEMM_FWD", "2")
_DGRAD = _pieces("SIGMA_GEMM_DGRAD", "2")
_WGRAD = _pieces("SIGMA_GEMM_WGRAD", "2")


class LinearSplit3Fn(torch.autograd.Function):
    """F.linear(x2, weight, bias) for a 2-D x2 on the split-operand MFMA kernels.  2-D in, 2-D out: the output must
    not be a view made inside the Function (in-place activations follow some linears)."""

    @staticmethod
    def forward(ctx, x2, weight, bias):
        y = gemm_nt(x2, weight, bias, pieces=_FWD) if _FWD else nn.functional.linear(x2, weight, bias)
        ctx.save_for_backward(x2, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, weight = ctx.saved_tensors
        g2 = dy if _rows_ok(dy) else dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = gemm_nn(g2, weight, pieces=_DGRAD) if (_DGRAD and nn_ok(g2, weight)) else torch.mm(g2, weight)
        if ctx.needs_input_grad[1]:
            dw = gemm_tn(g2, x2, pieces=_WGRAD) if (_WGRAD and tn_ok(g2, x2)) else torch.mm(g2.t(), x2)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = g2.sum(0)
        return dx, dw, db


def linear(x: torch.Tensor, weight: torch.Tensor, bias=None) -> torch.Tensor:
    """nn.functional.linear on the split-operand kernels where they apply (fp32 GPU tensors, aligned rows)."""
    x2 = x.reshape(-1, x.shape[-1])
    if not (x2.is_cuda and x2.dtype == torch.float32 and weight.dtype == torch.float32 and nt_ok(x2, weight)):
        if not x2.is_cuda:
            raise RuntimeError("sigma_amd.gemm.linear: GPU tensors only (no fallback)")
        return nn.functional.linear(x, weight, bias)
    y2 = LinearSplit3Fn.apply(x2, weight, bias)
    return y2.view(*x.shape[:-1], weight.shape[0])


def _forward(self, x):
    if x.is_cuda and x.dtype == torch.float32 and self.weight.dtype == torch.float32:
        return linear(x, self.weight, self.bias)
    return nn.functional.linear(x, self.weight, self.bias)


def enable_split3_linears(model: nn.Module, min_dim: int = MIN_DIM) -> int:
    """Route every nn.Linear of `model` with in_features and out_features >= min_dim through LinearSplit3Fn (module
    classes, parameter names and state-dict keys unchanged).  Returns the number of layers switched."""
    n = 0
    for m in model.modules():
        if isinstance(m, nn.Linear) and m.in_features >= min_dim and m.out_features >= min_dim and m.in_features % 4 == 0:
            m.forward = types.MethodType(_forward, m)
            n += 1
    return n


def gemm_mode() -> str:
    """'split3' (default): the hand-written kernels; 'fp32': vendor fp32 GEMMs (the round-2 path)."""
    return os.environ.get("SIGMA_GEMM", "split3")
