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


def _params(M, N, K, A, Bt, C, bias, lda, ldb, ldc, accumulate=False, batch=1, sA=0, sB=0, sC=0, a_mod=0, pieces=2, c_mod=0,
            residual=None, residual2=None, ldr=0, sR=0, out_t=None, ldct=0, t_cols=0, k_slices=0):
    p = _capi.GemmParams()
    p.M, p.N, p.K = int(M), int(N), int(K)
    p.A, p.Bt, p.C = A.data_ptr(), Bt.data_ptr(), C.data_ptr()
    p.bias = bias.data_ptr() if bias is not None else None
    p.lda, p.ldb, p.ldc = int(lda), int(ldb), int(ldc)
    p.accumulate, p.batch = int(bool(accumulate)), int(batch)
    p.strideA, p.strideB, p.strideC = int(sA), int(sB), int(sC)
    p.a_mod = int(a_mod)
    p.pieces = int(pieces)
    p.c_mod = int(c_mod)
    p.residual = residual.data_ptr() if residual is not None else None
    p.residual2 = residual2.data_ptr() if residual2 is not None else None
    p.ldr, p.strideR = int(ldr), int(sR)
    p.Ct = out_t.data_ptr() if out_t is not None else None
    p.ldct, p.t_cols, p.k_slices = int(ldct), int(t_cols), int(k_slices)
    return p


_SELFTESTED = set()


def _selftest(dev) -> None:
    """Once per process and device: sigma_gemm_selftest (csrc/gemm_split.hip) runs the three kernel forms on operands
    whose products are exact in fp32 and compares with host arithmetic -- the operand loads are hand-counted inline
    assembly, so a toolchain that schedules them differently must fail HERE, loudly."""
    key = (dev.index if dev.index is not None else torch.cuda.current_device())
    if key in _SELFTESTED:
        return
    if torch.cuda.is_current_stream_capturing():
        # the self test allocates, copies from pageable host memory and synchronises: all illegal inside a capture
        raise RuntimeError("sigma_amd.gemm: the first split-operand GEMM of this process on this device was issued inside a "
                           "stream capture; run one step (or sigma_amd.gemm.selftest(device)) eagerly before capturing")
    with torch.cuda.device(dev):
        rc = _capi.load().sigma_gemm_selftest(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        raise RuntimeError(f"sigma_gemm_selftest failed on {dev} (code {rc}): the split-operand GEMM kernels of this build do not "
                           f"compute A B^T exactly on an exactly representable problem; rebuild libsigma_hip.so with the supported ROCm")
    _SELFTESTED.add(key)


def selftest(device=None) -> None:
    """run the load-time self test of the GEMM kernels now (idempotent) -- before a capture whose first GEMM would
    otherwise trigger it"""
    _selftest(torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device))


_FORMS = {"sigma_gemm_nt_split3": 0, "sigma_gemm_nn_split3": 1, "sigma_gemm_tn_split3": 2}
# A/B knob: SIGMA_GEMM_TWO_STAGE=0 keeps the round-5 path -- atomic sums into an output the wrappers zero-fill first
_TWO_STAGE = os.environ.get("SIGMA_GEMM_TWO_STAGE", "1") != "0"


def _atomic_path(out: torch.Tensor, accumulate: bool) -> bool:
    """A/B path only: the output of a possibly sliced launch is zero-filled here and then added into"""
    if not accumulate:
        out.zero_()
    return True


def _run(name, p, dev):
    _selftest(dev)
    lib = _capi.load()
    # Launches whose work items do not each own their output (reduction slices of a weight gradient, problems sharing an
    # output) get the scratch for the two-stage sum: partial results stored plainly + one reduce kernel, instead of 64
    # dword atomics per thread and item (half of such a launch, round 6) -- and the output needs no zero fill.  The
    # buffer comes from torch's caching allocator and goes back when this call returns: stream-ordered, so the kernels
    # enqueued here finish with it before a later allocation on this stream can reuse it.
    need = int(lib.sigma_gemm_workspace_bytes(ctypes.byref(p), _FORMS[name])) if _TWO_STAGE else 0   # < 0: bad arguments -- the call below says so
    ws = None
    if need > 0:
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        p.workspace, p.workspace_bytes = ws.data_ptr(), need
    with torch.cuda.device(dev):
        rc = getattr(lib, name)(ctypes.byref(p), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    del ws
    if rc != 0:
        raise RuntimeError(f"{name} failed (sigma_ops status {rc}): M={p.M} N={p.N} K={p.K} lda={p.lda} ldb={p.ldb}")


def _check2d(*ts):
    for t in ts:
        if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2:
            raise RuntimeError("sigma_amd.gemm: 2-D fp32 GPU tensors only (no fallback)")


_MAX_LD = 1 << 22          # the kernels address a tile with 32-bit byte offsets: 127 rows x ld x 4 bytes < 2^32 (capi: same bound)


def _rows_ok(t: torch.Tensor) -> bool:
    """last dimension contiguous, 16-byte aligned rows, row stride inside the kernels' 32-bit tile offsets"""
    return t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0 and 0 < t.stride(0) <= _MAX_LD


def _check_aux(name, bias, out, N, dev):
    """bias / out are read and written as raw fp32 by the kernels: refuse anything else"""
    if bias is not None and not (bias.is_cuda and bias.device == dev and bias.dtype == torch.float32 and bias.dim() == 1
                                 and bias.numel() == N and bias.is_contiguous()):
        raise RuntimeError(f"{name}: bias must be a contiguous fp32 vector of {N} elements on {dev}")
    if out is not None and not (out.is_cuda and out.device == dev and out.dtype == torch.float32 and out.dim() == 2):
        raise RuntimeError(f"{name}: out must be a 2-D fp32 tensor on {dev}")


def nt_ok(a: torch.Tensor, bt: torch.Tensor) -> bool:
    return a.shape[1] % 4 == 0 and _rows_ok(a) and _rows_ok(bt)


def residual_ok(r, M: int, N: int) -> bool:
    """an epilogue addend: fp32 (M, N) with contiguous rows, small enough for 32-bit element offsets, two pieces only"""
    return (r is not None and r.is_cuda and r.dtype == torch.float32 and r.dim() == 2 and tuple(r.shape) == (M, N)
            and r.stride(1) == 1 and 0 < r.stride(0) and M * r.stride(0) < 2 ** 31 - 1)


def gemm_nt(a: torch.Tensor, bt: torch.Tensor, bias=None, out=None, accumulate=False, pieces: int = 2, residual=None) -> torch.Tensor:
    """out (M, N) (+)= a (M, K) @ bt (N, K)^T (+ bias) (+ residual).  pieces: bf16 pieces per operand element -- 2 =
    three MFMAs per block (~4e-6 rms error), 3 = six MFMAs (~1e-6, the accuracy of an fp32 GEMM).  ``residual``: an
    (M, N) tensor added in the kernel (the accumulators start from it), two pieces only."""
    _check2d(a, bt)
    M, K = a.shape
    N = bt.shape[0]
    if bt.shape[1] != K or not nt_ok(a, bt):
        raise RuntimeError("gemm_nt: operands must be (M, K) and (N, K) with K % 4 == 0 and 16-byte aligned rows")
    _check_aux("gemm_nt", bias, out, N, a.device)
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.float32)
    elif out.stride(1) != 1 or tuple(out.shape) != (M, N):
        raise RuntimeError("gemm_nt: out must be (M, N) with contiguous rows")
    if residual is not None and (pieces != 2 or not residual_ok(residual, M, N)):
        raise RuntimeError("gemm_nt: residual must be an fp32 (M, N) GPU tensor with contiguous rows (two-piece kernels only)")
    if M and N:
        _run("sigma_gemm_nt_split3", _params(M, N, K, a, bt, out, bias, a.stride(0), bt.stride(0), out.stride(0), accumulate, pieces=pieces,
                                             residual=residual, ldr=residual.stride(0) if residual is not None else 0), a.device)
    return out


def nn_ok(a: torch.Tensor, b: torch.Tensor) -> bool:
    return a.shape[1] % 4 == 0 and b.shape[1] % 4 == 0 and _rows_ok(a) and _rows_ok(b)


def gemm_nn(a: torch.Tensor, b: torch.Tensor, out=None, accumulate=False, pieces: int = 2, k_slices: bool = False) -> torch.Tensor:
    """out (M, N) (+)= a (M, K) @ b (K, N), b read row-major in place.  ``k_slices``: few output tiles and a long reduction
    (a weight gradient whose left operand is held channel-major): the kernel may cut K into slices summed with fp32 atomics;
    (two-stage sum through scratch: ``out`` is written, or added to with ``accumulate``)."""
    _check2d(a, b)
    M, K = a.shape
    N = b.shape[1]
    if b.shape[0] != K or not nn_ok(a, b):
        raise RuntimeError("gemm_nn: operands must be (M, K) and (K, N) with K % 4 == 0, N % 4 == 0 and 16-byte aligned rows")
    _check_aux("gemm_nn", None, out, N, a.device)
    if out is None:
        if accumulate:
            raise RuntimeError("gemm_nn: accumulate needs out")
        out = torch.empty((M, N), device=a.device, dtype=torch.float32)
    elif out.stride(1) != 1 or tuple(out.shape) != (M, N):
        raise RuntimeError("gemm_nn: out must be (M, N) with contiguous rows")
    if M and N:
        if k_slices and not _TWO_STAGE:
            accumulate = _atomic_path(out, accumulate)
        _run("sigma_gemm_nn_split3", _params(M, N, K, a, b, out, None, a.stride(0), b.stride(0), out.stride(0),
                                             accumulate, pieces=pieces, k_slices=1 if k_slices else 0), a.device)
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
    _check_aux("gemm_tn", None, out, K, a.device)
    if M == 0:                                               # no tokens: the gradient is zero (or what `out` holds)
        if out is None:
            return torch.zeros((N, K), device=a.device, dtype=torch.float32)
        return out if accumulate else out.zero_()
    if out is None:
        if accumulate:
            raise RuntimeError("gemm_tn: accumulate needs out")
        out = torch.empty((N, K), device=a.device, dtype=torch.float32)      # written by the kernel(s): slices are summed in two stages
    elif out.stride(1) != 1 or tuple(out.shape) != (N, K):
        raise RuntimeError("gemm_tn: out must be (N, K) with contiguous rows")
    if N and K:
        if not _TWO_STAGE:
            accumulate = _atomic_path(out, accumulate)
        _run("sigma_gemm_tn_split3", _params(M, N, K, a, b, out, None, a.stride(0), b.stride(0), out.stride(0), accumulate, pieces=pieces), a.device)
    return out


def _check3d(name, *ts):
    for t in ts:
        if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 3 and t.stride(2) == 1 and t.stride(1) % 4 == 0
                and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0):
            raise RuntimeError(f"{name}: 3-D fp32 GPU tensors (problem, row, column) with contiguous, 16-byte aligned rows and "
                               f"batch strides that are multiples of 4 (no fallback); got shape {tuple(t.shape)} strides {t.stride()}")


def bgemm_ok(*ts) -> bool:
    return all(t.is_cuda and t.dtype == torch.float32 and t.dim() == 3 and t.stride(2) == 1 and t.stride(1) % 4 == 0
               and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0 and t.shape[2] % 4 == 0 for t in ts)


def bgemm_nn(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, residual=None, residual2=None, pieces: int = 2) -> torch.Tensor:
    """Stacked problems  out[z] (M, N) = a[z % Za] (M, K) @ b[z] (K, N) (+ residual[z] (+ residual2[z])):  a (Za, M, K) is
    a weight stack shared by groups of problems (Za divides Z), b (Z, K, N) and out (Z, M, N) may be strided views of
    larger tensors (row slices of a stacked projection).  The x_proj / dt_proj einsums of cross_selective_scan on
    channels-first activations (vmamba.py:193-199) and their input gradients."""
    _check3d("bgemm_nn", a, b, out)
    Za, M, K = a.shape
    Z, Kb, N = b.shape
    if Kb != K or tuple(out.shape) != (Z, M, N) or Z % Za != 0 or K % 4 != 0 or N % 4 != 0:
        raise RuntimeError(f"bgemm_nn: shapes a {tuple(a.shape)} b {tuple(b.shape)} out {tuple(out.shape)}")
    kw = {}
    if residual is not None:
        _check3d("bgemm_nn residual", residual)
        if tuple(residual.shape) != (Z, M, N) or M * residual.stride(1) >= 2 ** 31 - 1 or pieces != 2:
            raise RuntimeError("bgemm_nn: residual must be (Z, M, N), two-piece kernels only")
        if residual2 is not None and (tuple(residual2.shape) != (Z, M, N) or residual2.stride() != residual.stride()
                                      or residual2.dtype != torch.float32):
            raise RuntimeError("bgemm_nn: residual2 must have the shape and strides of residual")
        kw = dict(residual=residual, residual2=residual2, ldr=residual.stride(1), sR=residual.stride(0))
    elif residual2 is not None:
        raise RuntimeError("bgemm_nn: residual2 without residual")
    if Z and M and N:
        _run("sigma_gemm_nn_split3", _params(M, N, K, a, b, out, None, a.stride(1), b.stride(1), out.stride(1), False, batch=Z,
                                             sA=a.stride(0), sB=b.stride(0), sC=out.stride(0), a_mod=Za if Za != Z else 0,
                                             pieces=pieces, **kw), a.device)
    return out


def bgemm_nt_sum(a: torch.Tensor, bt: torch.Tensor, out: torch.Tensor, pieces: int = 2, accumulate: bool = True) -> torch.Tensor:
    """out[z % Zc] (M, N) (+)= a[z] (M, K) @ bt[z] (N, K)^T summed over the problems that share an output (two-stage sum
    through scratch, fixed order): out (Zc, M, N) contiguous per problem; ``accumulate`` (default, the round-4 contract):
    added to what out holds, else out is overwritten and needs no zero fill; Zc divides Z.  The weight gradients of the
    stacked projections summed over the batch."""
    _check3d("bgemm_nt_sum", a, bt, out)
    Z, M, K = a.shape
    Zb, N, Kb = bt.shape
    Zc = out.shape[0]
    if Zb != Z or Kb != K or tuple(out.shape[1:]) != (M, N) or Z % max(Zc, 1) != 0 or K % 4 != 0:
        raise RuntimeError(f"bgemm_nt_sum: shapes a {tuple(a.shape)} bt {tuple(bt.shape)} out {tuple(out.shape)}")
    if Z and M and N:
        if not _TWO_STAGE:
            accumulate = _atomic_path(out, accumulate)
        _run("sigma_gemm_nt_split3", _params(M, N, K, a, bt, out, None, a.stride(1), bt.stride(1), out.stride(1), accumulate, batch=Z,
                                             sA=a.stride(0), sB=bt.stride(0), sC=out.stride(0), c_mod=Zc, pieces=pieces), a.device)
    return out


def _pieces(var: str, default: str) -> int:
    """per-GEMM-kind precision knob: '2' / '3' = bf16 pieces on the MFMA kernels, 'fp32' (-> 0) = vendor fp32 GEMM"""
    v = os.environ.get(var, default)
    return 0 if v == "fp32" else int(v)


_FWD = _pieces("SIGMA_GEMM_FWD", "2")
_DGRAD = _pieces("SIGMA_GEMM_DGRAD", "2")
_WGRAD = _pieces("SIGMA_GEMM_WGRAD", "2")


class LinearSplit3Fn(torch.autograd.Function):
    """F.linear(x2, weight, bias) (+ residual) for a 2-D x2 on the split-operand MFMA kernels.  2-D in, 2-D out: the
    output must not be a view made inside the Function (in-place activations follow some linears).  ``residual``
    (optional, (M, out)): added inside the kernel; its gradient is the output gradient itself (no kernel)."""

    @staticmethod
    def forward(ctx, x2, weight, bias, residual=None):
        if residual is not None and not (_FWD == 2 and residual_ok(residual, x2.shape[0], weight.shape[0])):
            y = (gemm_nt(x2, weight, bias, pieces=_FWD) if _FWD else nn.functional.linear(x2, weight, bias)) + residual
        else:
            y = gemm_nt(x2, weight, bias, pieces=_FWD, residual=residual) if _FWD else nn.functional.linear(x2, weight, bias)
        ctx.save_for_backward(x2, weight)
        ctx.has_bias = bias is not None
        ctx.has_res = residual is not None
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
        return dx, dw, db, (dy if ctx.has_res and ctx.needs_input_grad[3] else None)


def linear(x: torch.Tensor, weight: torch.Tensor, bias=None, residual=None) -> torch.Tensor:
    """nn.functional.linear (+ residual, a tensor of the output's shape) on the split-operand kernels where they apply
    (fp32 GPU tensors, aligned rows)."""
    x2 = x.reshape(-1, x.shape[-1])
    if not (x2.is_cuda and x2.dtype == torch.float32 and weight.dtype == torch.float32 and nt_ok(x2, weight)):
        if not x2.is_cuda:
            raise RuntimeError("sigma_amd.gemm.linear: GPU tensors only (no fallback)")
        y = nn.functional.linear(x, weight, bias)
        return y if residual is None else y + residual
    r2 = None
    if residual is not None:
        if tuple(residual.shape) != (*x.shape[:-1], weight.shape[0]):
            raise RuntimeError("sigma_amd.gemm.linear: residual must have the shape of the output")
        r2 = residual.reshape(-1, weight.shape[0])
    y2 = LinearSplit3Fn.apply(x2, weight, bias, r2)
    return y2.view(*x.shape[:-1], weight.shape[0])


def xz_ok(x2: torch.Tensor, weight: torch.Tensor) -> bool:
    """the in_proj GEMM with a transposed x half: (M, C) @ (2d, C)^T with d % 32 == 0, M % 4 == 0, two bf16 pieces forward"""
    return (_FWD == 2 and x2.is_cuda and x2.dtype == torch.float32 and weight.dtype == torch.float32 and x2.dim() == 2
            and weight.shape[0] % 64 == 0 and x2.shape[0] % 4 == 0 and x2.shape[0] > 0 and nt_ok(x2, weight))


class LinearXZFn(torch.autograd.Function):
    """SS2D.in_proj with its two consumers' layouts (vmamba.py:1067-1071: ``xz = in_proj(x); x, z = xz.chunk(2, -1);
    x = x.permute(0, 3, 1, 2).contiguous()``):  (x2 (M, C), weight (2d, C), bias) -> xT (d, M) CHANNEL-major, z (M, d).

    One GEMM writes both: the columns of the x half leave the kernel transposed (sigma_gemm.h, t_cols), so the depthwise
    convolution reads (d, B, H, W) planes in place and the tiled transpose that round 3 put between the two (one read and one
    write of the activation per block, and again in the backward) is gone.  Backward, from dxT (d, M) -- written channel-major
    by the convolution's backward -- and dz (M, d):
        dx2 = dz W_z + dxT^T W_x        nn, then tn with accumulate (reduction over the d channels: both operands slow-indexed)
        dW  = [dxT x2 ; dz^T x2]        nn with the token reduction cut into slices (atomics), tn
        db  = [sum_m dxT ; sum_m dz]"""

    @staticmethod
    def forward(ctx, x2, weight, bias):
        M = x2.shape[0]
        d = weight.shape[0] // 2
        xT = torch.empty((d, M), device=x2.device, dtype=torch.float32)
        z = torch.empty((M, d), device=x2.device, dtype=torch.float32)
        _check_aux("linear_xz", bias, None, 2 * d, x2.device)
        _run("sigma_gemm_nt_split3", _params(M, 2 * d, x2.shape[1], x2, weight, z, bias, x2.stride(0), weight.stride(0), d, pieces=2,
                                             out_t=xT, ldct=M, t_cols=d), x2.device)
        ctx.save_for_backward(x2, weight)
        ctx.has_bias = bias is not None
        return xT, z

    @staticmethod
    def backward(ctx, dxT, dz):
        x2, weight = ctx.saved_tensors
        M, C = x2.shape
        d = weight.shape[0] // 2
        dxT = torch.zeros((d, M), device=x2.device, dtype=torch.float32) if dxT is None else (dxT if _rows_ok(dxT) else dxT.contiguous())
        dz = torch.zeros((M, d), device=x2.device, dtype=torch.float32) if dz is None else (dz if _rows_ok(dz) else dz.contiguous())
        w_x, w_z = weight[:d], weight[d:]
        dx2 = dw = db = None
        if ctx.needs_input_grad[0]:
            if _DGRAD and nn_ok(dz, w_z) and tn_ok(dxT, w_x):
                dx2 = gemm_nn(dz, w_z, pieces=_DGRAD)
                gemm_tn(dxT, w_x, out=dx2, accumulate=True, pieces=_DGRAD)
            else:
                dx2 = torch.mm(dz, w_z) + torch.mm(dxT.t(), w_x)
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(weight)
            if _WGRAD and nn_ok(dxT, x2) and tn_ok(dz, x2):
                gemm_nn(dxT, x2, out=dw[:d], pieces=_WGRAD, k_slices=True)
                gemm_tn(dz, x2, out=dw[d:], pieces=_WGRAD)
            else:
                dw[:d] = torch.mm(dxT, x2)
                dw[d:] = torch.mm(dz.t(), x2)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.cat([dxT.sum(1), dz.sum(0)])
        return dx2, dw, db


def linear_xz(x: torch.Tensor, weight: torch.Tensor, bias=None):
    """in_proj of an SS2D block on a channels-last (B, H, W, C) input: returns (xi, z) with xi the (B, d, H, W) VIEW of a
    channel-major (d, B, H, W) buffer (its H x W planes are contiguous: csrc/dwconv.hip reads them in place) and z the
    (B, H, W, d) gate tensor.  Callers check ``xz_ok`` first."""
    B, H, W, C = x.shape
    x2 = x.reshape(-1, C)
    d = weight.shape[0] // 2
    xT, z = LinearXZFn.apply(x2, weight, bias)
    return xT.view(d, B, H, W).permute(1, 0, 2, 3), z.view(B, H, W, d)


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
