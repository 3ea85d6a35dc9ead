"""Fused core of SS2D (``cross_selective_scan``, reference models/encoders/vmamba.py:165-226)
as ONE autograd function around the gfx950 scan kernels.

What the reference does per SS2D call (and what autograd then mirrors in backward):
``CrossScan`` materialises 4 permuted copies of x (vmamba.py:80-98), two einsums project every
copy (x_proj, dt_proj: :193-199), 5 ``.contiguous()/.float()`` copies feed the CUDA operator
(:201-207), ``CrossMerge`` un-flips / un-transposes and adds (:100-121).

Here the four directions are never materialised:

* only TWO physical copies of x exist (row-major and column-major order); the operator reads
  them for four groups through ``u_group_shift`` and runs the two flipped directions backwards by
  addressing (``rev_group_mask``) -- include/sigma_scan.h;
* x_proj / dt_proj act point-wise along the sequence, so they are applied to the two memory
  orders in natural order (they commute with the flip); the scan reads delta / B / C of a flipped
  direction at L-1-l.  B and C are strided views of the projection output, dB / dC are written by
  the kernel straight into the gradient of that output -- no split / cat / contiguous copies;
* directions are processed in the group order [0, 2, 1, 3] (both directions of one memory order
  adjacent) so that the projection output of a stacked-weight GEMM IS the (B, G, N, L) operand;
* CrossMerge is two adds and one transposed add; its adjoint hands the SAME gradient to the two
  directions of a memory order (``dout_group_shift``) instead of four flipped copies.

Parameter layout is the reference's (x_proj_weight (4, R+2N, d), dt_projs_weight (4, d, R),
dt_projs_bias (4, d), A_logs (4d, N), Ds (4d)); the permutation to kernel group order touches only
these small tensors.
"""
from __future__ import annotations

import torch

from . import selective_scan_cuda_core as _core

# kernel group g -> reference direction k.  g = 2*j + i with j = memory order (0 row-major,
# 1 column-major) and i = flipped; reference k = j + 2*i (vmamba.py:84-89).  Self-inverse.
_PERM = (0, 2, 1, 3)
_REV_MASK = 0b1010


def _two_orders(x4: torch.Tensor) -> torch.Tensor:
    """(B, d, H, W) -> (B, 2, d, L): [row-major, column-major] sequences of the same image."""
    B, d, H, W = x4.shape
    out = x4.new_empty(B, 2, d, H * W)
    out[:, 0].view(B, d, H, W).copy_(x4)
    out[:, 1].view(B, d, W, H).copy_(x4.transpose(2, 3))
    return out


class DwConvSiLUTwoOrdersFn(torch.autograd.Function):
    """xs2 = [silu(dwconv3x3(x)) row-major, the same column-major]  (B, d, H, W) -> (B, 2, d, H*W).

    Reference: SS2D.forward vmamba.py:1075-1077 + the layout half of CrossScan (:80-89); kernels in
    sigma_amd/csrc/dwconv.hip, C ABI in include/sigma_ops.h."""

    @staticmethod
    def forward(ctx, x, weight, bias, n_orders=2):
        import ctypes
        from . import _capi
        lib = _capi.load()
        if not x.is_cuda:
            raise RuntimeError("dwconv3x3_silu: GPU tensors only (no fallback)")
        x = x.float().contiguous()
        w = weight.float().contiguous()
        b = None if bias is None else bias.float().contiguous()
        B, d, H, W = x.shape
        if tuple(w.shape) != (d, 1, 3, 3):
            raise RuntimeError("dwconv3x3_silu: weight must be (d, 1, 3, 3)")
        out2 = torch.empty(B, n_orders, d, H * W, device=x.device, dtype=torch.float32)
        ctx.n_orders = n_orders
        p = _capi.DwConvParams()
        p.batch, p.channels, p.height, p.width, p.n_orders = B, d, H, W, n_orders
        p.x, p.weight, p.bias, p.out2 = x.data_ptr(), w.data_ptr(), (b.data_ptr() if b is not None else None), out2.data_ptr()
        with torch.cuda.device(x.device):
            _capi.check(lib.sigma_dwconv3x3_silu_fwd(ctypes.byref(p), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
                        "dwconv3x3_silu_fwd")
        ctx.save_for_backward(x, w, b if b is not None else x.new_empty(0))
        ctx.has_bias = b is not None
        return out2

    @staticmethod
    def backward(ctx, g2):
        import ctypes
        from . import _capi
        lib = _capi.load()
        x, w, b = ctx.saved_tensors
        B, d, H, W = x.shape
        g2 = g2.float().contiguous()
        gpre = torch.empty_like(x)
        dx = torch.empty_like(x)
        dw = torch.zeros_like(w)
        db = torch.zeros(d, device=x.device, dtype=torch.float32) if ctx.has_bias else None
        p = _capi.DwConvParams()
        p.batch, p.channels, p.height, p.width, p.n_orders = B, d, H, W, ctx.n_orders
        p.x, p.weight, p.bias = x.data_ptr(), w.data_ptr(), (b.data_ptr() if ctx.has_bias else None)
        p.g2, p.gpre, p.dweight, p.dx = g2.data_ptr(), gpre.data_ptr(), dw.data_ptr(), dx.data_ptr()
        p.dbias = db.data_ptr() if db is not None else None
        with torch.cuda.device(x.device):
            _capi.check(lib.sigma_dwconv3x3_silu_bwd(ctypes.byref(p), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
                        "dwconv3x3_silu_bwd")
        return dx, dw, db, None


def dwconv_silu_two_orders(x, weight, bias):
    return DwConvSiLUTwoOrdersFn.apply(x, weight, bias, 2)


def dwconv_silu(x, weight, bias):
    """silu(depthwise conv3x3(x) + bias), (B, d, H, W) -> (B, d, H, W); HIP kernel, GPU tensors only."""
    B, d, H, W = x.shape
    return DwConvSiLUTwoOrdersFn.apply(x, weight, bias, 1).view(B, d, H, W)


class SS2DCoreFn(torch.autograd.Function):
    """y = CrossMerge(selective_scan(CrossScan(x), ...)); input xs2 = [row-major, column-major]
    sequences of x, (B, 2, d, H*W) -> y (B, d, H*W)."""

    @staticmethod
    def forward(ctx, xs2, H, W, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds):
        B, _, d, L = xs2.shape
        if L != H * W:
            raise RuntimeError("SS2DCoreFn: xs2 must be (B, 2, d, H*W)")
        K, c, _ = x_proj_weight.shape
        R = dt_projs_weight.shape[2]
        N = A_logs.shape[1]
        if K != 4:
            raise RuntimeError("SS2DCoreFn expects the 4-direction parameter stack")
        xs2 = xs2.float().contiguous()
        perm = list(_PERM)
        Wst = x_proj_weight.float()[perm].reshape(2, 2 * c, d)                 # [order j][(flip i, row)][d]
        p4 = torch.matmul(Wst.unsqueeze(0), xs2).view(B, 4, c, L)              # == (B, group g, R+2N, L)
        dtw = dt_projs_weight.float()[perm]                                    # (4, d, R)
        delta = torch.matmul(dtw.unsqueeze(0), p4[:, :, :R])                   # (B, 4, d, L)
        A = (-torch.exp(A_logs.float())).view(4, d, N)[perm].reshape(4 * d, N)
        Dp = Ds.float().view(4, d)[perm].reshape(-1)
        bias = dt_projs_bias.float()[perm].reshape(-1)
        Bv, Cv = p4[:, :, R:R + N], p4[:, :, R + N:]
        need_x = any(ctx.needs_input_grad)
        out, ck = _core.fwd_ext(xs2.view(B, 2 * d, L), delta.view(B, 4 * d, L), A, Bv, Cv, Dp, bias, True,
                                rev_mask=_REV_MASK, u_gshift=1, need_x=need_x)
        ys = out.view(B, 4, d, L)
        y = ys[:, 0] + ys[:, 1]
        y_cm = ys[:, 2] + ys[:, 3]
        y += y_cm.view(B, d, W, H).transpose(2, 3).reshape(B, d, L)
        ctx.save_for_backward(xs2, p4, delta, A, Dp, bias, ck, Wst, dtw)
        ctx.dims = (B, d, H, W, c, R, N)
        return y

    @staticmethod
    def backward(ctx, dy):
        xs2, p4, delta, A, Dp, bias, ck, Wst, dtw = ctx.saved_tensors
        B, d, H, W, c, R, N = ctx.dims
        L = H * W
        perm = list(_PERM)
        g2 = _two_orders(dy.float().reshape(B, d, H, W))                       # CrossMerge^T: 2 planes, not 4
        dp4 = torch.empty_like(p4)
        Bv, Cv = p4[:, :, R:R + N], p4[:, :, R + N:]
        du, ddelta, dA, _, _, dD, dbias = _core.bwd_ext(
            xs2.view(B, 2 * d, L), delta.view(B, 4 * d, L), A, Bv, Cv, Dp, bias, g2.view(B, 2 * d, L), ck, True,
            rev_mask=_REV_MASK, u_gshift=1, dout_gshift=1, dB_out=dp4[:, :, R:R + N], dC_out=dp4[:, :, R + N:])
        ddelta4 = ddelta.view(B, 4, d, L)
        # dt_proj: delta = dtw @ p4[:R]
        dp4[:, :, :R] = torch.matmul(dtw.transpose(1, 2).unsqueeze(0), ddelta4)
        d_dtw = torch.matmul(ddelta4, p4[:, :, :R].transpose(-1, -2)).sum(0)   # (4, d, R)
        # x_proj: p = Wst @ xs2
        dp2 = dp4.view(B, 2, 2 * c, L)
        dxs2 = torch.matmul(Wst.transpose(1, 2).unsqueeze(0), dp2)             # (B, 2, d, L)
        du4 = du.view(B, 2, 2, d, L)
        dxs2 += du4[:, :, 0]
        dxs2 += du4[:, :, 1]
        dWst = torch.matmul(dp2, xs2.transpose(-1, -2)).sum(0)                 # (2, 2c, d)
        d_xproj = dWst.view(4, c, d)[perm]
        d_dtw = d_dtw[perm]
        dA_logs = (dA * A).view(4, d, N)[perm].reshape(4 * d, N)               # A = -exp(A_logs)
        dDs = dD.view(4, d)[perm].reshape(-1)
        dbias = dbias.view(4, d)[perm]
        return dxs2, None, None, d_xproj, d_dtw, dbias, dA_logs, dDs


class _TwoOrdersFn(torch.autograd.Function):
    """(B, d, H, W) -> (B, 2, d, L) and its adjoint, with plain copies (callers without the fused conv)."""

    @staticmethod
    def forward(ctx, x):
        ctx.hw = x.shape[2:]
        return _two_orders(x.float())

    @staticmethod
    def backward(ctx, g):
        H, W = ctx.hw
        B, _, d, L = g.shape
        return g[:, 0].reshape(B, d, H, W) + g[:, 1].reshape(B, d, W, H).transpose(2, 3)


def ss2d_core(x, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds):
    """x (B, d, H, W) -> (B, d, H*W)"""
    B, d, H, W = x.shape
    return SS2DCoreFn.apply(_TwoOrdersFn.apply(x), H, W, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds)


def ss2d_core_from_orders(xs2, H, W, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds):
    """xs2 (B, 2, d, H*W) as produced by dwconv_silu_two_orders -> (B, d, H*W)"""
    return SS2DCoreFn.apply(xs2, H, W, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds)
