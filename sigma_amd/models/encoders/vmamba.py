"""Mamba building blocks of Sigma on MI355X: SS2D, fusion SSMs, VSS / CVSS blocks, VSSM backbone.

Host-side mirror of the LIVE path of the reference file models/encoders/vmamba.py (the dead
variants listed in SURVEY.md App. C-10 are deliberately not rebuilt).  Module and parameter
names reproduce the reference's state_dict exactly so that checkpoints move
both ways; the computation is organised differently:

  * every selective scan runs on the hand-written gfx950 kernels through
    ``sigma_amd.selective_scan`` (C ABI of include/sigma_scan.h) -- there is no fallback;
  * the four scan directions of SS2D share ONE projection GEMM per memory order: x_proj and
    dt_proj act point-wise along the sequence, so they commute with the CrossScan permutation
    (SURVEY.md App. E.3).  We project the row-major and the column-major image once each with
    the weights of two directions stacked, and only flip the small (R+2N)-row result instead
    of projecting four permuted copies of the d-row activations
    (reference: vmamba.py:193-199 einsums over the materialised (B,4,D,L) tensor);
  * CrossScan / CrossMerge are expressed with differentiable view/flip/transpose ops, so no
    custom autograd functions are needed (reference: vmamba.py:80-121, 123-163).

Reference line numbers in the docstrings are for /root/reference/models/encoders/vmamba.py.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Optional

import torch
import torch.nn as nn

from ... import gemm as _gemm
from ...layernorm import LayerNorm
import torch.nn.functional as F

import os

from ...selective_scan import selective_scan_fn
from ...layout import channels_first, channels_last, transpose_rows
from ...pointwise import channel_gate, channel_gate_ok, scale_residual
from ...ss2d_fused import (dwconv_silu, dwconv_silu_two_orders, selective_scan_ext, split_xz, ss2d_core,
                           ss2d_core_from_orders)

# SIGMA_SS2D_FUSED=0 selects the plain-autograd formulation of SS2D's core (A/B measurements and
# the fused-vs-unfused parity test); both run the same HIP scan kernels.
_FUSED_SS2D = os.environ.get("SIGMA_SS2D_FUSED", "1") != "0"
_FUSED_GATE = os.environ.get("SIGMA_FUSED_GATE", "1") != "0"      # out_norm * silu(z) as one HIP pass
_FUSED_SPLIT = os.environ.get("SIGMA_FUSED_SPLIT", "1") != "0"    # chunk + permute as one tiled transpose
_FUSED_XZ = os.environ.get("SIGMA_FUSED_XZ", "1") != "0"          # ... or none: in_proj writes its x half channel-major (round 6)


# --------------------------------------------------------------------------- small helpers
class DropPath(nn.Module):
    """Stochastic depth per sample (timm.models.layers.DropPath semantics, scale_by_keep)."""

    def __init__(self, drop_prob: float = 0.0):
        super().__init__()
        self.drop_prob = float(drop_prob)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        if keep > 0.0:
            mask.div_(keep)
        return x * mask

    def add_to(self, residual: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
        """residual + drop_path(x) as ONE elementwise pass (addcmul with the per-sample mask) instead of a multiply and an
        add over the whole activation; same random draw, same values."""
        if self.drop_prob == 0.0 or not self.training:
            return residual + x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        if keep > 0.0:
            mask.div_(keep)
        return torch.addcmul(residual, x, mask)

    def draw(self, x: torch.Tensor):
        """the per-sample factor of this call (mask / keep_prob, shape (B, 1, ..., 1)) or None when nothing is dropped --
        for callers that fold it into the branch (SS2D applies it inside its gated LayerNorm pass)"""
        if self.drop_prob == 0.0 or not self.training:
            return None
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        if keep > 0.0:
            mask.div_(keep)
        return mask

    def extra_repr(self) -> str:
        return f"drop_prob={self.drop_prob:.3f}"


class Permute(nn.Module):
    def __init__(self, *dims: int):
        super().__init__()
        self.dims = dims

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.dims == (0, 2, 3, 1) and x.is_cuda and x.dim() == 4 and x.is_contiguous():
            # patch_embed: conv (B, C, H, W) -> LayerNorm over C.  The LayerNorm needs contiguous rows: make them with the
            # tiled transpose (and hand the conv a contiguous gradient) instead of a strided ATen copy each way
            return channels_last(x)
        return x.permute(*self.dims)


def _auto_nrows(rows: int) -> int:
    """nrows choice of the reference wrappers (vmamba.py:183-191); only a shape check here."""
    for r in (4, 3, 2):
        if rows % r == 0:
            return r
    return 1


def _dt_projection(dt_rank: int, d_inner: int, dt_scale=1.0, dt_init="random", dt_min=0.001, dt_max=0.1,
                   dt_init_floor=1e-4) -> nn.Linear:
    """dt projection initialised so that softplus(bias) is log-uniform in [dt_min, dt_max]
    (vmamba.py:728-753)."""
    proj = nn.Linear(dt_rank, d_inner, bias=True)
    std = dt_rank ** -0.5 * dt_scale
    if dt_init == "constant":
        nn.init.constant_(proj.weight, std)
    elif dt_init == "random":
        nn.init.uniform_(proj.weight, -std, std)
    else:
        raise NotImplementedError(dt_init)
    dt = torch.exp(torch.rand(d_inner) * (math.log(dt_max) - math.log(dt_min)) + math.log(dt_min)).clamp(min=dt_init_floor)
    inv_softplus = dt + torch.log(-torch.expm1(-dt))
    with torch.no_grad():
        proj.bias.copy_(inv_softplus)
    return proj


def _a_log(d_state: int, d_inner: int, copies: int = 0) -> nn.Parameter:
    """S4D-real init: A = -(1..N) for every row (vmamba.py:755-770)."""
    a = torch.arange(1, d_state + 1, dtype=torch.float32).repeat(d_inner, 1)
    a_log = torch.log(a)
    if copies > 0:
        a_log = a_log.repeat(copies, 1)
    p = nn.Parameter(a_log)
    p._no_weight_decay = True
    return p


def _d_skip(d_inner: int, copies: int = 0) -> nn.Parameter:
    """Skip parameter D = 1 (vmamba.py:772-782)."""
    p = nn.Parameter(torch.ones(d_inner * max(copies, 1)))
    p._no_weight_decay = True
    return p


def _stacked_ssm_params(mod: nn.Module, K: int, d_inner: int, d_state: int, dt_rank: int, dt_kwargs: dict) -> None:
    """x_proj_weight (K, R+2N, d), dt_projs_weight (K, d, R), dt_projs_bias (K, d), A_logs (K*d, N),
    Ds (K*d) -- raw Parameters exactly as in vmamba.py:697-721 / 1149-1168."""
    xw = [nn.Linear(d_inner, dt_rank + 2 * d_state, bias=False).weight for _ in range(K)]
    mod.x_proj_weight = nn.Parameter(torch.stack([w.detach() for w in xw], dim=0))
    dts = [_dt_projection(dt_rank, d_inner, **dt_kwargs) for _ in range(K)]
    mod.dt_projs_weight = nn.Parameter(torch.stack([t.weight.detach() for t in dts], dim=0))
    mod.dt_projs_bias = nn.Parameter(torch.stack([t.bias.detach() for t in dts], dim=0))
    mod.A_logs = _a_log(d_state, d_inner, copies=K)
    mod.Ds = _d_skip(d_inner, copies=K)


def _conv_act(conv: nn.Conv2d, act: nn.Module, x: torch.Tensor) -> torch.Tensor:
    """act(conv(x)) for the depthwise 3x3 + SiLU pairs of the Mamba blocks: one HIP pass on the GPU
    (sigma_amd/csrc/dwconv.hip); the torch modules otherwise (CPU host-logic tests)."""
    if _FUSED_SS2D and x.is_cuda and isinstance(act, nn.SiLU) and conv.kernel_size == (3, 3) and conv.groups == conv.in_channels:
        return dwconv_silu(x, conv.weight, conv.bias)
    return act(conv(x))


# --------------------------------------------------------------------------- SS2D core
def ss2d_scan(x: torch.Tensor, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds,
              out_norm: nn.Module) -> torch.Tensor:
    """Four-direction 2-D selective scan; restates cross_selective_scan (vmamba.py:165-226).

    x: (B, d, H, W) fp32-able activations.  Returns (B, H, W, d) after out_norm.
    Direction order and index maps: SURVEY.md App. E.3 / vmamba.py:80-89:
      k=0 row-major, k=1 column-major, k=2 reversed row-major, k=3 reversed column-major.
    """
    B, d, H, W = x.shape
    if _FUSED_SS2D and x.is_cuda:          # CPU tensors take the plain formulation, whose scan raises (no fallback)
        y = ss2d_core(x, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds)     # (B, H, W, d)
        return out_norm(y).to(x.dtype)
    K, c, _ = x_proj_weight.shape                # c = R + 2N
    R = dt_projs_weight.shape[2]
    N = A_logs.shape[1]
    L = H * W
    x_rm = x.reshape(B, d, L)                                    # row-major sequence
    x_cm = x.transpose(2, 3).reshape(B, d, L)                    # column-major sequence (one copy)
    # project each memory order once with the weights of its two directions stacked
    w_rm = torch.cat([x_proj_weight[0], x_proj_weight[2]], dim=0)   # (2c, d)
    w_cm = torch.cat([x_proj_weight[1], x_proj_weight[3]], dim=0)
    p_rm = torch.matmul(w_rm, x_rm)                              # (B, 2c, L)
    p_cm = torch.matmul(w_cm, x_cm)
    x_dbl = torch.stack([p_rm[:, :c], p_cm[:, :c], p_rm[:, c:].flip(-1), p_cm[:, c:].flip(-1)], dim=1)  # (B,4,c,L)
    dts, Bs, Cs = torch.split(x_dbl, [R, N, N], dim=2)
    dts = torch.matmul(dt_projs_weight.unsqueeze(0), dts)        # (1,4,d,R) @ (B,4,R,L) -> (B,4,d,L)
    xs = torch.stack([x_rm, x_cm, x_rm.flip(-1), x_cm.flip(-1)], dim=1)                                 # (B,4,d,L)

    u = xs.reshape(B, K * d, L).float()
    delta = dts.reshape(B, K * d, L).float()
    As = -torch.exp(A_logs.float())
    ys = selective_scan_fn(u, delta, As, Bs.float().contiguous(), Cs.float().contiguous(), Ds.float(),
                           dt_projs_bias.float().reshape(-1), True, _auto_nrows(K * d))
    ys = ys.view(B, K, d, L)
    # CrossMerge (vmamba.py:100-108): undo the flips, undo the transpose, add the 4 directions
    y_rm = ys[:, 0] + ys[:, 2].flip(-1)
    y_cm = ys[:, 1] + ys[:, 3].flip(-1)
    y = y_rm + y_cm.view(B, d, W, H).transpose(2, 3).reshape(B, d, L)
    y = y.transpose(1, 2).reshape(B, H, W, d)
    return out_norm(y).to(x.dtype)


class SS2D(nn.Module):
    """2-D selective scan block (vmamba.py:640-782, forward :1067-1089, core v2 only)."""

    def __init__(self, d_model=96, d_state=16, ssm_ratio=2, dt_rank="auto", d_conv=3, conv_bias=True, dropout=0.0,
                 bias=False, dt_min=0.001, dt_max=0.1, dt_init="random", dt_scale=1.0, dt_init_floor=1e-4, **_):
        super().__init__()
        self.d_model = d_model
        self.d_state = math.ceil(d_model / 6) if d_state == "auto" else d_state
        self.d_conv = d_conv
        self.expand = ssm_ratio
        self.d_inner = int(ssm_ratio * d_model)
        self.dt_rank = math.ceil(d_model / 16) if dt_rank == "auto" else dt_rank
        self.K = 4
        if d_conv <= 1:
            raise NotImplementedError("Sigma always uses the 3x3 depthwise conv (d_conv=3)")
        self.in_proj = nn.Linear(d_model, 2 * self.d_inner, bias=bias)
        self.conv2d = nn.Conv2d(self.d_inner, self.d_inner, kernel_size=d_conv, padding=(d_conv - 1) // 2,
                                groups=self.d_inner, bias=conv_bias)
        self.act = nn.SiLU()
        _stacked_ssm_params(self, self.K, self.d_inner, self.d_state, self.dt_rank,
                            dict(dt_scale=dt_scale, dt_init=dt_init, dt_min=dt_min, dt_max=dt_max,
                                 dt_init_floor=dt_init_floor))
        self.out_norm = LayerNorm(self.d_inner)
        self.out_proj = nn.Linear(self.d_inner, d_model, bias=bias)
        self.dropout = nn.Dropout(dropout) if dropout > 0.0 else nn.Identity()

    def forward(self, x: torch.Tensor, branch_scale=None, residual=None) -> torch.Tensor:       # x: (B, H, W, C)
        """``branch_scale``: optional per-sample factor (B, 1, 1, 1) on the result -- the block's stochastic-depth mask.
        out_proj is linear and bias-free, so the factor is applied to its input inside the gated LayerNorm pass.
        ``residual``: optional tensor of the output's shape that is added to the result -- the block's residual stream,
        added inside the out_proj GEMM (its accumulators start from it) instead of in a pass of its own."""
        fold = branch_scale is not None and self.out_proj.bias is None and isinstance(self.dropout, nn.Identity)
        if (_FUSED_SS2D and _FUSED_SPLIT and _FUSED_XZ and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
                and _gemm.gemm_mode() == "split3" and not self.in_proj._forward_hooks and not self.in_proj._forward_pre_hooks
                and _gemm.xz_ok(x.reshape(-1, x.shape[-1]), self.in_proj.weight)):
            # round 6: ONE GEMM writes the x half channel-major (d, B, H, W) for the convolution and the z half
            # channels-last for the gate -- no transposing pass between in_proj and the convolution, in either direction
            xi, z = _gemm.linear_xz(x, self.in_proj.weight, self.in_proj.bias)   # (B, d, H, W) view, (B, H, W, d)
        else:
            xz = self.in_proj(x)
            if _FUSED_SS2D and _FUSED_SPLIT and xz.is_cuda and xz.dtype == torch.float32:
                xi, z = split_xz(xz)                                         # (B, d, H, W), view (B, H, W, d)
            else:
                xi, z = xz.chunk(2, dim=-1)
                xi = xi.permute(0, 3, 1, 2).contiguous()                     # (B, d, H, W)
        if _FUSED_SS2D and xi.is_cuda:
            # depthwise conv + SiLU + both scan orders in one HIP pass, then the fused scan core
            Bq, dq, Hq, Wq = xi.shape
            xs2 = dwconv_silu_two_orders(xi, self.conv2d.weight, self.conv2d.bias)
            y = ss2d_core_from_orders(xs2, Hq, Wq, self.x_proj_weight, self.dt_projs_weight, self.dt_projs_bias,
                                      self.A_logs, self.Ds)
            if _FUSED_GATE:
                # out_norm(y) * silu(z) (* mask), one pass
                y = self.out_norm.forward_gated(y, z, branch_scale if fold else None).to(x.dtype)
            else:
                y = self.out_norm(y).to(x.dtype) * F.silu(z)
                fold = False
        else:
            y = ss2d_scan(self.act(self.conv2d(xi)), self.x_proj_weight, self.dt_projs_weight, self.dt_projs_bias,
                          self.A_logs, self.Ds, self.out_norm)
            y = y * F.silu(z)
            fold = False
        scaled_after = not (branch_scale is None or fold)
        if (residual is not None and not scaled_after and isinstance(self.dropout, nn.Identity) and y.is_cuda
                and y.dtype == torch.float32 and _gemm.gemm_mode() == "split3"):
            return _gemm.linear(y, self.out_proj.weight, self.out_proj.bias, residual=residual)
        out = self.dropout(self.out_proj(y))
        if scaled_after:
            out = out * branch_scale
        return out if residual is None else residual + out


class PatchMerging2D(nn.Module):
    """2x2 patch merging, 'v1' downsample (vmamba.py:612-636): pad odd sizes, gather, LN(4C), Linear(4C->2C)."""

    def __init__(self, dim, out_dim=-1, norm_layer=LayerNorm):
        super().__init__()
        self.dim = dim
        self.reduction = nn.Linear(4 * dim, (2 * dim) if out_dim < 0 else out_dim, bias=False)
        self.norm = norm_layer(4 * dim)

    def forward(self, x: torch.Tensor) -> torch.Tensor:       # (B, H, W, C)
        H, W = x.shape[-3], x.shape[-2]
        if (H % 2) or (W % 2):
            x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
        # the reference's cat of the four strided slices [(h even, w even), (h odd, w even), (h even, w odd), (h odd, w odd)]
        # (vmamba.py:627-631) is one permutation: channel block index = 2 * (w parity) + (h parity).  As a single
        # permute + copy its backward is one copy too, instead of 4 x (zero fill + strided copy) + 3 adds
        # (SliceBackward0: 3.1 ms per step in profiles/r02_aten_tail_by_node.txt).
        lead = x.shape[:-3]
        Hp, Wp, C = x.shape[-3], x.shape[-2], x.shape[-1]
        x = x.reshape(*lead, Hp // 2, 2, Wp // 2, 2, C).permute(*range(len(lead)), len(lead), len(lead) + 2, len(lead) + 3,
                                                                 len(lead) + 1, len(lead) + 4)
        x = x.reshape(*lead, Hp // 2, Wp // 2, 4 * C)
        return self.reduction(self.norm(x))


class VSSBlock(nn.Module):
    """x + DropPath(SS2D(LN(x))) -- Sigma uses mlp_ratio = 0 so there is no FFN branch (vmamba.py:1673-1722)."""

    def __init__(self, hidden_dim=0, drop_path=0.0, norm_layer=LayerNorm, attn_drop_rate=0.0, d_state=16,
                 dt_rank="auto", ssm_ratio=2.0, mlp_ratio=0.0, **kwargs):
        super().__init__()
        if mlp_ratio and mlp_ratio > 0:
            raise NotImplementedError("Sigma builds VSS blocks with mlp_ratio=0 (dual_vmamba.py:119,130,141)")
        self.norm = norm_layer(hidden_dim)
        self.op = SS2D(d_model=hidden_dim, dropout=attn_drop_rate, d_state=d_state, ssm_ratio=ssm_ratio, dt_rank=dt_rank)
        self.drop_path = DropPath(drop_path)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        # x + drop_path(op(norm(x))): the per-sample mask rides inside the branch (SS2D.forward), the residual stream is
        # added by the branch's last GEMM
        if isinstance(self.norm, LayerNorm):
            y, x = self.norm.forward_with_pass(x)       # the residual gradient joins the LayerNorm backward (one pass less)
        else:
            y = self.norm(x)
        return self.op(y, self.drop_path.draw(x), residual=x)


# --------------------------------------------------------------------------- decoder block
class ChannelAttention(nn.Module):
    """avg+max pooled squeeze/excite gate (vmamba.py:1725-1741)."""

    def __init__(self, num_feat, squeeze_factor=16):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.max_pool = nn.AdaptiveMaxPool2d(1)
        self.fc = nn.Sequential(
            nn.Conv2d(num_feat, num_feat // squeeze_factor, 1, bias=False),
            nn.SiLU(inplace=True),
            nn.Conv2d(num_feat // squeeze_factor, num_feat, 1, bias=False),
        )
        self.sigmoid = nn.Sigmoid()

    def forward(self, x):
        w1, w2 = self.fc[0].weight, self.fc[2].weight
        if x.is_cuda and channel_gate_ok(x, w1, w2) and isinstance(self.fc[1], nn.SiLU):
            return channel_gate(x, w1, w2)              # one pooling pass + one scaling pass (csrc/pointwise.hip)
        # global pools as plain reductions (AdaptiveMaxPool2d(1) runs a 0.9 ms one-thread-per-output
        # kernel on ROCm and keeps an index tensor for backward); same values and gradients
        pooled = torch.cat([x.mean(dim=(2, 3), keepdim=True), x.amax(dim=(2, 3), keepdim=True)], dim=0)
        g = self.fc(pooled)
        gate = g[:x.shape[0]] + g[x.shape[0]:]
        return x * self.sigmoid(gate)


class ChannelAttentionBlock(nn.Module):
    """conv3x3 (C -> C/3) - GELU - conv3x3 (C/3 -> C) - ChannelAttention(C, 30) (vmamba.py:1744-1757)."""

    def __init__(self, num_feat, compress_ratio=3, squeeze_factor=30):
        super().__init__()
        self.cab = nn.Sequential(
            nn.Conv2d(num_feat, num_feat // compress_ratio, 3, 1, 1),
            nn.GELU(),
            nn.Conv2d(num_feat // compress_ratio, num_feat, 3, 1, 1),
            ChannelAttention(num_feat, squeeze_factor),
        )

    def forward(self, x):
        return self.cab(x)


class CVSSDecoderBlock(nn.Module):
    """Channel-aware VSS block of the decoder (vmamba.py:1760-1811)."""

    def __init__(self, hidden_dim=0, drop_path=0.0, norm_layer=LayerNorm, attn_drop_rate=0.0, d_state=16,
                 dt_rank="auto", ssm_ratio=2.0, **kwargs):
        super().__init__()
        self.norm1 = norm_layer(hidden_dim)
        self.scale1 = nn.Parameter(torch.ones(hidden_dim))
        self.op = SS2D(d_model=hidden_dim, dropout=attn_drop_rate, d_state=d_state, ssm_ratio=ssm_ratio, dt_rank=dt_rank)
        self.drop_path = DropPath(drop_path)
        self.conv_blk = ChannelAttentionBlock(hidden_dim)
        self.norm2 = norm_layer(hidden_dim)
        self.scale2 = nn.Parameter(torch.ones(hidden_dim))

    def forward(self, x: torch.Tensor) -> torch.Tensor:       # (B, H, W, C)
        # x * scale1 + drop_path(op(norm1(x))): mask inside the branch, scale + add in one pass (branch first: the result
        # inherits its contiguous channels-last layout)
        # (LayerNorm(x), x): the gradient of the scaled residual joins the LayerNorm backward in its kernel (layernorm.py)
        n1, x = self.norm1.forward_with_pass(x) if isinstance(self.norm1, LayerNorm) else (self.norm1(x), x)
        x = scale_residual(self.op(n1, self.drop_path.draw(x)), x, self.scale1)
        n2, x = self.norm2.forward_with_pass(x) if isinstance(self.norm2, LayerNorm) else (self.norm2(x), x)
        y = self.conv_blk(channels_first(n2))
        # the channels-last operand first: the sum then comes out contiguous in (B, H, W, C) and the next block's
        # LayerNorm / in_proj read it in place (with the permuted conv output first, the result inherited its NCHW
        # strides and every following LayerNorm started with a transposing copy: 7 x 413 MB per step at 120 x 160)
        if y.is_cuda:
            # conv branch back to channels-last with the tiled transpose (its gradient then arrives contiguous in the
            # (B, C, H, W) order the gate's backward reads), residual scale + add in one pass
            return scale_residual(channels_last(y), x, self.scale2)
        return x * self.scale2 + y.permute(0, 2, 3, 1)


# --------------------------------------------------------------------------- CroMB
class Cross_Mamba_Attention_SSM(nn.Module):
    """Two 1-D scans whose C matrices are swapped across modalities (vmamba.py:1407-1545, E.5)."""

    def __init__(self, d_model=96, d_state=4, ssm_ratio=2, dt_rank="auto", dt_min=0.001, dt_max=0.1, dt_init="random",
                 dt_scale=1.0, dt_init_floor=1e-4, **_):
        super().__init__()
        self.d_model = d_model
        self.d_state = math.ceil(d_model / 6) if d_state == "auto" else d_state
        self.d_inner = int(ssm_ratio * d_model)
        self.dt_rank = math.ceil(d_model / 16) if dt_rank == "auto" else dt_rank
        kw = dict(dt_scale=dt_scale, dt_init=dt_init, dt_min=dt_min, dt_max=dt_max, dt_init_floor=dt_init_floor)
        self.x_proj_1 = nn.Linear(self.d_inner, self.dt_rank + 2 * self.d_state, bias=False)
        self.x_proj_2 = nn.Linear(self.d_inner, self.dt_rank + 2 * self.d_state, bias=False)
        self.dt_proj_1 = _dt_projection(self.dt_rank, self.d_inner, **kw)
        self.dt_proj_2 = _dt_projection(self.dt_rank, self.d_inner, **kw)
        self.A_log_1 = _a_log(self.d_state, self.d_inner)
        self.A_log_2 = _a_log(self.d_state, self.d_inner)
        self.D_1 = _d_skip(self.d_inner)
        self.D_2 = _d_skip(self.d_inner)
        self.out_norm_1 = LayerNorm(self.d_inner)
        self.out_norm_2 = LayerNorm(self.d_inner)

    def _project(self, x_seq, x_proj, dt_proj):
        """x_seq (B, d, L) -> delta (B, d, L), B (B, N, L), C (B, N, L); bias enters via delta_bias."""
        # bmm with the weights as a stride-0 batch: torch.matmul folds the batch of a (2-D | 1 x 2-D) @ 3-D product into the
        # rows of ONE GEMM and for that copies the activations into (B, L, d) order first (2 x 59 MB per call at
        # 120 x 160, and a transposed gradient to add in backward)
        Bsz = x_seq.shape[0]
        dbl = torch.bmm(x_proj.weight.unsqueeze(0).expand(Bsz, -1, -1), x_seq)       # (B, R+2N, L)
        dt, Bm, Cm = torch.split(dbl, [self.dt_rank, self.d_state, self.d_state], dim=1)
        # B / C: row slices of dbl, read in place by the scan
        return torch.bmm(dt_proj.weight.unsqueeze(0).expand(Bsz, -1, -1), dt), Bm, Cm

    def forward(self, x_rgb: torch.Tensor, x_e: torch.Tensor):   # both (B, d, L) channel-major sequences
        dt_rgb, B_rgb, C_rgb = self._project(x_rgb, self.x_proj_1, self.dt_proj_1)
        dt_e, B_e, C_e = self._project(x_e, self.x_proj_2, self.dt_proj_2)
        y_rgb = selective_scan_fn(x_rgb, dt_rgb, -torch.exp(self.A_log_1.float()), B_rgb, C_e, self.D_1.float(),
                                  self.dt_proj_1.bias.float(), True)
        y_e = selective_scan_fn(x_e, dt_e, -torch.exp(self.A_log_2.float()), B_e, C_rgb, self.D_2.float(),
                                self.dt_proj_2.bias.float(), True)
        return self.out_norm_1(transpose_rows(y_rgb)), self.out_norm_2(transpose_rows(y_e))   # (B, L, d)


class CrossMambaFusion_SS2D_SSM(nn.Module):
    """CroMB operator (vmamba.py:1549-1640): per-modality in_proj, SHARED dwconv+SiLU, cross SSM, out_proj."""

    def __init__(self, d_model=96, d_state=16, ssm_ratio=2, dt_rank="auto", d_conv=3, conv_bias=True, dropout=0.0,
                 bias=False, **kwargs):
        super().__init__()
        self.d_model = d_model
        self.d_inner = int(ssm_ratio * d_model)
        self.in_proj = nn.Linear(d_model, self.d_inner, bias=bias)
        self.in_proj_modalx = nn.Linear(d_model, self.d_inner, bias=bias)
        self.conv2d = nn.Conv2d(self.d_inner, self.d_inner, kernel_size=d_conv, padding=(d_conv - 1) // 2,
                                groups=self.d_inner, bias=conv_bias)
        self.act = nn.SiLU()
        self.out_proj_rgb = nn.Linear(self.d_inner, d_model, bias=bias)
        self.out_proj_e = nn.Linear(self.d_inner, d_model, bias=bias)
        self.dropout_rgb = nn.Dropout(dropout) if dropout > 0.0 else nn.Identity()
        self.dropout_e = nn.Dropout(dropout) if dropout > 0.0 else nn.Identity()
        self.CMA_ssm = Cross_Mamba_Attention_SSM(d_model=d_model, d_state=d_state, ssm_ratio=ssm_ratio, dt_rank=dt_rank)

    def forward(self, x_rgb: torch.Tensor, x_e: torch.Tensor):   # (B, H, W, C) each
        B, H, W, _ = x_rgb.shape
        both = torch.cat([channels_first(self.in_proj(x_rgb)), channels_first(self.in_proj_modalx(x_e))], dim=0)
        both = _conv_act(self.conv2d, self.act, both).flatten(2)      # shared conv: one launch; (2B, d, L)
        b_rgb, b_e = both.split(B, dim=0)             # split: the backward is ONE cat (two slices: 2 x zero fill + copy + add)
        y_rgb, y_e = self.CMA_ssm(b_rgb, b_e)
        y_rgb = self.dropout_rgb(self.out_proj_rgb(y_rgb.view(B, H, W, -1)))
        y_e = self.dropout_e(self.out_proj_e(y_e.view(B, H, W, -1)))
        return y_rgb, y_e


class CrossMambaFusionBlock(nn.Module):
    """CroMB block: residual around the cross SSM for both modalities (vmamba.py:1814-1870)."""

    def __init__(self, hidden_dim=0, drop_path=0.0, attn_drop_rate=0.0, d_state=4, dt_rank="auto", ssm_ratio=2.0,
                 mlp_ratio=0.0, **kwargs):
        super().__init__()
        self.op = CrossMambaFusion_SS2D_SSM(d_model=hidden_dim, dropout=attn_drop_rate, d_state=d_state,
                                            ssm_ratio=ssm_ratio, dt_rank=dt_rank)
        self.drop_path1 = DropPath(drop_path)
        self.drop_path2 = DropPath(drop_path)

    def forward(self, x_rgb, x_e):
        c_rgb, c_e = self.op(x_rgb, x_e)
        return self.drop_path1.add_to(x_rgb, c_rgb), self.drop_path2.add_to(x_e, c_e)


# --------------------------------------------------------------------------- ConMB
class ConMB_SS2D(nn.Module):
    """Concat-Mamba operator (vmamba.py:1092-1284): scan over [rgb tokens ; x tokens] fwd + reversed,
    SE-style cross gating, channel concat, out_proj (App. E.4)."""

    def __init__(self, d_model=96, d_state=4, ssm_ratio=2, dt_rank="auto", d_conv=3, conv_bias=True, dropout=0.0,
                 bias=False, dt_min=0.001, dt_max=0.1, dt_init="random", dt_scale=1.0, dt_init_floor=1e-4, **_):
        super().__init__()
        self.d_model = d_model
        self.d_state = math.ceil(d_model / 6) if d_state == "auto" else d_state
        self.d_inner = int(ssm_ratio * d_model)
        self.dt_rank = math.ceil(d_model / 16) if dt_rank == "auto" else dt_rank
        self.K = 2
        self.in_proj = nn.Linear(d_model, self.d_inner, bias=bias)
        self.in_proj_modalx = nn.Linear(d_model, self.d_inner, bias=bias)
        conv = dict(kernel_size=d_conv, padding=(d_conv - 1) // 2, groups=self.d_inner, bias=conv_bias)
        self.conv2d = nn.Conv2d(self.d_inner, self.d_inner, **conv)
        self.conv2d_modalx = nn.Conv2d(self.d_inner, self.d_inner, **conv)
        self.act = nn.SiLU()
        _stacked_ssm_params(self, self.K, self.d_inner, self.d_state, self.dt_rank,
                            dict(dt_scale=dt_scale, dt_init=dt_init, dt_min=dt_min, dt_max=dt_max,
                                 dt_init_floor=dt_init_floor))
        self.out_norm1 = LayerNorm(self.d_inner)
        self.out_norm2 = LayerNorm(self.d_inner)
        self.out_proj = nn.Linear(2 * self.d_inner, d_model, bias=bias)
        self.dropout = nn.Dropout(dropout) if dropout > 0.0 else nn.Identity()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)

        def gate():
            return nn.Sequential(nn.Linear(self.d_inner, self.d_inner // 16, bias=False), nn.SiLU(inplace=True),
                                 nn.Linear(self.d_inner // 16, self.d_inner, bias=False), nn.Sigmoid())
        self.fc1 = gate()
        self.fc2 = gate()

    def _scan(self, c_rgb: torch.Tensor, c_e: torch.Tensor):
        """cross_selective_scan_multimodal_k2 (vmamba.py:369-430)."""
        B, d, H, W = c_rgb.shape
        HW = H * W
        L = 2 * HW
        R, N = self.dt_rank, self.d_state
        c = R + 2 * N
        seq = torch.cat([c_rgb.flatten(2), c_e.flatten(2)], dim=2)               # (B, d, 2HW): rgb tokens first
        # both directions in one batched GEMM (bmm, weights as a stride-0 batch: see Cross_Mamba_Attention_SSM._project)
        p = torch.bmm(self.x_proj_weight.reshape(1, 2 * c, d).expand(B, -1, -1), seq)
        if _FUSED_SS2D and seq.is_cuda:
            # the flipped direction is read backwards by the kernel: no flipped copies of seq / x_dbl / ys
            p4 = p.view(B, 2, c, L)
            p_dt, p_b, p_c = torch.split(p4, [R, N, N], dim=2)                   # views; backward = one cat
            dts = torch.matmul(self.dt_projs_weight.unsqueeze(0), p_dt)          # (B, 2, d, L)
            y = selective_scan_ext(seq, dts.reshape(B, 2 * d, L), -torch.exp(self.A_logs.float()), p_b,
                                   p_c, self.Ds.float(), self.dt_projs_bias.float().reshape(-1),
                                   rev_mask=0b10, u_gshift=1, pair_sum=True)     # (B, d, L): forward + reversed direction
            y1, y2 = y.split(HW, dim=-1)
            y_rgb = self.out_norm1(transpose_rows(y1).view(B, H, W, d))
            y_e = self.out_norm2(transpose_rows(y2).view(B, H, W, d))
            return y_rgb, y_e
        x_dbl = torch.stack([p[:, :c], p[:, c:].flip(-1)], dim=1)                # (B, 2, c, L)
        dts, Bs, Cs = torch.split(x_dbl, [R, N, N], dim=2)
        dts = torch.matmul(self.dt_projs_weight.unsqueeze(0), dts)               # (B, 2, d, L)
        xs = torch.stack([seq, seq.flip(-1)], dim=1)
        ys = selective_scan_fn(xs.reshape(B, 2 * d, L).float(), dts.reshape(B, 2 * d, L).float(),
                               -torch.exp(self.A_logs.float()), Bs.float().contiguous(), Cs.float().contiguous(),
                               self.Ds.float(), self.dt_projs_bias.float().reshape(-1), True, _auto_nrows(2 * d))
        ys = ys.view(B, 2, d, L)
        y = ys[:, 0] + ys[:, 1].flip(-1)                                         # CrossMerge_multimodal (:151-157)
        y_rgb = self.out_norm1(y[..., :HW].transpose(1, 2).reshape(B, H, W, d))
        y_e = self.out_norm2(y[..., HW:].transpose(1, 2).reshape(B, H, W, d))
        return y_rgb, y_e

    def forward(self, x_rgb: torch.Tensor, x_e: torch.Tensor) -> torch.Tensor:   # (B, H, W, C) each
        p_rgb = channels_first(self.in_proj(x_rgb))                              # (B, d, H, W)
        p_e = channels_first(self.in_proj_modalx(x_e))
        y_rgb, y_e = self._scan(_conv_act(self.conv2d, self.act, p_rgb), _conv_act(self.conv2d_modalx, self.act, p_e))
        # squeeze/excite: each modality is gated by the OTHER modality's pooled in_proj output (:1271-1281)
        g_rgb = self.fc1(p_rgb.mean(dim=(2, 3)))                                 # (B, d)
        g_e = self.fc2(p_e.mean(dim=(2, 3)))
        y = torch.cat([y_rgb * g_e[:, None, None, :], y_e * g_rgb[:, None, None, :]], dim=-1)
        return self.dropout(self.out_proj(y))


class ConcatMambaFusionBlock(nn.Module):
    """ConMB block: x_rgb + x_e + DropPath(ConMB_SS2D(x_rgb, x_e)) (vmamba.py:1873-1928)."""

    def __init__(self, hidden_dim=0, drop_path=0.0, attn_drop_rate=0.0, d_state=4, dt_rank="auto", ssm_ratio=2.0,
                 mlp_ratio=0.0, **kwargs):
        super().__init__()
        self.op = ConMB_SS2D(d_model=hidden_dim, dropout=attn_drop_rate, d_state=d_state, ssm_ratio=ssm_ratio,
                             dt_rank=dt_rank)
        self.drop_path = DropPath(drop_path)

    def forward(self, x_rgb, x_e):
        return self.drop_path.add_to(x_rgb + x_e, self.op(x_rgb, x_e))


# --------------------------------------------------------------------------- backbone
class Backbone_VSSM(nn.Module):
    """VMamba feature pyramid (VSSM :1931-2077 + Backbone_VSSM :2151-2212), v1 downsample.

    forward(x) -> list of 4 maps (B, C_i, H_i, W_i) after per-stage outnorm.
    """

    def __init__(self, patch_size=4, in_chans=3, num_classes=1000, depths=(2, 2, 9, 2), dims=(96, 192, 384, 768),
                 d_state=16, dt_rank="auto", ssm_ratio=2.0, attn_drop_rate=0.0, drop_rate=0.0, drop_path_rate=0.1,
                 mlp_ratio=0.0, patch_norm=True, norm_layer=LayerNorm, downsample_version="v1",
                 use_checkpoint=False, out_indices=(0, 1, 2, 3), pretrained=None, **kwargs):
        super().__init__()
        if downsample_version != "v1":
            raise NotImplementedError("Sigma selects downsample_version='v1' (dual_vmamba.py:120)")
        depths = list(depths)
        if isinstance(dims, int):
            dims = [int(dims * 2 ** i) for i in range(len(depths))]
        self.dims = list(dims)
        self.num_layers = len(depths)
        self.embed_dim = self.dims[0]
        self.num_features = self.dims[-1]
        self.patch_embed = nn.Sequential(
            nn.Conv2d(in_chans, self.embed_dim, kernel_size=patch_size, stride=patch_size, bias=True),
            Permute(0, 2, 3, 1),
            norm_layer(self.embed_dim) if patch_norm else nn.Identity(),
        )
        dpr = [r.item() for r in torch.linspace(0, drop_path_rate, sum(depths))]
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            blocks = [VSSBlock(hidden_dim=self.dims[i], drop_path=dpr[sum(depths[:i]) + j], norm_layer=norm_layer,
                               attn_drop_rate=attn_drop_rate, d_state=d_state, dt_rank=dt_rank, ssm_ratio=ssm_ratio,
                               mlp_ratio=mlp_ratio) for j in range(depths[i])]
            down = (PatchMerging2D(self.dims[i], self.dims[i + 1], norm_layer=norm_layer)
                    if i < self.num_layers - 1 else nn.Identity())
            self.layers.append(nn.Sequential(OrderedDict(blocks=nn.Sequential(*blocks), downsample=down)))
        self.apply(self._init_weights)                     # before the outnorms exist, as in the reference
        self.out_indices = tuple(out_indices)
        for i in self.out_indices:
            self.add_module(f"outnorm{i}", norm_layer(self.dims[i]))
        self.load_pretrained(pretrained)

    @staticmethod
    def _init_weights(m: nn.Module):
        """VSSM._init_weights (vmamba.py:2016-2023): Linear ~ trunc_normal(0.02), LayerNorm = (1, 0)."""
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02, a=-2.0, b=2.0)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def load_pretrained(self, ckpt: Optional[str] = None, key: str = "model"):
        """Best-effort ImageNet VMamba init; failures are reported and ignored exactly like the
        reference does (vmamba.py:2180-2191) -- the shipped .pth files are git-LFS stubs."""
        if ckpt is None:
            return
        try:
            blob = torch.load(ckpt, map_location="cpu")
            print(f"Successfully load ckpt {ckpt}")
            print("incompatible:", self.load_state_dict(blob[key], strict=False))
        except Exception as e:                               # noqa: BLE001 - mirrors the reference
            print(f"Failed loading checkpoint form {ckpt}: {e}")

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        """Accept checkpoints of the original VMamba code base (key renames of vmamba.py:2110-2147)."""
        def rename(src, dst):
            for k in [k for k in state_dict if k.startswith(prefix + src)]:
                state_dict[prefix + dst + k[len(prefix + src):]] = state_dict.pop(k)
        rename("patch_embed.proj", "patch_embed.0")
        rename("patch_embed.norm", "patch_embed.2")
        for i, layer in enumerate(self.layers):
            for j in range(len(layer.blocks)):
                rename(f"layers.{i}.blocks.{j}.ln_1", f"layers.{i}.blocks.{j}.norm")
                rename(f"layers.{i}.blocks.{j}.self_attention", f"layers.{i}.blocks.{j}.op")
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def _patch_embed(self, x: torch.Tensor) -> torch.Tensor:
        """patch_embed = Conv2d(kernel = stride = patch) -> (B, H, W, C) -> LayerNorm (vmamba.py:1965-1969).  A convolution
        whose windows do not overlap IS a matrix product of the flattened patches: on the GPU it runs on the
        split-operand bf16 MFMA kernels like every other projection (csrc/gemm_split.hip: forward nt, weight gradient tn;
        the images need no gradient), its result is channels-last as the LayerNorm wants it (no transposing copy), and
        the vendor convolution library is out of the stem.  SIGMA_GEMM=fp32 / CPU tensors keep nn.Conv2d."""
        conv, norm = self.patch_embed[0], self.patch_embed[2]
        p = conv.kernel_size
        # (a channels_last / cropped / flipped view keeps the layout-agnostic nn.Conv2d path, as do modules with forward hooks
        # on the stem, which this shortcut would skip -- ADVICE r5)
        hooked = bool(self.patch_embed._forward_hooks or self.patch_embed._forward_pre_hooks or conv._forward_hooks
                      or conv._forward_pre_hooks or norm._forward_hooks or norm._forward_pre_hooks)
        if (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous() and not hooked
                and isinstance(conv, nn.Conv2d) and _gemm.gemm_mode() == "split3"
                and conv.stride == p and conv.padding == (0, 0) and conv.dilation == (1, 1) and conv.groups == 1
                and x.shape[2] % p[0] == 0 and x.shape[3] % p[1] == 0 and (conv.in_channels * p[0] * p[1]) % 4 == 0
                and conv.weight.dtype == torch.float32):
            B, C, H, W = x.shape
            hp, wp = H // p[0], W // p[1]
            # one gather: row (b, i, j) = the patch's (c, ky, kx) values in the order of the weight's trailing dimensions
            cols = x.view(B, C, hp, p[0], wp, p[1]).permute(0, 2, 4, 1, 3, 5).reshape(B * hp * wp, C * p[0] * p[1])
            y = _gemm.linear(cols, conv.weight.view(conv.out_channels, -1), conv.bias)
            return norm(y.view(B, hp, wp, conv.out_channels))
        return self.patch_embed(x)

    def forward(self, x: torch.Tensor):
        x = self._patch_embed(x)
        outs = []
        for i, layer in enumerate(self.layers):
            o = layer.blocks(x)
            x = layer.downsample(o)
            if i in self.out_indices:
                # (B, C, H, W) as the reference returns it (vmamba.py:2200-2206), but as a channels-last VIEW: the fusion blocks
                # permute straight back to (B, H, W, C), so the reference's contiguous() copy would only be undone again
                outs.append(getattr(self, f"outnorm{i}")(o).permute(0, 3, 1, 2))
        return outs if self.out_indices else x
