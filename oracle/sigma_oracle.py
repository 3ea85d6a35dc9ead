"""CPU restatement of Sigma's model forward (EncoderDecoder logits) -- TEST INFRASTRUCTURE ONLY.

A plain function of (state_dict, rgb, modal_x): no nn.Module, no sigma_amd import, no GPU.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may use it.
It deliberately follows the REFERENCE's structure step by step (materialised CrossScan, one
einsum per direction set, two sequential backbone passes), not sigma_amd's re-organisation,
so that it can catch mistakes in the latter.  Selective scans go through oracle/scan_oracle.c.

Pinned by tests/test_oracle_model.py against tests/golden/model_*.npz, which were produced by
the reference's own Python model (tests/golden/make_golden_model.py).

Reference paths are relative to /root/reference.
"""
from __future__ import annotations

import math
from typing import Dict, List

import torch
import torch.nn.functional as F

from . import scan_oracle

SD = Dict[str, torch.Tensor]

PRESETS = {  # models/encoders/dual_vmamba.py:113-144
    "sigma_tiny": dict(depths=[2, 2, 9, 2], dims=96),
    "sigma_small": dict(depths=[2, 2, 27, 2], dims=96),
    "sigma_base": dict(depths=[2, 2, 27, 2], dims=128),
}


def _ln(x, sd: SD, prefix: str):
    w = sd[prefix + ".weight"]
    return F.layer_norm(x, (w.numel(),), w, sd[prefix + ".bias"], 1e-5)


class _ScanWithGrad(torch.autograd.Function):
    """the C oracle's forward and backward recurrences (oracle/scan_oracle.c) as one autograd node, for sigma_gradients"""

    @staticmethod
    def forward(ctx, u, delta, A, B, C, D, bias):
        ctx.save_for_backward(u, delta, A, B, C, D, bias)
        return scan_oracle.selective_scan_oracle(u, delta, A, B, C, D, bias, True, acc64=False)

    @staticmethod
    def backward(ctx, dout):
        u, delta, A, B, C, D, bias = ctx.saved_tensors
        du, dd, dA, dB, dC, dD, db = scan_oracle.selective_scan_oracle_bwd(u, delta, A, B, C, D, bias, dout.contiguous(), True)
        return du, dd, dA, dB, dC, dD, db


def _scan(u, delta, A, B, C, D, bias):
    if torch.is_grad_enabled() and any(t.requires_grad for t in (u, delta, A, B, C, D, bias)):
        return _ScanWithGrad.apply(u.contiguous(), delta.contiguous(), A.contiguous(), B.contiguous(), C.contiguous(), D, bias)
    return scan_oracle.selective_scan_oracle(u, delta, A, B, C, D, bias, True, acc64=False)


# ---- CrossScan / CrossMerge, models/encoders/vmamba.py:80-108
def cross_scan(x):                      # (B, C, H, W) -> (B, 4, C, L)
    B, C, H, W = x.shape
    xs = x.new_empty((B, 4, C, H * W))
    xs[:, 0] = x.flatten(2, 3)
    xs[:, 1] = x.transpose(2, 3).flatten(2, 3)
    xs[:, 2:4] = torch.flip(xs[:, 0:2], dims=[-1])
    return xs


def cross_merge(ys, H, W):              # (B, 4, D, L) -> (B, D, L)
    B, K, D, L = ys.shape
    ys = ys[:, 0:2] + ys[:, 2:4].flip(dims=[-1]).view(B, 2, D, L)
    return ys[:, 0] + ys[:, 1].view(B, D, W, H).transpose(2, 3).contiguous().view(B, D, L)


# ---- cross_selective_scan, vmamba.py:165-226
def cross_selective_scan(x, sd: SD, p: str):
    B, D, H, W = x.shape
    L = H * W
    xw, dtw, dtb = sd[p + ".x_proj_weight"], sd[p + ".dt_projs_weight"], sd[p + ".dt_projs_bias"]
    A_logs, Ds = sd[p + ".A_logs"], sd[p + ".Ds"]
    K, _, R = dtw.shape
    N = A_logs.shape[1]
    xs = cross_scan(x)
    x_dbl = torch.einsum("bkdl,kcd->bkcl", xs, xw)
    dts, Bs, Cs = torch.split(x_dbl, [R, N, N], dim=2)
    dts = torch.einsum("bkrl,kdr->bkdl", dts, dtw)
    ys = _scan(xs.reshape(B, -1, L).float(), dts.contiguous().view(B, -1, L).float(), -torch.exp(A_logs.float()),
               Bs.contiguous().float(), Cs.contiguous().float(), Ds.float(), dtb.reshape(-1).float())
    y = cross_merge(ys.view(B, K, -1, L), H, W)
    y = y.transpose(1, 2).contiguous().view(B, H, W, -1)
    return _ln(y, sd, p + ".out_norm")


# ---- SS2D.forward, vmamba.py:1067-1089
def ss2d(x, sd: SD, p: str):
    xz = F.linear(x, sd[p + ".in_proj.weight"])
    xi, z = xz.chunk(2, dim=-1)
    xi = xi.permute(0, 3, 1, 2).contiguous()
    d = xi.shape[1]
    xi = F.silu(F.conv2d(xi, sd[p + ".conv2d.weight"], sd[p + ".conv2d.bias"], padding=1, groups=d))
    y = cross_selective_scan(xi, sd, p)
    return F.linear(y * F.silu(z), sd[p + ".out_proj.weight"])


# ---- PatchMerging2D, vmamba.py:612-636
def patch_merging(x, sd: SD, p: str):
    H, W = x.shape[1], x.shape[2]
    if (W % 2) or (H % 2):
        x = F.pad(x, (0, 0, 0, W % 2, 0, H % 2))
    x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
    return F.linear(_ln(x, sd, p + ".norm"), sd[p + ".reduction.weight"])


# ---- Backbone_VSSM.forward, vmamba.py:2193-2212 (eval: DropPath = identity)
def backbone_vssm(img, sd: SD, p: str, depths: List[int]):
    x = F.conv2d(img, sd[p + ".patch_embed.0.weight"], sd[p + ".patch_embed.0.bias"], stride=4)
    x = _ln(x.permute(0, 2, 3, 1), sd, p + ".patch_embed.2")
    outs = []
    for i, depth in enumerate(depths):
        for j in range(depth):
            bp = f"{p}.layers.{i}.blocks.{j}"
            x = x + ss2d(_ln(x, sd, bp + ".norm"), sd, bp + ".op")            # VSSBlock, vmamba.py:1712-1716
        outs.append(_ln(x, sd, f"{p}.outnorm{i}").permute(0, 3, 1, 2).contiguous())
        if i < len(depths) - 1:
            x = patch_merging(x, sd, f"{p}.layers.{i}.downsample")
    return outs


# ---- CroMB: CrossMambaFusionBlock -> CrossMambaFusion_SS2D_SSM -> Cross_Mamba_Attention_SSM
#      vmamba.py:1857-1860, 1622-1640, 1508-1545
def cromb(x_rgb, x_e, sd: SD, p: str):
    o = p + ".op"
    B, H, W, _ = x_rgb.shape
    a = F.linear(x_rgb, sd[o + ".in_proj.weight"]).permute(0, 3, 1, 2).contiguous()
    b = F.linear(x_e, sd[o + ".in_proj_modalx.weight"]).permute(0, 3, 1, 2).contiguous()
    d = a.shape[1]
    conv = lambda t: F.silu(F.conv2d(t, sd[o + ".conv2d.weight"], sd[o + ".conv2d.bias"], padding=1, groups=d))
    a, b = conv(a).flatten(2), conv(b).flatten(2)                              # (B, d, L); the conv is SHARED
    c = o + ".CMA_ssm"
    R = sd[c + ".dt_proj_1.weight"].shape[1]
    N = sd[c + ".A_log_1"].shape[1]

    def proj(xs, i):
        dbl = torch.einsum("cd,bdl->bcl", sd[f"{c}.x_proj_{i}.weight"], xs)
        dt, Bm, Cm = torch.split(dbl, [R, N, N], dim=1)
        return torch.einsum("dr,brl->bdl", sd[f"{c}.dt_proj_{i}.weight"], dt), Bm.contiguous(), Cm.contiguous()

    dt_a, B_a, C_a = proj(a, 1)
    dt_b, B_b, C_b = proj(b, 2)
    y_a = _scan(a, dt_a, -torch.exp(sd[c + ".A_log_1"].float()), B_a, C_b, sd[c + ".D_1"].float(),
                sd[c + ".dt_proj_1.bias"].float())                             # C swapped, vmamba.py:1528-1539
    y_b = _scan(b, dt_b, -torch.exp(sd[c + ".A_log_2"].float()), B_b, C_a, sd[c + ".D_2"].float(),
                sd[c + ".dt_proj_2.bias"].float())
    y_a = _ln(y_a.transpose(1, 2), sd, c + ".out_norm_1").view(B, H, W, -1)
    y_b = _ln(y_b.transpose(1, 2), sd, c + ".out_norm_2").view(B, H, W, -1)
    return (x_rgb + F.linear(y_a, sd[o + ".out_proj_rgb.weight"]),
            x_e + F.linear(y_b, sd[o + ".out_proj_e.weight"]))


# ---- ConMB: ConcatMambaFusionBlock -> ConMB_SS2D -> cross_selective_scan_multimodal_k2
#      vmamba.py:1915-1916, 1265-1284, 369-430, 123-163
def conmb(x_rgb, x_e, sd: SD, p: str):
    o = p + ".op"
    B, H, W, _ = x_rgb.shape
    HW = H * W
    pa = F.linear(x_rgb, sd[o + ".in_proj.weight"]).permute(0, 3, 1, 2).contiguous()
    pb = F.linear(x_e, sd[o + ".in_proj_modalx.weight"]).permute(0, 3, 1, 2).contiguous()
    d = pa.shape[1]
    ca = F.silu(F.conv2d(pa, sd[o + ".conv2d.weight"], sd[o + ".conv2d.bias"], padding=1, groups=d))
    cb = F.silu(F.conv2d(pb, sd[o + ".conv2d_modalx.weight"], sd[o + ".conv2d_modalx.bias"], padding=1, groups=d))
    xw, dtw, dtb = sd[o + ".x_proj_weight"], sd[o + ".dt_projs_weight"], sd[o + ".dt_projs_bias"]
    R, N = dtw.shape[2], sd[o + ".A_logs"].shape[1]
    L = 2 * HW
    seq = torch.cat([ca.flatten(2), cb.flatten(2)], dim=2)
    xs = torch.stack([seq, seq.flip(-1)], dim=1)                               # CrossScan_multimodal
    x_dbl = torch.einsum("bkdl,kcd->bkcl", xs, xw)
    dts, Bs, Cs = torch.split(x_dbl, [R, N, N], dim=2)
    dts = torch.einsum("bkrl,kdr->bkdl", dts, dtw)
    ys = _scan(xs.reshape(B, -1, L), dts.contiguous().view(B, -1, L), -torch.exp(sd[o + ".A_logs"].float()),
               Bs.contiguous(), Cs.contiguous(), sd[o + ".Ds"].float(), dtb.reshape(-1).float()).view(B, 2, d, L)
    y = ys[:, 0] + ys[:, 1].flip(-1)                                           # CrossMerge_multimodal
    y_a = _ln(y[..., :HW].transpose(1, 2).contiguous().view(B, H, W, d), sd, o + ".out_norm1")
    y_b = _ln(y[..., HW:].transpose(1, 2).contiguous().view(B, H, W, d), sd, o + ".out_norm2")

    def gate(t, fc):
        s = t.mean(dim=(2, 3))
        return torch.sigmoid(F.linear(F.silu(F.linear(s, sd[f"{o}.{fc}.0.weight"])), sd[f"{o}.{fc}.2.weight"]))

    g_a, g_b = gate(pa, "fc1"), gate(pb, "fc2")
    y = torch.cat([y_a * g_b[:, None, None, :], y_b * g_a[:, None, None, :]], dim=-1)
    return x_rgb + x_e + F.linear(y, sd[o + ".out_proj.weight"])


# ---- RGBXTransformer.forward_features, dual_vmamba.py:78-107
def encoder(rgb, modal_x, sd: SD, depths: List[int]):
    f_rgb = backbone_vssm(rgb, sd, "backbone.vssm", depths)                   # two sequential passes
    f_x = backbone_vssm(modal_x, sd, "backbone.vssm", depths)
    fused = []
    for i in range(4):
        a = f_rgb[i].permute(0, 2, 3, 1).contiguous()
        b = f_x[i].permute(0, 2, 3, 1).contiguous()
        a, b = cromb(a, b, sd, f"backbone.cross_mamba.{i}")
        fused.append(conmb(a, b, sd, f"backbone.channel_attn_mamba.{i}").permute(0, 3, 1, 2).contiguous())
    return fused


# ---- CVSSDecoderBlock + ChannelAttentionBlock, vmamba.py:1800-1805, 1725-1757
def cvss_block(x, sd: SD, p: str):
    x = x * sd[p + ".scale1"] + ss2d(_ln(x, sd, p + ".norm1"), sd, p + ".op")
    t = _ln(x, sd, p + ".norm2").permute(0, 3, 1, 2).contiguous()
    t = F.conv2d(t, sd[p + ".conv_blk.cab.0.weight"], sd[p + ".conv_blk.cab.0.bias"], padding=1)
    t = F.conv2d(F.gelu(t), sd[p + ".conv_blk.cab.2.weight"], sd[p + ".conv_blk.cab.2.bias"], padding=1)
    fc = lambda v: F.conv2d(F.silu(F.conv2d(v, sd[p + ".conv_blk.cab.3.fc.0.weight"])), sd[p + ".conv_blk.cab.3.fc.2.weight"])
    att = fc(F.adaptive_avg_pool2d(t, 1)) + fc(F.adaptive_max_pool2d(t, 1))
    t = t * torch.sigmoid(att)
    return (t + (x * sd[p + ".scale2"]).permute(0, 3, 1, 2)).permute(0, 2, 3, 1).contiguous()


def _up2(x):
    return F.interpolate(x.permute(0, 3, 1, 2).contiguous(), scale_factor=2, mode="bilinear",
                         align_corners=False).permute(0, 2, 3, 1).contiguous()


# ---- MambaDecoder.forward, MambaDecoder.py:222-280
def decoder(feats, sd: SD):
    p = "decode_head"
    x = feats[3].permute(0, 2, 3, 1).contiguous()
    x = F.linear(x, sd[p + ".layers_up.0.expand.weight"])                     # PatchExpand, :20-30
    B, H, W, C = x.shape
    x = x.view(B, H, W, 2, 2, C // 4).permute(0, 1, 3, 2, 4, 5).reshape(B, 2 * H, 2 * W, C // 4)
    y = _ln(x, sd, p + ".layers_up.0.norm")
    for i in (1, 2, 3):
        skip = feats[3 - i]
        Hs, Ws = skip.shape[2], skip.shape[3]
        y = F.interpolate(y.permute(0, 3, 1, 2).contiguous(), size=(Hs, Ws), mode="bilinear",
                          align_corners=False).permute(0, 2, 3, 1).contiguous()  # :231-232 (identity unless odd sizes)
        x = y + skip.permute(0, 2, 3, 1).contiguous()
        for j in range(4):
            x = cvss_block(x, sd, f"{p}.layers_up.{i}.blocks.{j}")
        if i < 3:                                                               # UpsampleExpand, :43-51
            x = _ln(_up2(F.linear(x, sd[f"{p}.layers_up.{i}.upsample.linear.weight"])), sd,
                    f"{p}.layers_up.{i}.upsample.norm")
        y = x
    x = _ln(y, sd, p + ".norm_up")
    x = _up2(F.linear(x, sd[p + ".up.linear1.weight"]))                        # FinalUpsample_X4, :87-97
    x = _up2(F.linear(x, sd[p + ".up.linear2.weight"]))
    x = _ln(x, sd, p + ".up.norm")
    return F.conv2d(x.permute(0, 3, 1, 2).contiguous(), sd[p + ".output.weight"])


# ---- EncoderDecoder.encode_decode / forward, models/builder.py:128-157
def _logits(sd: SD, rgb, modal_x, backbone):
    feats = encoder(rgb, modal_x, sd, PRESETS[backbone]["depths"])
    out = decoder(feats, sd)
    return F.interpolate(out, size=rgb.shape[2:], mode="bilinear", align_corners=False), feats


@torch.no_grad()
def sigma_forward(sd: SD, rgb: torch.Tensor, modal_x: torch.Tensor, backbone: str = "sigma_tiny",
                  return_features: bool = False):
    """Logits (B, num_classes, H, W) of the reference model in eval mode, on CPU, fp32."""
    sd = {k: v.detach().to("cpu", torch.float32) for k, v in sd.items()}
    rgb, modal_x = rgb.detach().cpu().float(), modal_x.detach().cpu().float()
    out, feats = _logits(sd, rgb, modal_x, backbone)
    return (out, feats) if return_features else out


def sigma_gradients(sd: SD, rgb, modal_x, label, backbone: str = "sigma_tiny", names=None):
    """(logits, loss, {name: d loss / d parameter}) of the restated model: torch autograd over the plain functions of
    this file, the scans differentiated by the C oracle's backward recurrence (scan_oracle.c).  `names` = the state-dict
    entries that are parameters (default: every floating-point entry).  Independent of sigma_amd's modules, so it
    also catches host-side wiring mistakes of their backward (checkpoint hand-offs, merged projections, fused
    operators).  Pinned by tests/test_oracle_model.py against the reference's own gradient digests."""
    leaves = {}
    for k, v in sd.items():
        t = v.detach().to("cpu", torch.float32).clone()
        if v.is_floating_point() and (names is None or k in names):
            t.requires_grad_(True)
        leaves[k] = t
    rgb, modal_x = rgb.detach().cpu().float(), modal_x.detach().cpu().float()
    with torch.enable_grad():
        out, _ = _logits(leaves, rgb, modal_x, backbone)
        loss = F.cross_entropy(out, label.cpu().long(), ignore_index=255)
        loss.backward()
    grads = {k: t.grad for k, t in leaves.items() if t.requires_grad}
    return out.detach(), loss.detach(), grads


def sigma_loss(sd: SD, rgb, modal_x, label, backbone: str = "sigma_tiny") -> torch.Tensor:
    """CrossEntropy(ignore_index=255) of the logits (models/builder.py:153, train.py:75)."""
    return F.cross_entropy(sigma_forward(sd, rgb, modal_x, backbone), label.long(), ignore_index=255)
