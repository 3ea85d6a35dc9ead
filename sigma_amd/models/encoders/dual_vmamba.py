"""Siamese VMamba encoder with per-stage CroMB + ConMB fusion.

Mirror of /root/reference/models/encoders/dual_vmamba.py (RGBXTransformer :17-108, size presets
:113-144).  One difference in execution, none in the function computed: the shared-weight
backbone processes the RGB and the X image as ONE batch of 2B samples (the reference runs two
sequential passes, :85-86).  Every op in the backbone is per-sample (LayerNorm, per-sample
DropPath masks), so results are identical; it halves the launch count and doubles the rows
each scan kernel sees.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ...layernorm import LayerNorm

from .vmamba import Backbone_VSSM, ConcatMambaFusionBlock, CrossMambaFusionBlock


class RGBXTransformer(nn.Module):
    def __init__(self, num_classes=1000, norm_layer=LayerNorm, depths=(2, 2, 27, 2), dims=96, pretrained=None,
                 mlp_ratio=4.0, downsample_version="v1", ape=False, img_size=(480, 640), patch_size=4,
                 drop_path_rate=0.2, **kwargs):
        super().__init__()
        if ape:
            raise NotImplementedError("absolute position embedding is disabled in every Sigma preset")
        self.ape = False
        self.vssm = Backbone_VSSM(pretrained=pretrained, norm_layer=norm_layer, num_classes=num_classes,
                                  depths=depths, dims=dims, mlp_ratio=mlp_ratio,
                                  downsample_version=downsample_version, drop_path_rate=drop_path_rate)
        self.cross_mamba = nn.ModuleList(
            CrossMambaFusionBlock(hidden_dim=dims * (2 ** i), mlp_ratio=0.0, d_state=4) for i in range(4))
        self.channel_attn_mamba = nn.ModuleList(
            ConcatMambaFusionBlock(hidden_dim=dims * (2 ** i), mlp_ratio=0.0, d_state=4) for i in range(4))

    def forward_features(self, x_rgb: torch.Tensor, x_e: torch.Tensor):
        B = x_rgb.shape[0]
        feats = self.vssm(torch.cat([x_rgb, x_e], dim=0))           # 4 x (2B, C_i, H_i, W_i)
        fused = []
        for i, f in enumerate(feats):
            f = f.permute(0, 2, 3, 1)                                # NHWC views
            c_rgb, c_e = self.cross_mamba[i](f[:B], f[B:])           # CroMB
            # ConMB -> (B, C, H, W) for the decoder (dual_vmamba.py:100-104), as a channels-last view: the decoder
            # permutes back to (B, H, W, C) at once (MambaDecoder.py:222-232), so no NCHW copy is made and undone
            fused.append(self.channel_attn_mamba[i](c_rgb, c_e).permute(0, 3, 1, 2))
        return fused

    def forward(self, x_rgb, x_e):
        return self.forward_features(x_rgb, x_e)


class vssm_tiny(RGBXTransformer):
    def __init__(self, fuse_cfg=None, **kwargs):
        super().__init__(depths=[2, 2, 9, 2], dims=96, pretrained="pretrained/vmamba/vssmtiny_dp01_ckpt_epoch_292.pth",
                         mlp_ratio=0.0, downsample_version="v1", drop_path_rate=0.2)


class vssm_small(RGBXTransformer):
    def __init__(self, fuse_cfg=None, **kwargs):
        super().__init__(depths=[2, 2, 27, 2], dims=96, pretrained="pretrained/vmamba/vssmsmall_dp03_ckpt_epoch_238.pth",
                         mlp_ratio=0.0, downsample_version="v1", drop_path_rate=0.3)


class vssm_base(RGBXTransformer):
    def __init__(self, fuse_cfg=None, **kwargs):
        super().__init__(depths=[2, 2, 27, 2], dims=128, pretrained="pretrained/vmamba/vssmbase_dp06_ckpt_epoch_241.pth",
                         mlp_ratio=0.0, downsample_version="v1", drop_path_rate=0.6)
