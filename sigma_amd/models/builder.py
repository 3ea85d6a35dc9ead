"""``EncoderDecoder`` -- drop-in boundary #1.

Mirror of /root/reference/models/builder.py:13-166 for the Sigma configurations
(``cfg.backbone`` in {sigma_tiny, sigma_small, sigma_base}, ``cfg.decoder == 'MambaDecoder'``):
same constructor signature, same attributes (``backbone``, ``decode_head``, ``criterion``,
``deep_supervision``), same ``forward(rgb, modal_x, label=None)`` contract (loss when a label
is given, logits (B, num_classes, H, W) otherwise) and the same state_dict keys.
Other backbones / decoders of the reference (SegFormer, Swin, UPerNet, ...) are other model
families and out of this path's scope; asking for them raises.
"""
from __future__ import annotations

import logging

import torch.nn as nn
import torch.nn.functional as F

logger = logging.getLogger("sigma_amd")

_BACKBONES = {
    "sigma_tiny": ("vssm_tiny", [96, 192, 384, 768]),
    "sigma_small": ("vssm_small", [96, 192, 384, 768]),
    "sigma_base": ("vssm_base", [128, 256, 512, 1024]),
}


def _init_decoder_weights(module: nn.Module, norm_layer, bn_eps, bn_momentum):
    """utils/init_func.py:10-29 as called from builder.py:120-122: kaiming-normal (fan_in, relu)
    on every conv of the decode head; `norm_layer` instances (BatchNorm in the reference's
    train.py -- none exist in the Mamba decoder) get eps/momentum and (1, 0)."""
    for m in module.modules():
        if isinstance(m, (nn.Conv1d, nn.Conv2d, nn.Conv3d)):
            nn.init.kaiming_normal_(m.weight, mode="fan_in", nonlinearity="relu")
        elif isinstance(m, norm_layer):
            m.eps = bn_eps
            m.momentum = bn_momentum
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)


class EncoderDecoder(nn.Module):
    def __init__(self, cfg=None, criterion=nn.CrossEntropyLoss(reduction="mean", ignore_index=255),
                 norm_layer=nn.BatchNorm2d):
        super().__init__()
        # The committed solution table for the remaining vendor GEMMs (sigma_amd/tuning.py) switches PyTorch's TunableOp
        # on PROCESS-WIDE: constructing a model does not do that behind the caller's back.  Entry points call
        # sigma_amd.tuning.enable_tuned_gemms() themselves (bench.py, tools/); an unchanged train.py opts in with
        # SIGMA_TUNED_GEMMS=1 in its environment.
        import os
        if os.environ.get("SIGMA_TUNED_GEMMS") == "1":
            from ..tuning import enable_tuned_gemms
            enable_tuned_gemms()
        else:
            from ..tuning import table_path, tuned_gemms_requested
            if os.path.exists(table_path()) and not tuned_gemms_requested() and not getattr(EncoderDecoder, "_tuning_note", False):
                EncoderDecoder._tuning_note = True           # once per process
                logger.info("sigma_amd: the tuned vendor-GEMM table %s is NOT in use (a few per cent of the step): set "
                            "SIGMA_TUNED_GEMMS=1 or call sigma_amd.tuning.enable_tuned_gemms() before training", table_path())
        self.norm_layer = norm_layer
        if cfg.backbone not in _BACKBONES:
            raise NotImplementedError(
                f"backbone {cfg.backbone!r}: only the Sigma (VMamba) backbones {sorted(_BACKBONES)} are on this path")
        name, self.channels = _BACKBONES[cfg.backbone]
        logger.info("Using backbone: V-MAMBA (%s)", cfg.backbone)
        from .encoders import dual_vmamba
        self.backbone = getattr(dual_vmamba, name)()
        self.aux_head = None
        if cfg.decoder != "MambaDecoder":
            raise NotImplementedError("only cfg.decoder == 'MambaDecoder' is a live configuration of the reference "
                                      "(every other decoder hits the deep_supervision AttributeError, builder.py:131)")
        logger.info("Using Mamba Decoder")
        from .decoders.MambaDecoder import MambaDecoder
        self.deep_supervision = False
        self.decode_head = MambaDecoder(img_size=[cfg.image_height, cfg.image_width], in_channels=self.channels,
                                        num_classes=cfg.num_classes, embed_dim=self.channels[0],
                                        deep_supervision=self.deep_supervision)
        self.criterion = criterion
        if self.criterion:
            self.init_weights(cfg, pretrained=getattr(cfg, "pretrained_model", None))
        # nn.Linear GEMMs: hand-written split-operand bf16 MFMA kernels (sigma_amd/gemm.py, csrc/gemm_split.hip);
        # SIGMA_GEMM=fp32 keeps the vendor fp32 GEMMs
        from ..gemm import enable_split3_linears, gemm_mode
        self.gemm_mode = gemm_mode()
        self.split_linears = 0
        if self.gemm_mode == "split3":
            self.split_linears = enable_split3_linears(self)

    def init_weights(self, cfg, pretrained=None):
        if pretrained:
            # builder.py:110-114 calls backbone.init_weights, which the Sigma backbone does not define
            raise AttributeError("cfg.pretrained_model must stay None for Sigma backbones (configs/*: 'do not need to change')")
        logger.info("Initing weights ...")
        _init_decoder_weights(self.decode_head, self.norm_layer, cfg.bn_eps, cfg.bn_momentum)

    def encode_decode(self, rgb, modal_x):
        """backbone -> decoder -> bilinear resize to the input size (builder.py:128-144)."""
        feats = self.backbone(rgb, modal_x)
        out = self.decode_head(feats)
        if tuple(out.shape[2:]) == tuple(rgb.shape[2:]):
            # builder.py:135 resizes to the input size; the Mamba decoder already delivers it, and a bilinear resize to the
            # same size (align_corners=False) is the identity (source index == destination index, weight 1): skipped
            return out
        return F.interpolate(out, size=rgb.shape[2:], mode="bilinear", align_corners=False)

    def forward(self, rgb, modal_x, label=None):
        out = self.encode_decode(rgb, modal_x)
        if label is not None:
            # mean cross entropy straight from the channels-last logits of the classifier GEMM (csrc/pointwise.hip); any
            # other criterion / layout: the criterion itself on the (B, nc, H, W) view
            from ..pointwise import cross_entropy
            loss = cross_entropy(self.criterion, out, label) if out.is_cuda else None
            return loss if loss is not None else self.criterion(out, label.long())
        return out if out.is_contiguous() else out.contiguous()     # callers get the reference's (B, nc, H, W) layout
