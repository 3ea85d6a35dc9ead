import ast
import os
import sys
import types

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLDEN)
import fill  # noqa: E402,F401


def cfg_for(backbone="sigma_tiny", num_classes=9, H=480, W=640):
    return types.SimpleNamespace(backbone=backbone, decoder="MambaDecoder", num_classes=num_classes, image_height=H,
                                 image_width=W, pretrained_model=None, bn_eps=1e-3, bn_momentum=0.1,
                                 decoder_embed_dim=512)


def build_model(backbone="sigma_tiny", num_classes=9, H=480, W=640, criterion=True):
    from sigma_amd.models.builder import EncoderDecoder
    crit = torch.nn.CrossEntropyLoss(reduction="mean", ignore_index=255) if criterion else None
    cwd = os.getcwd()
    os.chdir("/tmp")                  # the (absent) pretrained/ path is relative, as in the reference
    try:
        model = EncoderDecoder(cfg_for(backbone, num_classes, H, W), criterion=crit, norm_layer=torch.nn.BatchNorm2d)
    finally:
        os.chdir(cwd)
    fill.fill_parameters(model)
    return model


def load_model_golden(name):
    z = np.load(os.path.join(GOLDEN, f"model_{name}.npz"), allow_pickle=False)
    meta = ast.literal_eval(str(z["meta"]))
    return meta, z


def digest(t: torch.Tensor):
    t = t.detach().double().flatten().cpu()
    w = torch.cos(torch.arange(t.numel(), dtype=torch.float64) * 0.37)
    return np.array([t.sum().item(), t.abs().sum().item(), (t * w).sum().item()])


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max |a - b| / max |b|: error relative to the scale of the reference tensor.  (A per-element
    ratio is meaningless for logits that cross zero; BASELINE.json's "within 1e-3 rel" is read
    against the logit scale, and elementwise closeness is asserted separately with allclose.)"""
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).abs().max() / b.abs().max())


def assert_logits_close(a: torch.Tensor, b: torch.Tensor, rel: float):
    assert a.shape == b.shape
    e = rel_err(a, b)
    assert e < rel, f"max|a-b|/max|b| = {e:.3e} >= {rel:.1e}"
    torch.testing.assert_close(a.double().cpu(), b.double().cpu(), rtol=10 * rel, atol=rel * float(b.abs().max()))
