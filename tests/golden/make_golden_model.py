#!/usr/bin/env python3
"""Generate model-level golden fixtures by running the REFERENCE's own Python model on CPU.

Run in the build container only (needs /root/reference, read-only):

    python tests/golden/make_golden_model.py

What executes is the reference's code, unmodified and imported from where it lies:
models/builder.py (EncoderDecoder), models/encoders/dual_vmamba.py, models/encoders/vmamba.py,
models/decoders/MambaDecoder.py and the Python package models/encoders/selective_scan/selective_scan.
Only third-party modules that are not installed here are stubbed:
  * ``timm.models.layers``  -> DropPath (identity in eval) / trunc_normal_ / to_2tuple
  * ``fvcore.nn``           -> FLOP counters (never called)
  * ``selective_scan_cuda_core`` (the CUDA extension, cannot exist here) -> ``fwd`` = the
    reference's own ``selective_scan_ref`` (selective_scan_interface.py:86-131), ``bwd`` =
    torch autograd through that same function -- the exact oracle the reference's unit test
    uses for its CUDA kernel (test_selective_scan.py:186-201).
Weights: tests/golden/fill.py (name-keyed deterministic fill); inputs: fill.make_inputs.
Outputs: tests/golden/model_<cfg>.npz with logits, loss, and gradient digests.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import fill  # noqa: E402

REF = "/root/reference"


def install_stubs():
    # --- timm
    timm = types.ModuleType("timm")
    tm = types.ModuleType("timm.models")
    tl = types.ModuleType("timm.models.layers")

    class DropPath(torch.nn.Module):
        def __init__(self, drop_prob=0.0, scale_by_keep=True):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            assert not self.training, "golden fixtures are generated in eval mode"
            return x

    tl.DropPath = DropPath
    tl.trunc_normal_ = lambda t, mean=0.0, std=1.0, a=-2.0, b=2.0: torch.nn.init.trunc_normal_(t, mean, std, a, b)
    tl.to_2tuple = lambda v: v if isinstance(v, (tuple, list)) else (v, v)
    timm.models, tm.layers = tm, tl
    sys.modules.update({"timm": timm, "timm.models": tm, "timm.models.layers": tl})
    # --- fvcore
    fv = types.ModuleType("fvcore")
    fn = types.ModuleType("fvcore.nn")
    for n in ("FlopCountAnalysis", "flop_count_str", "flop_count", "parameter_count"):
        setattr(fn, n, None)
    fv.nn = fn
    sys.modules.update({"fvcore": fv, "fvcore.nn": fn})
    # --- the reference's own python package `selective_scan` (+ a stub for the CUDA extension)
    core = types.ModuleType("selective_scan_cuda_core")
    sys.modules["selective_scan_cuda_core"] = core
    sys.path.insert(0, os.path.join(REF, "models", "encoders", "selective_scan"))
    sys.path.insert(0, REF)
    ssi = importlib.import_module("selective_scan.selective_scan_interface")
    ref_fn = ssi.selective_scan_ref

    def fwd(u, delta, A, B, C, D, delta_bias, delta_softplus, nrows):
        out = ref_fn(u, delta, A, B, C, D, delta_bias, delta_softplus)
        n_chunks = (u.shape[-1] + 2047) // 2048
        return [out, u.new_zeros(u.shape[0], u.shape[1], n_chunks, 2 * A.shape[1])]

    def bwd(u, delta, A, B, C, D, delta_bias, dout, x, delta_softplus, nrows):
        leaves = [t.detach().clone().requires_grad_() if t is not None else None
                  for t in (u, delta, A, B, C, D, delta_bias)]
        with torch.enable_grad():
            out = ref_fn(*leaves[:5], leaves[5], leaves[6], delta_softplus)
        live = [t for t in leaves if t is not None]
        grads = list(torch.autograd.grad(out, live, dout))
        return [grads.pop(0) if t is not None else None for t in leaves]

    core.fwd, core.bwd = fwd, bwd


CASES = [
    # name, backbone, num_classes, H, W, batch
    dict(name="tiny_64x96", backbone="sigma_tiny", num_classes=9, H=64, W=96, batch=1),
    dict(name="tiny_72x88_b2", backbone="sigma_tiny", num_classes=5, H=72, W=88, batch=2),   # odd stage sizes: 9x11 -> 5x6
]


def digest(t: torch.Tensor):
    t = t.detach().double().flatten()
    w = torch.cos(torch.arange(t.numel(), dtype=torch.float64) * 0.37)      # position-sensitive
    return np.array([t.sum().item(), t.abs().sum().item(), (t * w).sum().item()])


def main():
    install_stubs()
    os.chdir("/tmp")                                   # the reference tries to open pretrained/...: let it fail
    from models.builder import EncoderDecoder
    for c in CASES:
        cfg = types.SimpleNamespace(backbone=c["backbone"], decoder="MambaDecoder", num_classes=c["num_classes"],
                                    image_height=c["H"], image_width=c["W"], pretrained_model=None, bn_eps=1e-3,
                                    bn_momentum=0.1, decoder_embed_dim=512)
        torch.manual_seed(0)
        model = EncoderDecoder(cfg=cfg, criterion=torch.nn.CrossEntropyLoss(reduction="mean", ignore_index=255),
                               norm_layer=torch.nn.BatchNorm2d)
        fill.fill_parameters(model)
        model.eval()
        rgb, x, label = fill.make_inputs(c["batch"], c["H"], c["W"], c["num_classes"])
        with torch.no_grad():
            logits = model(rgb, x)
            feats = model.backbone(rgb, x)
        loss = model(rgb, x, label)
        loss.backward()
        blob = {"meta": np.array(repr(c)), "logits": logits.numpy(), "loss": np.array(loss.item()),
                "n_params": np.array(sum(p.numel() for p in model.parameters())),
                "keys": np.array(sorted(model.state_dict().keys()))}
        for i, f in enumerate(feats):
            blob[f"feat{i}"] = f.numpy() if f.numel() < 40000 else digest(f)
        gnames, gdig = [], []
        for n, p in model.named_parameters():
            gnames.append(n)
            gdig.append(digest(p.grad) if p.grad is not None else np.full(3, np.nan))
        blob["grad_names"] = np.array(gnames)
        blob["grad_digest"] = np.stack(gdig)
        # full gradients of the scan-adjacent parameters (small tensors) for a per-element comparison
        blocks = ("vssm.layers.0.blocks.0.", "vssm.layers.2.blocks.0.", "vssm.layers.2.blocks.8.", "vssm.layers.3.blocks.1.",
                  "cross_mamba.0.", "cross_mamba.3.", "channel_attn_mamba.0.", "channel_attn_mamba.3.",
                  "layers_up.1.blocks.0.", "layers_up.3.blocks.3.")
        full = [n for n, p in model.named_parameters() if p.grad is not None and p.numel() <= 40000 and any(b in n for b in blocks)
                and any(t in n for t in ("x_proj", "dt_proj", "A_log", ".Ds", ".D_1", ".D_2", "conv2d.bias", "out_norm", "scale1", "scale2"))]
        blob["grad_full_names"] = np.array(full)
        named = dict(model.named_parameters())
        for i, n in enumerate(full):
            blob[f"grad_full_{i}"] = named[n].grad.detach().numpy()
        path = os.path.join(HERE, f"model_{c['name']}.npz")
        np.savez_compressed(path, **blob)
        print(path, "logits", tuple(logits.shape), "loss %.6f" % loss.item(), "%.0f KiB" % (os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
