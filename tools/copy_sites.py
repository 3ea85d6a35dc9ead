#!/usr/bin/env python3
"""Where do the layout copies of a training step come from?  Wraps Tensor.contiguous / reshape / flatten / clone /
torch.cat / torch.stack for one step and lists, per call site inside sigma_amd/, the bytes actually copied.

    python tools/copy_sites.py [--batch 8]
"""
import argparse
import collections
import contextlib
import io
import os
import sys
import traceback
import types

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SITES = collections.defaultdict(lambda: [0, 0])


def _site():
    for fr in reversed(traceback.extract_stack(limit=14)[:-2]):
        if "sigma_amd" in fr.filename and "copy_sites" not in fr.filename:
            return f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno} {fr.line.strip()[:90]}"
    return "?"


def _note(op, res, srcs):
    if not isinstance(res, torch.Tensor) or not res.is_cuda or res.numel() < 50000:
        return
    ptrs = {s.untyped_storage().data_ptr() for s in srcs if isinstance(s, torch.Tensor)}
    if res.untyped_storage().data_ptr() in ptrs:
        return                                   # a view: nothing copied
    d = SITES[(op, _site(), tuple(res.shape))]
    d[0] += 1
    d[1] += res.numel() * res.element_size()


def wrap_method(name):
    orig = getattr(torch.Tensor, name)

    def f(self, *a, **k):
        r = orig(self, *a, **k)
        _note(name, r, [self])
        return r
    setattr(torch.Tensor, name, f)
    return orig


def wrap_fn(name):
    orig = getattr(torch, name)

    def f(ts, *a, **k):
        r = orig(ts, *a, **k)
        _note(name, r, list(ts))
        return r
    setattr(torch, name, f)
    return orig


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    from sigma_amd import train_step as ts
    from sigma_amd.models.builder import EncoderDecoder
    dev = torch.device("cuda", 0)
    cfg = types.SimpleNamespace(backbone="sigma_small", decoder="MambaDecoder", num_classes=40, image_height=480, image_width=640,
                                pretrained_model=None, bn_eps=1e-3, bn_momentum=0.1)
    cwd = os.getcwd()
    os.chdir("/tmp")
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            model = EncoderDecoder(cfg, criterion=nn.CrossEntropyLoss(reduction="mean", ignore_index=255), norm_layer=nn.BatchNorm2d)
    finally:
        os.chdir(cwd)
    model.to(dev).train()
    opt = ts.make_optimizer(model)
    g = torch.Generator(device="cpu").manual_seed(1234)
    rgb = torch.randn(a.batch, 3, 480, 640, generator=g).to(dev)
    mx = torch.randn(a.batch, 3, 480, 640, generator=g).to(dev)
    label = torch.randint(0, 40, (a.batch, 480, 640), generator=g).to(dev)
    step = ts.make_step(model, opt, (rgb, mx, label))
    step()
    torch.cuda.synchronize()
    for n in ("contiguous", "reshape", "flatten", "clone", "float"):
        wrap_method(n)
    for n in ("cat", "stack"):
        wrap_fn(n)
    step()
    torch.cuda.synchronize()
    tot = sum(v[1] for v in SITES.values())
    print(f"# explicit layout copies of one step (batch {a.batch}): {tot / 1e9:.2f} GB written in {sum(v[0] for v in SITES.values())} calls")
    for (op, site, shape), (n, b) in sorted(SITES.items(), key=lambda kv: -kv[1][1])[:60]:
        print(f"{b / 1e6:9.1f} MB x{n:<3d} {op:10s} {str(shape):28s} {site}")


if __name__ == "__main__":
    main()
