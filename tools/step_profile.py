#!/usr/bin/env python3
"""Attribute the GPU time of one training step to operators and Python call sites (torch.profiler).

    python tools/step_profile.py [--batch 8] [--out gpurun_out/x/step_profile.txt]
"""
import argparse
import os
import sys
import types

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--backbone", default="sigma_small")
    ap.add_argument("--out", default="")
    ap.add_argument("--stack", type=int, default=6)
    a = ap.parse_args()
    from sigma_amd import train_step as ts
    from sigma_amd.models.builder import EncoderDecoder
    dev = torch.device("cuda", 0)
    cfg = types.SimpleNamespace(backbone=a.backbone, decoder="MambaDecoder", num_classes=40, image_height=480, image_width=640,
                                pretrained_model=None, bn_eps=1e-3, bn_momentum=0.1)
    cwd = os.getcwd()
    os.chdir("/tmp")
    try:
        import contextlib, io
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
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        step()
        torch.cuda.synchronize()
    lines = []
    ka = prof.key_averages(group_by_stack_n=a.stack, group_by_input_shape=True)
    rows = []
    for e in ka:
        t = getattr(e, "self_device_time_total", None)
        if t is None:
            t = getattr(e, "self_cuda_time_total", 0)
        if t <= 0:
            continue
        stack = [s for s in (e.stack or []) if "sigma_amd" in s or "bench" in s or "tools/" in s]
        rows.append((t, e.count, e.key, str(e.input_shapes)[:120], " <- ".join(s.split("/")[-1] for s in stack[:3])))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    lines.append(f"# one training step, {a.backbone} batch {a.batch}: GPU self time by operator + input shapes + first sigma_amd frames; total {tot / 1e3:.1f} ms")
    for t, c, k, sh, st in rows[:120]:
        lines.append(f"{t / 1e3:9.2f} ms {100 * t / tot:5.1f}% x{c:<4d} {k[:48]:48s} {sh:60s} {st}")
    # second table: ATen operators only (elementwise / copies / reductions), by input shapes -> call sites
    fam = {}
    for e in ka:
        if not e.key.startswith("aten::"):
            continue
        t = getattr(e, "self_device_time_total", None)
        if t is None:
            t = getattr(e, "self_cuda_time_total", 0)
        if t <= 0:
            continue
        k = (e.key, str(e.input_shapes)[:150])
        d = fam.setdefault(k, [0.0, 0])
        d[0] += t
        d[1] += e.count
    lines.append("")
    lines.append("# aten operators by input shapes (self GPU time)")
    agg = {}
    for (k, sh), (t, c) in fam.items():
        agg[k] = agg.get(k, 0.0) + t
    for k, t in sorted(agg.items(), key=lambda kv: -kv[1])[:25]:
        lines.append(f"{t / 1e3:9.2f} ms  {k}")
    lines.append("")
    for (k, sh), (t, c) in sorted(fam.items(), key=lambda kv: -kv[1][0])[:150]:
        if k in ("aten::mm", "aten::bmm", "aten::addmm", "aten::convolution_backward", "aten::miopen_convolution", "aten::_convolution"):
            continue
        lines.append(f"{t / 1e3:9.2f} ms x{c:<4d} {k:34s} {sh}")
    text = "\n".join(lines)
    print(text)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)) or ".", exist_ok=True)
        open(a.out, "w").write(text + "\n")


if __name__ == "__main__":
    main()
