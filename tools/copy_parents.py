#!/usr/bin/env python3
"""Which autograd nodes / forward operators launch the ATen copy, add, mul, fill and reduce kernels of one training
step?  torch.profiler event tree: every such kernel is attributed to the nearest enclosing
`autograd::engine::evaluate_function: <Node>` (backward) or to the top-level aten operator chain (forward).

    python tools/copy_parents.py [--batch 8]
"""
import argparse
import collections
import contextlib
import io
import os
import sys
import types

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FAMILIES = ("aten::copy_", "aten::add", "aten::add_", "aten::mul", "aten::mul_", "aten::fill_", "aten::zero_", "aten::sum", "aten::cat",
            "aten::div", "aten::neg", "aten::sub", "aten::clone", "aten::contiguous", "aten::zeros", "aten::zeros_like")


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
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0.0, 0])
    tot = 0.0
    for e in prof.events():
        if e.name not in FAMILIES:
            continue
        t = getattr(e, "self_device_time_total", None)
        if t is None:
            t = getattr(e, "self_cuda_time_total", 0)
        if t <= 0:
            continue
        # walk up: nearest autograd node, else the outermost aten / custom op
        p, node, top = e.cpu_parent, None, e.name
        while p is not None:
            if p.name.startswith("autograd::engine::evaluate_function:"):
                node = p.name.split(":", 3)[-1].strip()
                break
            top = p.name
            p = p.cpu_parent
        site = ""
        if node is None:                              # forward: the innermost sigma_amd frame of the Python stack
            q = e
            while q is not None and not site:
                for fr in (q.stack or []):
                    if "sigma_amd" in fr:
                        site = " @" + fr.split("sigma_amd/")[-1][:60]
                        break
                q = q.cpu_parent
        key = (("bwd " + node) if node else ("fwd " + top + site), e.name, str(e.input_shapes)[:70])
        agg[key][0] += t
        agg[key][1] += 1
        tot += t
    print(f"# ATen copy / elementwise / reduce kernels of one step (batch {a.batch}): {tot / 1e3:.1f} ms; by launching node")
    by_node = collections.defaultdict(float)
    for (node, op, sh), (t, c) in agg.items():
        by_node[node] += t
    for node, t in sorted(by_node.items(), key=lambda kv: -kv[1])[:30]:
        print(f"{t / 1e3:8.2f} ms  {node}")
    print()
    for (node, op, sh), (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:60]:
        print(f"{t / 1e3:8.2f} ms x{c:<4d} {node[:84]:84s} {op:14s} {sh}")


if __name__ == "__main__":
    main()
