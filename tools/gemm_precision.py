#!/usr/bin/env python3
"""GEMM precision experiment: nn.Linear GEMMs with bf16 inputs and fp32 accumulation/output
against the shipped fp32 GEMMs, on sigma_small at 480x640.

    python tools/gemm_precision.py [--batch 8] [--steps 4]

Prints one JSON line per mode: logits error against the fp32 run (eval mode, same weights and inputs), loss and
gradient error of one training step, and the step time (fwd + bwd + AdamW, HIP events).  The bf16 mode patches
torch.nn.functional.linear only -- in_proj / out_proj / PatchMerging reduction / decoder and fusion linears, i.e.
the GEMMs that carry the FLOPs; the scan-side matmuls of SS2DCoreFn (x_proj, dt_proj: K = d -> 38..56) and the
MIOpen convolutions stay fp32.
"""
import argparse
import contextlib
import io
import json
import os
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

_orig_linear = F.linear


def _bf16_linear(x, w, b=None):
    if x.dtype != torch.float32 or not x.is_cuda:
        return _orig_linear(x, w, b)
    y = _orig_linear(x.to(torch.bfloat16), w.to(torch.bfloat16)).float()       # MFMA bf16, fp32 accumulate inside
    return y if b is None else y + b


@contextlib.contextmanager
def linear_mode(mode):
    if mode == "bf16":
        F.linear = _bf16_linear
        torch.nn.functional.linear = _bf16_linear
    try:
        yield
    finally:
        F.linear = _orig_linear
        torch.nn.functional.linear = _orig_linear


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=4)
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
            torch.manual_seed(0)
            model = EncoderDecoder(cfg, criterion=nn.CrossEntropyLoss(reduction="mean", ignore_index=255), norm_layer=nn.BatchNorm2d)
    finally:
        os.chdir(cwd)
    model.to(dev)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    g = torch.Generator(device="cpu").manual_seed(1234)
    rgb = torch.randn(a.batch, 3, 480, 640, generator=g).to(dev)
    x = torch.randn(a.batch, 3, 480, 640, generator=g).to(dev)
    label = torch.randint(0, 40, (a.batch, 480, 640), generator=g).to(dev)
    ref = {}
    for mode in ("fp32", "bf16"):
        model.load_state_dict(state)
        with linear_mode(mode):
            model.eval()
            with torch.no_grad():
                logits = model(rgb[:1], x[:1]).float()
            model.train()
            opt = ts.make_optimizer(model)
            for p in model.parameters():
                p.grad = None
            loss = model(rgb, x, label)
            loss.backward()
            grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
            opt.step()
            for _ in range(2):                      # warm-up: MIOpen / TunableOp lookups, allocator
                opt.zero_grad(set_to_none=True)
                model(rgb, x, label).backward()
                opt.step()
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(a.steps):
                opt.zero_grad(set_to_none=True)
                model(rgb, x, label).backward()
                opt.step()
            e.record()
            torch.cuda.synchronize()
            ms = s.elapsed_time(e) / a.steps
        rec = dict(mode=mode, batch=a.batch, ms_per_step=round(ms, 2), images_per_s=round(a.batch / ms * 1e3, 2), loss=round(float(loss), 5))
        if mode == "fp32":
            ref = dict(logits=logits, grads=grads, loss=float(loss))
        else:
            d = (logits - ref["logits"]).abs()
            rec["logits_max_abs_err"] = float(d.max())
            rec["logits_rel_err_max_over_maxabs"] = float(d.max() / ref["logits"].abs().max())
            rec["logits_rel_l2"] = float(d.norm() / ref["logits"].norm())
            rec["argmax_agreement"] = float((logits.argmax(1) == ref["logits"].argmax(1)).float().mean())
            rec["loss_abs_err"] = abs(float(loss) - ref["loss"])
            worst, name = 0.0, ""
            num = den = 0.0
            for n, gr in grads.items():
                r = ref["grads"][n]
                err = float((gr - r).norm() / (r.norm() + 1e-12))
                num += float((gr - r).norm() ** 2)
                den += float(r.norm() ** 2)
                if err > worst:
                    worst, name = err, n
            rec["grad_rel_l2_all_params"] = (num / den) ** 0.5
            rec["grad_rel_l2_worst_param"] = [name, worst]
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
