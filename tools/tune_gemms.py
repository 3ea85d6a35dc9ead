#!/usr/bin/env python3
"""Tune the library GEMMs of the training / eval step with PyTorch TunableOp (rocBLAS + hipBLASLt solution
search) and write the result table that sigma_amd/tuning.py loads at run time.

    python tools/tune_gemms.py --out gpurun_out/x/tunableop_results.csv [--batch 8] [--backbones sigma_small]

Runs on the GPU box (minutes: every distinct GEMM shape is timed against every candidate solution once)."""
import argparse
import contextlib
import io
import os
import sys
import time
import types

import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--batch", default="8")
    ap.add_argument("--backbones", default="sigma_small")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--classes", type=int, default=40)
    ap.add_argument("--max-ms", type=int, default=20)
    ap.add_argument("--max-iters", type=int, default=20)
    a = ap.parse_args()
    import torch.cuda.tunable as tun
    os.makedirs(os.path.dirname(os.path.abspath(a.out)) or ".", exist_ok=True)
    tun.enable(True)
    tun.tuning_enable(True)
    tun.set_filename(a.out)
    tun.set_max_tuning_duration(a.max_ms)
    tun.set_max_tuning_iterations(a.max_iters)
    from sigma_amd import train_step as ts
    from sigma_amd.models.builder import EncoderDecoder
    dev = torch.device("cuda", 0)
    for backbone in a.backbones.split(","):
        cfg = types.SimpleNamespace(backbone=backbone, decoder="MambaDecoder", num_classes=a.classes, image_height=a.height, image_width=a.width,
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
        for batch in [int(b) for b in a.batch.split(",")]:
            g = torch.Generator(device="cpu").manual_seed(1234)
            rgb = torch.randn(batch, 3, a.height, a.width, generator=g).to(dev)
            mx = torch.randn(batch, 3, a.height, a.width, generator=g).to(dev)
            label = torch.randint(0, a.classes, (batch, a.height, a.width), generator=g).to(dev)
            step = ts.make_step(model, opt, (rgb, mx, label))
            t0 = time.time()
            step()
            torch.cuda.synchronize()
            print(f"{backbone} batch {batch}: tuning step took {time.time() - t0:.1f} s, {len(tun.get_results())} entries", flush=True)
            step()
            torch.cuda.synchronize()
        del model, opt
        torch.cuda.empty_cache()
    tun.tuning_enable(False)
    # the result file is written incrementally by PyTorch; keep a copy of the in-memory table as well
    import json
    with open(a.out + ".json", "w") as f:
        json.dump(dict(validators=[list(v) for v in tun.get_validators()], results=[list(r) for r in tun.get_results()]), f, indent=0)
    print("entries:", len(tun.get_results()), "file:", a.out, os.path.exists(a.out), flush=True)


if __name__ == "__main__":
    main()
