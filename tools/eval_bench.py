#!/usr/bin/env python3
"""Forward-only (eval, no_grad) throughput of the HIP path -- BASELINE.json configs[1]:
    python tools/eval_bench.py [--backbone sigma_tiny] [--batch 2] [--height 480 --width 640] [--iters 10]"""
import argparse, json, os, sys, time, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backbone", default="sigma_tiny")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--classes", type=int, default=9)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    from sigma_amd.models.builder import EncoderDecoder
    from sigma_amd.tuning import enable_tuned_gemms
    enable_tuned_gemms()
    cfg = types.SimpleNamespace(backbone=a.backbone, decoder="MambaDecoder", num_classes=a.classes, image_height=a.height,
                                image_width=a.width, pretrained_model=None, bn_eps=1e-3, bn_momentum=0.1)
    cwd = os.getcwd(); os.chdir("/tmp")
    try:
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            model = EncoderDecoder(cfg).cuda().eval()
    finally:
        os.chdir(cwd)
    g = torch.Generator().manual_seed(0)
    rgb = torch.randn(a.batch, 3, a.height, a.width, generator=g).cuda()
    x = torch.randn(a.batch, 3, a.height, a.width, generator=g).cuda()
    with torch.no_grad():
        for _ in range(3):
            model(rgb, x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            model(rgb, x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.iters
    print(json.dumps(dict(metric=f"images/sec fwd only {a.backbone} {a.height}x{a.width}", value=round(a.batch / dt, 2),
                          ms_per_forward=round(dt * 1e3, 2), batch=a.batch, dtype="f32")))


if __name__ == "__main__":
    main()
