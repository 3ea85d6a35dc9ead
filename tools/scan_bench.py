#!/usr/bin/env python3
"""Operator-level micro-benchmark of the HIP selective scan (fwd and bwd) on one MI355X.

    python tools/scan_bench.py [--sweep] [--shapes s0,s1,...] [--iters 20] [--out file.jsonl]

Times the kernels with HIP events on the stream they are launched on (torch's current
stream) and prints algorithmic GB/s per SURVEY.md 8(d):
    fwd bytes = s*(3*B*KD*L) + s*(2*B*G*N*L) + 4*(KD*N + 2*KD) [+ checkpoints]
    bwd bytes = s*(5*B*KD*L) + s*(4*B*G*N*L)
--sweep tries every (items per lane, rows per workgroup) launch geometry.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from sigma_amd import _capi  # noqa: E402
from sigma_amd import selective_scan_cuda_core as core  # noqa: E402

SHAPES = {
    # name: (batch, KD, L, N, G)   SURVEY.md Appendix A (sigma_tiny/small @ 480x640)
    "enc_s0": (1, 768, 19200, 16, 4),
    "enc_s1": (1, 1536, 4800, 16, 4),
    "enc_s2": (1, 3072, 1200, 16, 4),
    "enc_s3": (1, 6144, 300, 16, 4),
    "cromb_s0": (1, 192, 19200, 4, 1),
    "conmb_s0": (1, 384, 38400, 4, 2),
    "dec_s0": (1, 768, 19200, 4, 4),
    "enc_s0_b8": (8, 768, 19200, 16, 4),
    "enc_s2_b8": (8, 3072, 1200, 16, 4),
    "dec_s0_b8": (8, 768, 19200, 4, 4),
    "conmb_s0_b8": (8, 384, 38400, 4, 2),
    "dec_s1_b8": (8, 1536, 4800, 4, 4),
    "dec_s2_b8": (8, 3072, 1200, 4, 4),
    "cromb_s0_b8": (8, 192, 19200, 4, 1),
    "enc_s0_b16": (16, 768, 19200, 16, 4),
    "enc_s1_b16": (16, 1536, 4800, 16, 4),
    "enc_s2_b16": (16, 3072, 1200, 16, 4),
    "enc_s3_b16": (16, 6144, 300, 16, 4),
    # sigma_base @720x1280, batch 1 (BASELINE configs[4]): both modalities stacked
    "base_s0_b2": (2, 1024, 57600, 16, 4),
    "base_s1_b2": (2, 2048, 14400, 16, 4),
    "base_s2_b2": (2, 4096, 3600, 16, 4),
    "base_s3_b2": (2, 8192, 900, 16, 4),
    "enc_s0_b2": (2, 768, 19200, 16, 4),
    "enc_s1_b2": (2, 1536, 4800, 16, 4),
    "enc_s2_b2": (2, 3072, 1200, 16, 4),
    "enc_s3_b2": (2, 6144, 300, 16, 4),
    "cromb_s1_b8": (8, 384, 4800, 4, 1),
    "cromb_s2_b8": (8, 768, 1200, 4, 1),
    "cromb_s3_b8": (8, 1536, 300, 4, 1),
    "conmb_s1_b8": (8, 768, 9600, 4, 2),
    "conmb_s2_b8": (8, 1536, 2400, 4, 2),
    "conmb_s3_b8": (8, 3072, 600, 4, 2),
    "dec_s0": (1, 768, 19200, 4, 4),
    "dec_s1": (1, 1536, 4800, 4, 4),
    "dec_s2": (1, 3072, 1200, 4, 4),
    "cromb_s1": (1, 384, 4800, 4, 1),
    "cromb_s2": (1, 768, 1200, 4, 1),
    "cromb_s3": (1, 1536, 300, 4, 1),
    "conmb_s1": (1, 768, 9600, 4, 2),
    "conmb_s2": (1, 1536, 2400, 4, 2),
    "conmb_s3": (1, 3072, 600, 4, 2),
}

HBM_PEAK = 8.0e12


def fwd_bytes(B, KD, L, N, G, s=4):
    return s * 3 * B * KD * L + s * 2 * B * G * N * L + 4 * (KD * N + 2 * KD)


def bwd_bytes(B, KD, L, N, G, s=4):
    return s * 5 * B * KD * L + s * 4 * B * G * N * L


def make(shape, dtype=torch.float32, dev="cuda"):
    B, KD, L, N, G = shape
    g = torch.Generator(device="cpu").manual_seed(0)
    A = (-torch.arange(1, N + 1, dtype=torch.float32).repeat(KD, 1)).to(dev)
    u = torch.randn(B, KD, L, generator=g).to(dev, dtype)
    delta = (0.5 * torch.randn(B, KD, L, generator=g)).to(dev, dtype)
    Bm = torch.randn(B, G, N, L, generator=g).to(dev, dtype)
    Cm = torch.randn(B, G, N, L, generator=g).to(dev, dtype)
    D = torch.ones(KD, device=dev)
    bias = torch.full((KD,), -4.0, device=dev)
    dout = torch.randn(B, KD, L, generator=g).to(dev, dtype)
    return u, delta, A, Bm, Cm, D, bias, dout


def time_call(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default=",".join(SHAPES))
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--dtype", default="float32")
    ap.add_argument("--out", default="")
    ap.add_argument("--fine", action="store_true", help="one checkpoint per backward tile with the pitch the fused model "
                    "path picks for the shape (sigma_amd.ss2d_fused.ckpt_pitch_for): the launches of the training step")
    ap.add_argument("--opt", action="append", default=[], help="library option name=value (sigma_scan_set_option), repeatable")
    ap.add_argument("--pitch", type=int, default=0, help="force this checkpoint pitch (with --fine): 16 = row-lane kernels, 160 = quad-row")
    a = ap.parse_args()
    if os.environ.get("SIGMA_BENCH_NO_SELFTEST"):       # timing-only ablation builds (wrong results by construction)
        core._ROWLANE_TESTED.update(range(16))
    for kv in a.opt:
        k, v = kv.split("=")
        _capi.set_option(k, int(v))
    dt = getattr(torch, a.dtype)
    es = 4 if dt == torch.float32 else 2
    rows = []
    for name in a.shapes.split(","):
        shape = SHAPES[name]
        u, delta, A, Bm, Cm, D, bias, dout = make(shape, dt)
        pitch = 0
        if a.fine:
            from sigma_amd.ss2d_fused import ckpt_pitch_for
            pitch = a.pitch or ckpt_pitch_for(shape[2], shape[3], shape[0] * shape[1], core.quad_backward_ok(u, delta, Bm, Cm),
                                              core.rowlane_ok(u, delta, Bm, Cm))
            _, x = core.fwd_ext(u, delta, A, Bm, Cm, D, bias, True, ckpt_pitch=pitch)
        else:
            _, x = core.fwd(u, delta, A, Bm, Cm, D, bias, True, 1)
        fb, bb = fwd_bytes(*shape, s=es), bwd_bytes(*shape, s=es)
        geos = [(0, 0, 0)]
        if a.sweep:
            geos += [(t, w, tl) for t in (5, 10, 20) for (w, tl) in ((16, 1), (8, 1), (8, 2), (4, 4), (3, 5), (2, 8), (12, 1))]
        for items, waves, tiles in geos:
            rec = {"shape": name, "dims": shape, "dtype": a.dtype, "items": items, "waves": waves, "tiles": tiles, "ckpt_pitch": pitch}
            _capi.set_option("fwd_items", items)
            _capi.set_option("fwd_waves", waves)
            _capi.set_option("fwd_tiles", tiles)
            if a.fine:
                t = time_call(lambda: core.fwd_ext(u, delta, A, Bm, Cm, D, bias, True, ckpt_pitch=pitch), a.iters)
            else:
                t = time_call(lambda: core.fwd(u, delta, A, Bm, Cm, D, bias, True, 1), a.iters)
            rec.update(fwd_us=t * 1e6, fwd_GBs=fb / t / 1e9, fwd_frac_of_8TBs=fb / t / HBM_PEAK)
            if items in (0, 5, 10) and tiles <= 1:
                _capi.set_option("bwd_items", items)
                _capi.set_option("bwd_waves", waves)
                t = time_call(lambda: core.bwd_ext(u, delta, A, Bm, Cm, D, bias, dout, x, True, ckpt_pitch=pitch), max(3, a.iters // 2))
                rec.update(bwd_us=t * 1e6, bwd_GBs=bb / t / 1e9, bwd_frac_of_8TBs=bb / t / HBM_PEAK)
            for k in ("fwd_items", "fwd_waves", "fwd_tiles", "bwd_items", "bwd_waves"):
                _capi.set_option(k, 0)
            rows.append(rec)
            print(json.dumps(rec), flush=True)
        del u, delta, Bm, Cm, dout, x
        torch.cuda.empty_cache()
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
