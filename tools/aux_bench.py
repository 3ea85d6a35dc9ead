#!/usr/bin/env python3
"""Roofline table of this repository's own layout / stencil / normalisation kernels (VERDICT r2 "next" #5): each is
timed at the shapes of the sigma_small training step (batch 8: 16 encoder images, 8 decoder images) and priced against
the bytes of `read every input once + write every output once` at 6.3 TB/s (the measured copy ceiling) and 8 TB/s.

    python tools/aux_bench.py [--iters 20] [--out file.jsonl]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigma_amd import ss2d_fused as sf  # noqa: E402
from sigma_amd.layernorm import LayerNorm  # noqa: E402
from sigma_amd.models.decoders.MambaDecoder import _Up2xFn  # noqa: E402
from tools.scan_bench import time_call  # noqa: E402

# (tag, images, d_inner, H, W): encoder stage 0 / 2 (2 x 8 images in one pass), decoder level at 120x160 (8 images)
STAGES = [("enc_s0", 16, 192, 120, 160), ("enc_s2", 16, 768, 30, 40), ("dec_s0", 8, 192, 120, 160)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    dev = "cuda"
    rows = []

    def rec(kernel, tag, nbytes, fn):
        t = time_call(fn, a.iters)
        r = dict(kernel=kernel, shape=tag, us=round(t * 1e6, 1), MB=round(nbytes / 1e6, 1), GBs=round(nbytes / t / 1e9, 1),
                 frac_6300=round(nbytes / t / 6.3e12, 3), frac_8000=round(nbytes / t / 8e12, 3))
        rows.append(r)
        print(json.dumps(r), flush=True)

    for tag, B, d, H, W in STAGES:
        L = H * W
        T = B * d * L * 4                       # bytes of one (B, d, H, W) fp32 tensor
        g = torch.Generator().manual_seed(0)
        x = torch.randn(B, d, H, W, generator=g).to(dev)
        w = (0.3 * torch.randn(d, 1, 3, 3, generator=g)).to(dev)
        b = torch.zeros(d, device=dev)
        # depthwise conv + SiLU + two orders: 1 read + 2 writes; backward: g2 (2 reads) + x -> gpre (write), gpre -> dx: 3R + 1W, 1R + 1W
        rec("dwconv_silu_two_orders fwd", tag, 3 * T, lambda: sf.dwconv_silu_two_orders(x, w, b))
        xg = x.clone().requires_grad_()
        wg, bg = w.clone().requires_grad_(), b.clone().requires_grad_()
        out2 = sf.dwconv_silu_two_orders(xg, wg, bg)
        g2 = torch.randn_like(out2)
        rec("dwconv_silu_two_orders bwd (2 kernels)", tag, 6 * T, lambda: torch.autograd.grad(out2, (xg, wg, bg), g2, retain_graph=True))
        ys = torch.randn(B, 4, d, L, device=dev)
        rec("cross_merge_nhwc", tag, 5 * T, lambda: sf.cross_merge_nhwc(ys, H, W))
        dy = torch.randn(B, H, W, d, device=dev)
        rec("cross_split_nhwc", tag, 3 * T, lambda: sf.cross_split_nhwc(dy))
        du = torch.randn(B, 4 * d, L, device=dev)
        acc = torch.randn(B, 2, d, L, device=dev)
        rec("pair_sum_add", tag, 4 * T + 2 * T, lambda: sf._pair_sum_add(du, acc, B * 2, d * L))
        xz = torch.randn(B, H, W, 2 * d, device=dev, requires_grad=True)
        rec("split_xz fwd (transpose2d)", tag, 2 * T, lambda: sf.split_xz(xz))
        x1, z1 = sf.split_xz(xz)
        gx, gz = torch.randn_like(x1), torch.randn(B, H, W, d, device=dev)
        rec("split_xz bwd (transpose2d + copy)", tag, 4 * T, lambda: torch.autograd.grad((x1, z1), xz, (gx, gz), retain_graph=True))
        # LayerNorm over d (out_norm, gated) and over C = d / 2 (block norm)
        for C, gated in ((d, True), (d // 2, False)):
            ln = LayerNorm(C).to(dev)
            xin = torch.randn(B, H, W, C, device=dev, requires_grad=True)
            TC = B * L * C * 4
            if gated:
                zz = torch.randn(B, H, W, 2 * C, device=dev, requires_grad=True)
                zv = zz[..., C:]
                rec(f"layernorm gated fwd C={C}", tag, 3 * TC, lambda: ln.forward_gated(xin, zv))
                yo = ln.forward_gated(xin, zv)
                gy = torch.randn_like(yo)
                rec(f"layernorm gated bwd C={C}", tag, 5 * TC, lambda: torch.autograd.grad(yo, (xin, zz, ln.weight, ln.bias), gy, retain_graph=True))
            else:
                rec(f"layernorm fwd C={C}", tag, 2 * TC, lambda: ln(xin))
                yo = ln(xin)
                gy = torch.randn_like(yo)
                rec(f"layernorm bwd C={C}", tag, 3 * TC, lambda: torch.autograd.grad(yo, (xin, ln.weight, ln.bias), gy, retain_graph=True))
        del x, ys, dy, du, acc, xz, x1, z1
        torch.cuda.empty_cache()
    # bilinear x2 of the decoder's last two up-samplings
    for (B, H, W, C) in ((8, 120, 160, 96), (8, 240, 320, 96)):
        xin = torch.randn(B, H, W, C, device=dev, requires_grad=True)
        Tn = B * H * W * C * 4
        rec("upsample2x fwd", f"{B}x{H}x{W}x{C}", 5 * Tn, lambda: _Up2xFn.apply(xin))
        yo = _Up2xFn.apply(xin)
        gy = torch.randn_like(yo)
        rec("upsample2x bwd", f"{B}x{H}x{W}x{C}", 5 * Tn, lambda: torch.autograd.grad(yo, xin, gy, retain_graph=True))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
