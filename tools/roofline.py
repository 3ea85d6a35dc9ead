#!/usr/bin/env python3
"""Algorithmic-byte table of Sigma's hot path emitted from code, so that the
roofline numbers in DESIGN.md / bench.py can be recomputed.

    python tools/roofline.py [--backbone sigma_small] [--height 480] [--width 640] [--batch 1]
                             [--classes 40] [--json]

Conventions:
  * scans are charged at the reference OPERATOR boundary fwd(u,delta,A,B,C,D,delta_bias)->out:
      fwd = s*(3*B*KD*L) + s*(2*B*G*N*L) + 4*(KD*N + 2*KD) [+ checkpoints 4*B*KD*ceil(L/2048)*2N]
      bwd = s*(5*B*KD*L) + s*(4*B*G*N*L)
    u, delta read once, out written once, B/C read once per GROUP;
  * every other op is charged "read inputs once + write outputs once" at the reference MODULE
    boundaries (SS2D in/out, LayerNorm, PatchMerging, fusion blocks, decoder blocks, upsamples,
    logits); fwd+bwd of those is taken as 3x the forward bytes.
Peaks: MI355X HBM3E 8.0 TB/s spec, 6.3 TB/s measured copy ceiling (MI355X_MICROARCH.md).
"""
from __future__ import annotations

import argparse
import json

SPEC_TBS = 8.0
COPY_TBS = 6.3


def scan_fwd_bytes(B, KD, L, N, G, s=4, ckpt=False):
    b = s * 3 * B * KD * L + s * 2 * B * G * N * L + 4 * (KD * N + 2 * KD)
    if ckpt:
        b += 4 * B * KD * ((L + 2047) // 2048) * 2 * N
    return b


def scan_bwd_bytes(B, KD, L, N, G, s=4):
    return s * 5 * B * KD * L + s * 4 * B * G * N * L


def stage_sizes(H, W):
    h, w = H // 4, W // 4
    out = []
    for _ in range(4):
        out.append((h, w))
        h, w = (h + 1) // 2, (w + 1) // 2
    return out


def scan_calls(backbone, H, W, batch):
    """(site, calls, (B, KD, L, N, G)) for one training step of `batch` RGB-X pairs.
    The two encoder passes run as one batch-2B pass (identical weights)."""
    E = 128 if backbone == "sigma_base" else 96
    depths = [2, 2, 9, 2] if backbone == "sigma_tiny" else [2, 2, 27, 2]
    rows = []
    for i, (h, w) in enumerate(stage_sizes(H, W)):
        C = E * 2 ** i
        d, L = 2 * C, h * w
        rows.append((f"enc s{i}", depths[i], (2 * batch, 4 * d, L, 16, 4)))
        rows.append((f"CroMB s{i}", 2, (batch, d, L, 4, 1)))
        rows.append((f"ConMB s{i}", 1, (batch, 2 * d, 2 * L, 4, 2)))
        if i < 3:
            rows.append((f"dec @s{i}", 4, (batch, 4 * d, L, 4, 4)))
    return rows


def module_bytes_fwd(backbone, H, W, batch, classes):
    """Forward bytes of everything that is not a scan, at reference module boundaries (fp32)."""
    E = 128 if backbone == "sigma_base" else 96
    depths = [2, 2, 9, 2] if backbone == "sigma_tiny" else [2, 2, 27, 2]
    f = 4
    total = {}
    enc = 0
    sizes = stage_sizes(H, W)
    enc += 2 * batch * (3 * H * W + sizes[0][0] * sizes[0][1] * E) * f                  # stem in + out
    for i, (h, w) in enumerate(sizes):
        C, L = E * 2 ** i, h * w
        enc += 2 * batch * depths[i] * (2 * L * C) * f * 2                             # VSS block: LN in/out + SS2D in/out
        if i < 3:
            enc += 2 * batch * (L * C + L * C // 2) * f                                # PatchMerging in (4C x L/4) + out
    total["encoder modules"] = enc
    fus = 0
    for i, (h, w) in enumerate(sizes):
        C, L = E * 2 ** i, h * w
        fus += batch * (2 * L * C + 2 * L * C) * f                                     # CroMB in (2) + out (2)
        fus += batch * (2 * L * C + L * C) * f                                         # ConMB in (2) + out (1)
    total["fusion modules"] = fus
    dec = 0
    h3, w3 = sizes[3]
    dec += batch * (h3 * w3 * 8 * E + 4 * h3 * w3 * 4 * E) * f                         # PatchExpand
    for i in (2, 1, 0):
        C, L = E * 2 ** i, sizes[i][0] * sizes[i][1]
        dec += batch * 4 * (2 * L * C) * f * 2                                         # 4 CVSS blocks: SS2D + conv branch
        if i > 0:
            dec += batch * (L * C + 4 * L * C // 2) * f                                # UpsampleExpand
    L0 = sizes[0][0] * sizes[0][1]
    dec += batch * (L0 * E + 4 * L0 * E + 16 * L0 * E) * f                             # FinalUpsample_X4
    dec += batch * (16 * L0 * E + H * W * classes) * f                                 # classifier
    dec += batch * (2 * H * W * classes) * f                                           # final interpolate + loss read
    total["decoder modules"] = dec
    return total


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backbone", default="sigma_small")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--classes", type=int, default=40)
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    rows = []
    tf = tb = 0
    for site, calls, shp in scan_calls(a.backbone, a.height, a.width, a.batch):
        fb, bb = scan_fwd_bytes(*shp, ckpt=True), scan_bwd_bytes(*shp)
        rows.append(dict(site=site, calls=calls, shape=shp, fwd_MB=fb / 1e6, bwd_MB=bb / 1e6,
                         fwd_us_at_8TBs=fb / SPEC_TBS / 1e6, bwd_us_at_8TBs=bb / SPEC_TBS / 1e6))
        tf += calls * fb
        tb += calls * bb
    mods = module_bytes_fwd(a.backbone, a.height, a.width, a.batch, a.classes)
    mod_f = sum(mods.values())
    step_bytes = tf + tb + 3 * mod_f
    summary = dict(backbone=a.backbone, height=a.height, width=a.width, batch=a.batch,
                   scan_fwd_GB=tf / 1e9, scan_bwd_GB=tb / 1e9, module_fwd_GB={k: v / 1e9 for k, v in mods.items()},
                   forward_GB=(tf + mod_f) / 1e9, step_GB=step_bytes / 1e9,
                   forward_ms_at_6p3=(tf + mod_f) / COPY_TBS / 1e9, step_ms_at_6p3=step_bytes / COPY_TBS / 1e9,
                   step_ms_at_8=step_bytes / SPEC_TBS / 1e9,
                   images_per_s_bound_at_6p3=a.batch / (step_bytes / COPY_TBS / 1e12),
                   images_per_s_bound_at_8=a.batch / (step_bytes / SPEC_TBS / 1e12))
    if a.json:
        print(json.dumps(dict(scans=rows, summary=summary)))
        return
    print(f"# {a.backbone} {a.height}x{a.width}, batch {a.batch} RGB-X pairs, fp32 -- algorithmic bytes")
    print(f"{'site':10s} {'calls':>5s} {'(B, KD, L, N, G)':28s} {'fwd MB':>9s} {'bwd MB':>9s} {'fwd us@8TB/s':>13s} {'bwd us@8TB/s':>13s}")
    for r in rows:
        print(f"{r['site']:10s} {r['calls']:5d} {str(r['shape']):28s} {r['fwd_MB']:9.1f} {r['bwd_MB']:9.1f} "
              f"{r['fwd_us_at_8TBs']:13.1f} {r['bwd_us_at_8TBs']:13.1f}")
    print(f"scans: fwd {tf / 1e9:.2f} GB, bwd {tb / 1e9:.2f} GB")
    for k, v in mods.items():
        print(f"{k}: fwd {v / 1e9:.2f} GB")
    print(f"forward total {summary['forward_GB']:.2f} GB -> >= {summary['forward_ms_at_6p3']:.2f} ms at 6.3 TB/s")
    print(f"step (fwd+bwd) total {summary['step_GB']:.2f} GB -> >= {summary['step_ms_at_6p3']:.2f} ms at 6.3 TB/s, "
          f">= {summary['step_ms_at_8']:.2f} ms at 8 TB/s -> <= {summary['images_per_s_bound_at_6p3']:.0f} images/s "
          f"({summary['images_per_s_bound_at_8']:.0f} at spec)")


if __name__ == "__main__":
    main()
