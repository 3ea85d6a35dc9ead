#!/usr/bin/env python3
"""Correctness sweep of the row-lane kernels (csrc/scan_fwdr.hip / scan_bwdr.hip, ckpt_pitch 16) against the CPU oracle,
one JSON line per case with the error of every output -- the development companion of tests/test_scan_gpu.py
(test_row_lane_kernels_*), which asserts the same comparisons.

    python tools/rowlane_check.py [--out file.jsonl] [--quick]
"""
import argparse
import json
import os
import sys
import traceback

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from sigma_amd import _capi  # noqa: E402
from sigma_amd import selective_scan_cuda_core as core  # noqa: E402
from oracle import scan_oracle as so  # noqa: E402


def model_like(batch, KD, L, N, G, seed=0):
    g = torch.Generator().manual_seed(seed)
    A = -torch.arange(1, N + 1, dtype=torch.float32).repeat(KD, 1) * (1 + 0.05 * torch.rand(KD, N, generator=g))
    B = torch.randn(batch, G, N, L, generator=g)
    C = torch.randn(batch, G, N, L, generator=g)
    D = 1.0 + 0.1 * torch.randn(KD, generator=g)
    tgt = torch.exp(torch.rand(KD, generator=g) * (np.log(0.1) - np.log(0.001)) + np.log(0.001))
    bias = tgt + torch.log(-torch.expm1(-tgt))
    u = torch.randn(batch, KD, L, generator=g)
    delta = 0.5 * torch.randn(batch, KD, L, generator=g)
    dout = torch.randn(batch, KD, L, generator=g)
    return u, delta, A, B, C, D, bias, dout


def run_case(shape, opts, softplus=True, with_D=True, with_bias=True, seed=23):
    batch, KD, L, N, G, mask, ush = shape
    u, delta, A, B, C, D, bias, dout = model_like(batch, KD, L, N, G, seed=seed)
    if not softplus:
        delta = delta.abs()                     # a negative step size makes the recurrence itself explode
        bias = bias.abs()
    if not with_D:
        D = None
    if not with_bias:
        bias = None
    rpg = KD // G
    keep = lambda t: torch.cat([t[:, g * rpg:(g + 1) * rpg] for g in range(0, G, 1 << ush)], dim=1).contiguous()
    full = lambda t: torch.cat([t[:, (g >> ush) * rpg:((g >> ush) + 1) * rpg] for g in range(G)], dim=1)
    u_h, g_h = keep(u), keep(dout)
    u_f, g_f = full(u_h), full(g_h)
    dev = "cuda"
    args = [None if t is None else t.to(dev) for t in (u_h, delta, A, B, C, D, bias)]
    rec = {"shape": list(shape), "opts": opts, "softplus": softplus, "D": with_D, "bias": with_bias}
    assert core.rowlane_ok(args[0], args[1], args[3], args[4]), "rowlane_ok is False"
    try:
        for k, v in opts.items():
            _capi.set_option(k, v)
        out, x = core.fwd_ext(*args, softplus, rev_mask=mask, u_gshift=ush, ckpt_pitch=16)
        out_nox, _ = core.fwd_ext(*args, softplus, rev_mask=mask, u_gshift=ush, ckpt_pitch=16, need_x=False)
        grads = core.bwd_ext(*args, g_h.to(dev), x, softplus, rev_mask=mask, u_gshift=ush, dout_gshift=ush, ckpt_pitch=16)
        torch.cuda.synchronize()
    finally:
        for k in opts:
            _capi.set_option(k, 0)
    revs = [(mask >> g) & 1 for g in range(G)]
    fr = lambda t: torch.cat([t[:, g * rpg:(g + 1) * rpg].flip(-1) if revs[g] else t[:, g * rpg:(g + 1) * rpg] for g in range(G)], 1)
    fg = lambda t: torch.stack([t[:, g].flip(-1) if revs[g] else t[:, g] for g in range(G)], 1)
    ref = fr(so.selective_scan_oracle(fr(u_f), fr(delta), A, fg(B), fg(C), D, bias, softplus, acc64=True))
    rg = list(so.selective_scan_oracle_bwd(fr(u_f), fr(delta), A, fg(B), fg(C), D, bias, fr(g_f), softplus))
    rg[0], rg[1], rg[3], rg[4] = fr(rg[0]), fr(rg[1]), fg(rg[3]), fg(rg[4])
    ok = True

    def cmp(name, got, want, rtol, atol):
        nonlocal ok
        got, want = got.float().cpu(), want.float()
        err = (got - want).abs()
        tol = atol + rtol * want.abs()
        bad = int((err > tol).sum())
        worst = int(torch.argmax(err - tol))
        rec[name] = {"max_abs": float(err.max()), "ref_max": float(want.abs().max()), "bad": bad, "n": err.numel(),
                     "worst_index": [int(i) for i in np.unravel_index(worst, tuple(err.shape))] if bad else None,
                     "nan": int(torch.isnan(got).sum())}
        ok = ok and bad == 0 and rec[name]["nan"] == 0

    cmp("out", out, ref, 6e-4, 2e-3)
    cmp("out_nox", out_nox, ref, 6e-4, 2e-3)
    for name, g, r in zip(["du", "ddelta", "dA", "dB", "dC", "dD", "dbias"], grads, rg):
        if g is None:
            continue
        cmp(name, g, r, 3e-3, 2e-3 + 2e-4 * float(r.abs().max()))
    rec["ok"] = ok
    return rec


CASES = [
    # (batch, KD, L, N, G, rev_mask, u_gshift), options
    ((2, 256, 1200, 16, 4, 0b1010, 1), {}),                  # one row block per group: dB/dC written directly
    ((2, 512, 1200, 16, 4, 0b1010, 1), {}),                  # two row blocks per group: workspace slabs + reduce
    ((1, 256, 1204, 16, 4, 0b0110, 1), {}),                  # partial last tile (L % 16 == 4), other flip pattern
    ((3, 192, 300, 8, 1, 0, 0), {}),                         # 8 states, one group, L % 16 == 12
    ((3, 192, 300, 8, 1, 1, 0), {"rl_waves": 8}),            # 8 states, eight state waves, reversed
    ((2, 384, 2564, 4, 2, 0b10, 1), {}),                     # 4 states (fusion / decoder)
    ((2, 256, 1200, 16, 4, 0b1010, 1), {"rl_segs": 3}),      # forced sequence segments
    ((2, 256, 1200, 16, 4, 0b1010, 1), {"rl_segs": 5, "rl_waves": 8}),
    ((2, 256, 1200, 16, 4, 0b1010, 1), {"rl_waves": 16}),    # forward with 16 state waves (backward keeps 8)
    ((2, 256, 1200, 16, 4, 0b1010, 1), {"rl_waves": 4}),     # forward with 4 state waves
    ((1, 256, 4800, 16, 4, 0b1010, 1), {}),                  # few rows: automatic segments
    ((1, 768, 19200, 16, 4, 0b1010, 1), {}),                 # one image per GPU, encoder stage 0
    ((11, 3072, 176, 16, 4, 0b1010, 1), {"rl_chain": 2}),    # 528 row blocks > 512 resident workgroups: chained walk
    ((11, 3072, 172, 16, 4, 0b0101, 1), {"rl_chain": 2}),    # chained walk with a partial last tile
    ((11, 3072, 176, 16, 4, 0b1010, 1), {"rl_chain": 1}),    # the same without the chain
    ((2, 64, 16, 4, 1, 0, 0), {}),                           # one tile
    ((2, 64, 8, 4, 1, 1, 0), {}),                            # less than one tile, reversed
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    rows = []
    cases = CASES[:6] if a.quick else CASES
    extra = [] if a.quick else [dict(softplus=False), dict(with_D=False, with_bias=False)]
    for shape, opts in cases:
        for kw in [dict()] + (extra if shape == CASES[0][0] and not opts else []):
            try:
                rec = run_case(shape, opts, **kw)
            except Exception as e:                                     # noqa: BLE001
                rec = {"shape": list(shape), "opts": opts, "ok": False, "error": repr(e), "trace": traceback.format_exc()[-1500:]}
            rows.append(rec)
            print(json.dumps(rec), flush=True)
    print("ALL OK" if all(r["ok"] for r in rows) else "FAILURES: %d of %d" % (sum(not r["ok"] for r in rows), len(rows)))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
