#!/usr/bin/env python3
"""Development harness of the second-generation backward (csrc/scan_bwd2.hip) on one MI355X:

  check  -- v2 (ckpt pitch 640 and 320, with/without register accumulation, reversed groups, shared
            u/dout rows, odd lengths) against the first-generation kernel and the CPU oracle;
  bench  -- A/B of launch variants on the dominant shapes (HIP events on the launch stream).

    python tools/bwd2_check.py check
    python tools/bwd2_check.py bench [shape ...]
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigma_amd import _capi                                    # noqa: E402
from sigma_amd import selective_scan_cuda_core as core         # noqa: E402
from tools.scan_bench import SHAPES, make, time_call, bwd_bytes   # noqa: E402

OPTS = ("bwd_gen", "bwd_items", "bwd_waves", "bwd_nb", "bwd_slab2", "bwd_rb", "bwd_sb", "bwd_touch", "bwd_wgs", "bwd_seg")


def bwd_plan(shape, pitch):
    """(items, rows, workgroups, lds_bytes, tiles or -row_blocks, nb) the library would use (current options)."""
    import ctypes
    B, KD, L, N, G = shape
    bp = _capi.BwdParams()
    fp = bp.fwd
    fp.batch, fp.dim, fp.seqlen, fp.dstate, fp.n_groups = B, KD, L, N, G
    fp.n_chunks = (L + 2047) // 2048
    fp.ckpt_pitch = pitch
    fp.x_row_stride = ((L + pitch - 1) // pitch) * N if pitch else 0
    plan = (ctypes.c_int32 * 6)()
    rc = _capi.load().sigma_scan_bwd_plan(ctypes.byref(bp), ctypes.byref(plan))
    return list(plan) if rc == 0 else None


def with_opts(opts, fn):
    for k, v in opts.items():
        _capi.set_option(k, v)
    try:
        return fn()
    finally:
        for k in opts:
            _capi.set_option(k, 0)


def model_like(batch, KD, L, N, G, seed=0, dtype=torch.float32):
    import numpy as np
    g = torch.Generator().manual_seed(seed)
    A = -torch.arange(1, N + 1, dtype=torch.float32).repeat(KD, 1) * (1 + 0.05 * torch.rand(KD, N, generator=g))
    B = torch.randn(batch, G, N, L, generator=g).to(dtype)
    C = torch.randn(batch, G, N, L, generator=g).to(dtype)
    D = 1.0 + 0.1 * torch.randn(KD, generator=g)
    tgt = torch.exp(torch.rand(KD, generator=g) * (np.log(0.1) - np.log(0.001)) + np.log(0.001))
    bias = tgt + torch.log(-torch.expm1(-tgt))
    u = torch.randn(batch, KD, L, generator=g).to(dtype)
    delta = (0.5 * torch.randn(batch, KD, L, generator=g)).to(dtype)
    dout = torch.randn(batch, KD, L, generator=g).to(dtype)
    return u, delta, A, B, C, D, bias, dout


def check():
    from oracle import scan_oracle as so
    dev = "cuda"
    assert _capi.load().sigma_scan_selftest(None) == 0, _capi.last_error()
    print("selftest ok (incl. multiplicative DPP scans)", flush=True)
    bad = 0
    cases = [
        # (batch, KD, L, N, G, rev_mask, u_gshift, dtype, oracle?)
        (2, 48, 1200, 16, 4, 0, 0, torch.float32, True),
        (2, 48, 1283, 16, 4, 0b1010, 0, torch.float32, True),
        (1, 96, 2564, 4, 2, 0b10, 0, torch.float32, True),
        (2, 96, 300, 16, 4, 0b1100, 1, torch.float32, True),
        (2, 48, 777, 8, 1, 0, 0, torch.float32, True),
        (2, 48, 1500, 16, 4, 0b0110, 0, torch.float16, True),
        (2, 48, 900, 4, 4, 0, 0, torch.bfloat16, True),
        (4, 768, 1200, 16, 4, 0b1100, 1, torch.float32, False),     # RB > 1 with 16 accumulators
        (8, 192, 4800, 4, 4, 0b1100, 1, torch.float32, False),      # RB > 1 with 4 accumulators
        (16, 3072, 1200, 16, 4, 0b1100, 1, torch.float32, False),   # the dominant launch
        (1, 768, 19200, 16, 4, 0, 0, torch.float32, False),
    ]
    sel = os.environ.get("CASES")
    if sel:
        cases = [cases[int(i)] for i in sel.split(",")]
    for (batch, KD, L, N, G, mask, ush, dt, use_oracle) in cases:
        u, delta, A, Bm, Cm, D, bias, dout = model_like(batch, KD, L, N, G, seed=L + N, dtype=dt)
        if ush:
            rpg = KD // G
            keep = torch.cat([u[:, g * rpg:(g + 1) * rpg] for g in range(0, G, 1 << ush)], dim=1).contiguous()
            u_in = keep
            u_full = torch.cat([keep[:, (g >> ush) * rpg:((g >> ush) + 1) * rpg] for g in range(G)], dim=1)
        else:
            u_in, u_full = u, u
        args = [t.to(dev) for t in (u_in, delta, A, Bm, Cm, D, bias)]
        g = dout.to(dev)
        ref = None
        if use_oracle:
            rpg = KD // G
            revs = [(mask >> gi) & 1 for gi in range(G)]
            fr = lambda t: torch.cat([t[:, gi * rpg:(gi + 1) * rpg].flip(-1) if revs[gi] else t[:, gi * rpg:(gi + 1) * rpg] for gi in range(G)], 1)
            fg = lambda t: torch.stack([t[:, gi].flip(-1) if revs[gi] else t[:, gi] for gi in range(G)], 1)
            rg = list(so.selective_scan_oracle_bwd(fr(u_full.float()), fr(delta.float()), A, fg(Bm.float()), fg(Cm.float()), D, bias,
                                                   fr(dout.float()), True))
            rg[0], rg[1], rg[3], rg[4] = fr(rg[0]), fr(rg[1]), fg(rg[3]), fg(rg[4])
            ref = rg
        verbose = os.environ.get("VERBOSE")
        def mark(msg):
            if verbose:
                torch.cuda.synchronize()
                print("   ..", msg, flush=True)
        mark("inputs on device")
        _, x1 = core.fwd_ext(*args, True, rev_mask=mask, u_gshift=ush)
        mark("fwd default pitch done")
        g1 = with_opts(dict(bwd_gen=1), lambda: core.bwd_ext(*args, g, x1, True, rev_mask=mask, u_gshift=ush))
        mark("v1 bwd done")
        variants = [(640, dict(bwd_gen=2)), (640, dict(bwd_gen=2, bwd_rb=2, bwd_slab2=2)), (640, dict(bwd_gen=2, bwd_waves=12, bwd_rb=4)),
                    (320, dict(bwd_gen=2)), (320, dict(bwd_gen=2, bwd_rb=8)),
                    (320, dict(bwd_gen=3)), (320, dict(bwd_gen=3, bwd_waves=12)), (320, dict(bwd_gen=3, bwd_rb=1)),
                    (320, dict(bwd_gen=3, bwd_waves=8, bwd_rb=2))]
        if os.environ.get("CHECK_VARIANTS"):      # JSON [[pitch, {option: value}], ...]
            variants = [(int(v[0]), dict(v[1])) for v in json.loads(os.environ["CHECK_VARIANTS"])]
        for pitch, opts in variants:
            if pitch == 160 and (dt != torch.float32 or L % 4 != 0):
                continue                          # quad-row backward: f32 IO, LDS-DMA staging (L % 4 == 0)
            out, x = core.fwd_ext(*args, True, rev_mask=mask, u_gshift=ush, ckpt_pitch=pitch)
            mark(f"fwd pitch {pitch} done")
            plan = None
            try:
                g2 = with_opts(dict(**opts), lambda: core.bwd_ext(*args, g, x, True, rev_mask=mask, u_gshift=ush, ckpt_pitch=pitch))
            except RuntimeError as e:
                print("ERR", (batch, KD, L, N, G, mask, ush, str(dt)), pitch, opts, e, flush=True)
                bad += 1
                continue
            worst = 0.0
            tol_r = 3e-3 if dt == torch.float32 else (2e-2 if dt == torch.float16 else 8e-2)
            for name, a, b in zip(["du", "ddelta", "dA", "dB", "dC", "dD", "dbias"], g2, g1):
                d = float((a.float() - b.float()).abs().max())
                s = float(b.float().abs().max()) + 1e-6
                worst = max(worst, d / s)
                if not (d / s < tol_r * 0.5):
                    print("MISMATCH v2 vs v1", name, d, s, flush=True)
                    bad += 1
            wo = 0.0
            if ref is not None:
                for name, a, b in zip(["du", "ddelta", "dA", "dB", "dC", "dD", "dbias"], g2, ref):
                    bb = b.to(dt).float() if name in ("du", "ddelta", "dB", "dC") else b
                    d = float((a.float().cpu() - bb).abs().max())
                    s = float(bb.abs().max()) + 1e-6
                    wo = max(wo, d / s)
                    if not (d / s < tol_r):
                        print("MISMATCH v2 vs oracle", name, d, s, flush=True)
                        bad += 1
            print("ok" if bad == 0 else "..", (batch, KD, L, N, G, bin(mask), ush, str(dt).split(".")[-1]), pitch, opts,
                  "max rel vs v1 %.2e" % worst, ("vs oracle %.2e" % wo) if ref is not None else "", flush=True)
    print("CHECK", "PASSED" if bad == 0 else f"FAILED ({bad})", flush=True)
    return bad


def bench(names):
    rows = []
    for name in names or ["enc_s2_b16", "enc_s0_b16", "enc_s1_b16", "enc_s3_b16", "dec_s0_b8", "dec_s1_b8", "dec_s2_b8", "conmb_s0_b8", "cromb_s0_b8", "enc_s0", "enc_s2", "dec_s0"]:
        shape = SHAPES[name]
        u, delta, A, Bm, Cm, D, bias, dout = make(shape)
        bb = bwd_bytes(*shape)
        xs = {}
        for pitch in (640, 320, 160):
            xs[pitch] = core.fwd_ext(u, delta, A, Bm, Cm, D, bias, True, ckpt_pitch=pitch)[1]
        variants = [
            ("v1 fine", 640, dict(bwd_gen=1)),
            ("T10 auto", 640, dict()),
            ("T10 rb1", 640, dict(bwd_rb=1)),
            ("T10 rb2", 640, dict(bwd_rb=2)),
            ("T5 auto", 320, dict()),
            ("T5 rb1", 320, dict(bwd_rb=1, bwd_gen=2)),
            ("T5 v2 auto", 320, dict(bwd_gen=2)),
        ]
        if os.environ.get("VARIANTS"):       # JSON [[label, pitch, {option: value}], ...]
            variants = [tuple(v) for v in json.loads(os.environ["VARIANTS"])]
        for label, pitch, opts in variants:
            x = xs[pitch]
            try:
                t = with_opts(opts, lambda: min(time_call(lambda: core.bwd_ext(u, delta, A, Bm, Cm, D, bias, dout, x, True, ckpt_pitch=pitch), 5) for _ in range(2)))
            except RuntimeError as e:
                print(name, label, "ERR", e, flush=True)
                continue
            rec = dict(shape=name, dims=shape, variant=label, us=round(t * 1e6, 1), GBs=round(bb / t / 1e9, 1), frac8=round(bb / t / 8e12, 4),
                       plan=with_opts(opts, lambda: bwd_plan(shape, pitch)))
            rows.append(rec)
            print(json.dumps(rec), flush=True)
        del u, delta, Bm, Cm, dout, xs
        torch.cuda.empty_cache()
    return rows


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "check"
    if mode == "check":
        sys.exit(1 if check() else 0)
    rows = bench(sys.argv[2:])
    out = os.environ.get("BWD2_OUT")
    if out:
        with open(out, "w") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")
