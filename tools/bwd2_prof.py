#!/usr/bin/env python3
"""Per-phase cycle shares of the second-generation backward (needs the profiling build:
python -m sigma_amd.build --variant prof --flags=-DSIGMA_BWD2_PROF=1; SIGMA_HIP_LIB=sigma_amd/lib/libsigma_hip_prof.so)."""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigma_amd import _capi                                    # noqa: E402
from sigma_amd import selective_scan_cuda_core as core         # noqa: E402
from tools.scan_bench import SHAPES, make, time_call           # noqa: E402
from tools.bwd2_check import with_opts, bwd_plan               # noqa: E402

PHASES = ["setup", "row prologue", "stage issue", "forward", "rev fold+scan", "rev replay+slab", "barrier", "column sum",
          "row epilogue", "acc flush"]


def read():
    buf = (ctypes.c_uint64 * 16)()
    _capi.check(_capi.load().sigma_scan_debug_read(ctypes.byref(buf)), "debug_read")
    return list(buf)


def main():
    variants = [("T10 rb1", 640, dict(bwd_gen=2, bwd_rb=1)), ("T10 auto", 640, dict(bwd_gen=2)),
                ("T10 R16", 640, dict(bwd_gen=2, bwd_waves=16, bwd_nb=2)), ("T5 R16", 320, dict(bwd_gen=2, bwd_waves=16)),
                ("T5 R8 rb1", 320, dict(bwd_gen=2, bwd_waves=8, bwd_rb=1))]
    if os.environ.get("VARIANTS"):       # JSON [[label, pitch, {option: value}], ...]
        variants = [tuple(v) for v in json.loads(os.environ["VARIANTS"])]
    for name in sys.argv[1:] or ["enc_s2_b16", "dec_s0_b8"]:
        shape = SHAPES[name]
        u, delta, A, Bm, Cm, D, bias, dout = make(shape)
        for label, pitch, opts in variants:
            x = core.fwd_ext(u, delta, A, Bm, Cm, D, bias, True, ckpt_pitch=pitch)[1]
            read()
            t = with_opts(opts, lambda: time_call(lambda: core.bwd_ext(u, delta, A, Bm, Cm, D, bias, dout, x, True, ckpt_pitch=pitch), 3, warmup=0))
            c = read()
            tot = sum(c[:10]) or 1
            waves = c[15] or 1
            print(json.dumps(dict(shape=name, variant=label, us=round(t * 1e6, 1), plan=with_opts(opts, lambda: bwd_plan(shape, pitch)),
                                  cycles_per_wave=round(tot / waves), shares={p: round(v / tot, 3) for p, v in zip(PHASES, c[:10])})), flush=True)


if __name__ == "__main__":
    main()
