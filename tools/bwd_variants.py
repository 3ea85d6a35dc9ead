#!/usr/bin/env python3
"""A/B of launch variants (options of sigma_scan_set_option) on the dominant shapes, forward and backward.
SIGMA_HIP_LIB=<variant .so> (python -m sigma_amd.build --variant ...) compares builds of the same ABI."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigma_amd import _capi
from sigma_amd import selective_scan_cuda_core as core
from tools.scan_bench import SHAPES, make, time_call, bwd_bytes, fwd_bytes

VARIANTS = [dict(), dict(bwd_nb=2), dict(bwd_nb=2, bwd_slab2=1), dict(bwd_waves=6), dict(bwd_waves=6, bwd_nb=2),
            dict(bwd_waves=8, bwd_nb=2, bwd_slab2=1), dict(bwd_items=5), dict(bwd_items=5, bwd_slab2=1)]
FV = [dict(), dict(fwd_prefetch=2), dict(fwd_waves=16), dict(fwd_waves=4), dict(fwd_items=20), dict(fwd_items=5)]
VARIANTS = [dict(), dict(bwd_waves=8), dict(bwd_nb=2, bwd_slab2=1), dict(bwd_items=5)]
for name in sys.argv[1:] or ["enc_s2_b16", "enc_s0_b8", "dec_s0_b8"]:
    shape = SHAPES[name]
    u, delta, A, Bm, Cm, D, bias, dout = make(shape)
    _, x = core.fwd(u, delta, A, Bm, Cm, D, bias, True, 1)
    for v in VARIANTS:
        for k, val in v.items(): _capi.set_option(k, val)
        t = min(time_call(lambda: core.bwd(u, delta, A, Bm, Cm, D, bias, dout, x, True, 1), 5) for _ in range(2))
        for k in v: _capi.set_option(k, 0)
        print(name, "bwd", v, "%.0f us %.0f GB/s" % (t * 1e6, bwd_bytes(*shape) / t / 1e9), flush=True)
    for v in FV:
        for k, val in v.items(): _capi.set_option(k, val)
        t = min(time_call(lambda: core.fwd(u, delta, A, Bm, Cm, D, bias, True, 1), 10) for _ in range(2))
        for k in v: _capi.set_option(k, 0)
        print(name, "fwd", v, "%.0f us %.0f GB/s" % (t * 1e6, fwd_bytes(*shape) / t / 1e9), flush=True)
