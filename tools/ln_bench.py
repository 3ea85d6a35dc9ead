#!/usr/bin/env python3
"""LayerNorm micro-benchmark: HIP (sigma_amd.layernorm) vs ATen, fwd and bwd, model shapes."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sigma_amd.layernorm import LayerNorm
from tools.scan_bench import time_call

for rows, C in [(16 * 19200, 96), (16 * 19200, 192), (16 * 4800, 384), (16 * 1200, 384), (16 * 1200, 768), (16 * 300, 1536),
                (8 * 307200, 96)]:
    ln = LayerNorm(C).cuda()
    x = torch.randn(rows, C, device="cuda", requires_grad=True)
    dy = torch.randn(rows, C, device="cuda")
    res = {}
    for name, fn in (("hip", lambda: ln(x)), ("aten", lambda: F.layer_norm(x, (C,), ln.weight, ln.bias, ln.eps))):
        tf = time_call(fn, 10)
        y = fn()
        def bwd():
            y.backward(dy, retain_graph=True)
        tb = time_call(bwd, 10)
        res[name] = (tf, tb)
    by = rows * C * 4
    print(f"rows {rows:8d} C {C:5d}  fwd hip {res['hip'][0]*1e6:7.1f} us ({2*by/res['hip'][0]/1e9:6.0f} GB/s) aten {res['aten'][0]*1e6:7.1f} us | "
          f"bwd hip {res['hip'][1]*1e6:7.1f} us ({3*by/res['hip'][1]/1e9:6.0f} GB/s) aten {res['aten'][1]*1e6:7.1f} us", flush=True)
