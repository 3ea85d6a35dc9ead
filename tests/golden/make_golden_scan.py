#!/usr/bin/env python3
"""Generate operator-level golden vectors from the REFERENCE's own CPU code.

Run in the build container only (needs /root/reference, read-only):

    python tests/golden/make_golden_scan.py

It loads the reference file
    /root/reference/models/encoders/selective_scan/selective_scan/selective_scan_interface.py
with the CUDA extension import stubbed out (``selective_scan_cuda_core`` is only
used by the autograd wrapper, not by ``selective_scan_ref``), calls the
reference's ``selective_scan_ref`` (lines 86-131) on seeded inputs drawn from the
distributions of the reference's unit test
(models/encoders/selective_scan/test_selective_scan.py:153-179) and back-propagates
a seeded ``dout`` through it with torch autograd -- exactly what that test uses
as its oracle (:186-201).  Inputs and outputs are stored in tests/golden/scan_*.npz.

Nothing on the GPU box reads /root/reference; the tests only read the .npz files.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/models/encoders/selective_scan/selective_scan/selective_scan_interface.py"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_reference_interface():
    stub = types.ModuleType("selective_scan_cuda_core")  # never called by selective_scan_ref
    sys.modules.setdefault("selective_scan_cuda_core", stub)
    spec = importlib.util.spec_from_file_location("ref_selective_scan_interface", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


CASES = [
    # name, batch, dim, dstate, seqlen, groups(0 => 3-D B/C), has_D, has_bias, softplus, dtype, modelish
    dict(name="t64_g0_plain", batch=2, dim=24, dstate=8, seqlen=64, groups=0, has_D=False, has_bias=False,
         softplus=False, dtype="float32", modelish=False),
    dict(name="t372_g2_full", batch=2, dim=24, dstate=8, seqlen=372, groups=2, has_D=True, has_bias=True,
         softplus=True, dtype="float32", modelish=False),
    dict(name="t128_g0_fp16", batch=2, dim=24, dstate=8, seqlen=128, groups=0, has_D=True, has_bias=True,
         softplus=True, dtype="float16", modelish=False),
    dict(name="t256_g2_bf16", batch=2, dim=24, dstate=8, seqlen=256, groups=2, has_D=True, has_bias=False,
         softplus=True, dtype="bfloat16", modelish=False),
    dict(name="t1134_g1_nosp", batch=1, dim=12, dstate=8, seqlen=1134, groups=1, has_D=True, has_bias=True,
         softplus=False, dtype="float32", modelish=False),
    # crosses the 2048-element checkpoint boundary; model-like magnitudes (dt bias ~ softplus^-1(1e-3..1e-1),
    # A = -(1..N), D = 1) as initialised by vmamba.py:729-782
    dict(name="m2100_g2_n4", batch=1, dim=8, dstate=4, seqlen=2100, groups=2, has_D=True, has_bias=True,
         softplus=True, dtype="float32", modelish=True),
    dict(name="m300_g4_n16", batch=1, dim=16, dstate=16, seqlen=300, groups=4, has_D=True, has_bias=True,
         softplus=True, dtype="float32", modelish=True),
]


def make_inputs(c):
    """Seeded CPU inputs; creation order follows test_selective_scan.py:153-179."""
    torch.random.manual_seed(0)
    dt = getattr(torch, c["dtype"])
    b, d, n, L, g = c["batch"], c["dim"], c["dstate"], c["seqlen"], c["groups"]
    if c["modelish"]:
        A = -torch.arange(1, n + 1, dtype=torch.float32).repeat(d, 1) * (1.0 + 0.1 * torch.rand(d, n))
    else:
        A = -0.5 * torch.rand(d, n, dtype=torch.float32)
    bshape = (b, n, L) if g == 0 else (b, g, n, L)
    B = torch.randn(*bshape, dtype=torch.float32).to(dt)
    C = torch.randn(*bshape, dtype=torch.float32).to(dt)
    D = torch.randn(d, dtype=torch.float32) if c["has_D"] else None
    if c["has_bias"]:
        if c["modelish"]:
            tgt = torch.exp(torch.rand(d) * (np.log(0.1) - np.log(0.001)) + np.log(0.001)).clamp(min=1e-4)
            delta_bias = tgt + torch.log(-torch.expm1(-tgt))
        else:
            delta_bias = 0.5 * torch.rand(d, dtype=torch.float32)
    else:
        delta_bias = None
    u = torch.randn(b, d, L, dtype=torch.float32).to(dt)
    if c["modelish"]:
        delta = (0.5 * torch.randn(b, d, L, dtype=torch.float32)).to(dt)
    else:
        delta = (0.5 * torch.rand(b, d, L, dtype=torch.float32)).to(dt)
    dout = torch.randn(b, d, L, dtype=torch.float32).to(dt)
    return A, B, C, D, delta_bias, u, delta, dout


def main():
    ref = load_reference_interface()
    for c in CASES:
        A, B, C, D, delta_bias, u, delta, dout = make_inputs(c)
        leaves = {}
        for k, v in dict(A=A, B=B, C=C, D=D, delta_bias=delta_bias, u=u, delta=delta).items():
            leaves[k] = None if v is None else v.detach().clone().requires_grad_()
        out = ref.selective_scan_ref(leaves["u"], leaves["delta"], leaves["A"], leaves["B"], leaves["C"],
                                     leaves["D"], delta_bias=leaves["delta_bias"],
                                     delta_softplus=c["softplus"])
        out.backward(dout)
        blob = {"meta": np.array(repr(c))}
        for k, v in dict(A=A, B=B, C=C, D=D, delta_bias=delta_bias, u=u, delta=delta, dout=dout).items():
            if v is not None:
                blob["in_" + k] = v.float().numpy()
        blob["out"] = out.detach().float().numpy()
        for k, v in leaves.items():
            if v is not None:
                blob["grad_" + k] = v.grad.float().numpy()
        path = os.path.join(OUT, f"scan_{c['name']}.npz")
        np.savez_compressed(path, **blob)
        print(f"{path}: out {tuple(out.shape)} max|out|={out.abs().max().item():.4f} "
              f"{os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
