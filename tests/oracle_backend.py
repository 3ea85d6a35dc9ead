"""TEST-ONLY scan backend: an autograd function with the operator's signature whose forward and
backward are the CPU oracle (oracle/scan_oracle.c).  Tests inject it into the product model
with ``use_oracle_scan()`` to check the host-side model logic on a box without a GPU.  It lives
under tests/ on purpose: the product package has no CPU path and never imports this.
"""
import contextlib

import torch

from oracle import scan_oracle as so


class _OracleScan(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=False, nrows=1):
        ctx.softplus = delta_softplus
        ctx.save_for_backward(u, delta, A, B, C, D, delta_bias)
        return so.selective_scan_oracle(u, delta, A, B, C, D, delta_bias, delta_softplus, acc64=False)

    @staticmethod
    def backward(ctx, dout):
        u, delta, A, B, C, D, delta_bias = ctx.saved_tensors
        du, dd, dA, dB, dC, dD, db = so.selective_scan_oracle_bwd(u, delta, A, B, C, D, delta_bias, dout, ctx.softplus)
        return du.to(u.dtype), dd.to(delta.dtype), dA, dB.to(B.dtype), dC.to(C.dtype), dD, db, None, None


def oracle_scan_fn(u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=False, nrows=1):
    return _OracleScan.apply(u, delta, A, B, C, D, delta_bias, delta_softplus, nrows)


@contextlib.contextmanager
def use_oracle_scan():
    """Temporarily route sigma_amd's model code to the CPU oracle scan (tests only)."""
    from sigma_amd.models.encoders import vmamba
    saved = vmamba.selective_scan_fn
    vmamba.selective_scan_fn = oracle_scan_fn
    try:
        yield
    finally:
        vmamba.selective_scan_fn = saved
