"""The reference's CPU selective-scan fallback, restated in torch -- TEST / BASELINE INFRASTRUCTURE ONLY.

``selective_scan_ref`` (models/encoders/selective_scan/selective_scan/selective_scan_interface.py:86-131)
is what Sigma runs when the CUDA extension is missing: fp32 upcast, ``delta += bias``, softplus, the
dense ``deltaA = exp(delta x A)`` and ``deltaB_u`` tensors (B, D, L, N), a Python loop over the L
positions and one einsum per position.  This file restates it with the same tensor program (same
memory footprint, same op count), so that bench.py's ``cpu_baseline`` can time the reference's own
algorithm on the GPU box's host cores -- /root/reference does not exist there.  It is checked against the
C oracle in tests/test_oracle_scan.py; only tests/ and bench.py's cpu_baseline leg import it.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def selective_scan_ref(u, delta, A, B, C, D=None, delta_bias=None, delta_softplus=False):
    dtype_in = u.dtype
    u = u.float()
    delta = delta.float()
    if delta_bias is not None:
        delta = delta + delta_bias[..., None].float()
    if delta_softplus:
        delta = F.softplus(delta)
    batch, dim, dstate = u.shape[0], A.shape[0], A.shape[1]
    B = B.float()
    C = C.float()
    x = A.new_zeros((batch, dim, dstate))
    ys = []
    deltaA = torch.exp(torch.einsum("bdl,dn->bdln", delta, A))
    if B.dim() == 3:
        deltaB_u = torch.einsum("bdl,bnl,bdl->bdln", delta, B, u)
    else:
        B = B.repeat_interleave(dim // B.shape[1], dim=1)            # "B G N L -> B (G H) N L"
        deltaB_u = torch.einsum("bdl,bdnl,bdl->bdln", delta, B, u)
    if C.dim() == 4:
        C = C.repeat_interleave(dim // C.shape[1], dim=1)
    for i in range(u.shape[2]):
        x = deltaA[:, :, i] * x + deltaB_u[:, :, i]
        if C.dim() == 3:
            y = torch.einsum("bdn,bn->bd", x, C[:, :, i])
        else:
            y = torch.einsum("bdn,bdn->bd", x, C[:, :, :, i])
        ys.append(y)
    y = torch.stack(ys, dim=2)
    out = y if D is None else y + u * D[:, None]
    return out.to(dtype=dtype_in)


def time_fwd_bwd(batch, dim, seqlen, dstate, groups, threads=None, seed=0, backward=True):
    """Seconds of one forward and one autograd backward of ``selective_scan_ref`` on the host cores."""
    import time
    if threads:
        torch.set_num_threads(int(threads))
    g = torch.Generator().manual_seed(seed)
    u = torch.randn(batch, dim, seqlen, generator=g, requires_grad=True)
    delta = (0.5 * torch.randn(batch, dim, seqlen, generator=g)).requires_grad_()
    A = (-torch.arange(1, dstate + 1, dtype=torch.float32).repeat(dim, 1)).requires_grad_()
    Bm = torch.randn(batch, groups, dstate, seqlen, generator=g, requires_grad=True)
    Cm = torch.randn(batch, groups, dstate, seqlen, generator=g, requires_grad=True)
    D = torch.ones(dim, requires_grad=True)
    bias = torch.full((dim,), -4.0, requires_grad=True)
    t0 = time.perf_counter()
    out = selective_scan_ref(u, delta, A, Bm, Cm, D, bias, True)
    t1 = time.perf_counter()
    if backward:
        out.backward(torch.ones_like(out))
    t2 = time.perf_counter()
    return t1 - t0, t2 - t1, torch.get_num_threads()
