#!/usr/bin/env python3
"""in_proj with the transposed x half (gemm.LinearXZFn) against the plain formulation: forward, input / weight / bias gradients."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sigma_amd import gemm
torch.manual_seed(0)
for (B, H, W, C, d) in [(2, 6, 10, 96, 192), (2, 30, 40, 384, 768), (1, 15, 20, 768, 1536), (2, 23, 40, 128, 256)]:
    x = torch.randn(B, H, W, C, device="cuda", requires_grad=True)
    w = (0.05 * torch.randn(2 * d, C, device="cuda")).requires_grad_()
    b = (0.1 * torch.randn(2 * d, device="cuda")).requires_grad_()
    ok = gemm.xz_ok(x.reshape(-1, C), w)
    print("shape", (B, H, W, C, d), "xz_ok", ok, flush=True)
    xi, z = gemm.linear_xz(x, w, b)
    ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
    rx, rz = ref[..., :d].permute(0, 3, 1, 2), ref[..., d:]
    print("  fwd err", float((xi.double() - rx).abs().max() / rx.abs().max()), float((z.double() - rz).abs().max() / rz.abs().max()), flush=True)
    gx, gz = torch.randn_like(xi), torch.randn_like(z)
    (xi * gx).sum().backward(retain_graph=True) if False else None
    loss = (xi * gx).sum() + (z * gz).sum()
    loss.backward()
    xd = x.detach().double().requires_grad_(); wd = w.detach().double().requires_grad_(); bd = b.detach().double().requires_grad_()
    r = torch.nn.functional.linear(xd, wd, bd)
    ((r[..., :d].permute(0, 3, 1, 2) * gx.double()).sum() + (r[..., d:] * gz.double()).sum()).backward()
    for name, a, bb in (("dx", x.grad, xd.grad), ("dw", w.grad, wd.grad), ("db", b.grad, bd.grad)):
        print("  %s rel err %.2e" % (name, float((a.double() - bb).abs().max() / bb.abs().max())), flush=True)
