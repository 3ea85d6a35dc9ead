"""Pin the CPU oracle (oracle/scan_oracle.c) to the reference's own outputs.

Golden vectors come from the reference's ``selective_scan_ref`` + torch autograd
(tests/golden/make_golden_scan.py, run in the build container).  Tolerances are
those of the reference's own unit test
(models/encoders/selective_scan/test_selective_scan.py:148-151, 216-224) for the
low-precision cases and much tighter for fp32, where oracle and reference differ
only by fp32 summation order.
"""
import ast
import glob
import os

import numpy as np
import pytest
import torch

from oracle import scan_oracle as so

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FILES = sorted(glob.glob(os.path.join(GOLDEN, "scan_*.npz")))


def load_case(path):
    z = np.load(path, allow_pickle=False)
    meta = ast.literal_eval(str(z["meta"]))
    dt = getattr(torch, meta["dtype"])
    t = {}
    for k in z.files:
        if k == "meta":
            continue
        v = torch.from_numpy(z[k])
        if k in ("in_u", "in_delta", "in_B", "in_C", "in_dout", "out"):
            v = v.to(dt)
        t[k] = v
    return meta, t


def test_golden_files_present():
    assert len(FILES) >= 7, "golden scan fixtures missing (tests/golden/make_golden_scan.py)"


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[5:-4] for p in FILES])
@pytest.mark.parametrize("acc64", [False, True])
def test_oracle_forward_matches_reference(path, acc64):
    meta, t = load_case(path)
    out = so.selective_scan_oracle(t["in_u"], t["in_delta"], t["in_A"], t["in_B"], t["in_C"],
                                   t.get("in_D"), t.get("in_delta_bias"), meta["softplus"], acc64=acc64)
    ref = t["out"]
    assert out.dtype == ref.dtype and out.shape == ref.shape
    if meta["dtype"] == "float32":
        rtol, atol = 2e-5, 2e-5 * float(ref.abs().max())
    elif meta["dtype"] == "float16":
        rtol, atol = 3e-3, 5e-3
    else:
        rtol, atol = 3e-2, 5e-2
    torch.testing.assert_close(out.float(), ref.float(), rtol=rtol, atol=atol)


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[5:-4] for p in FILES])
def test_oracle_backward_matches_reference(path):
    meta, t = load_case(path)
    grads = so.selective_scan_oracle_bwd(t["in_u"], t["in_delta"], t["in_A"], t["in_B"], t["in_C"],
                                         t.get("in_D"), t.get("in_delta_bias"), t["in_dout"], meta["softplus"])
    names = ["u", "delta", "A", "B", "C", "D", "delta_bias"]
    for name, g in zip(names, grads):
        key = "grad_" + name
        if key not in t:
            assert g is None
            continue
        ref = t[key].float()
        assert g.shape == ref.shape, name
        scale = float(ref.abs().max()) + 1e-6
        # reference grads were accumulated in fp32 by autograd over L steps; the oracle is fp64
        rtol, atol = 2e-4, 2e-5 * scale
        if meta["dtype"] != "float32" and name in ("u", "delta", "B", "C"):
            # the reference returns these grads rounded to the io dtype
            eps = 2.0 ** -10 if meta["dtype"] == "float16" else 2.0 ** -7
            rtol, atol = 2 * eps, 2 * eps * scale * 1e-2 + 1e-3
        torch.testing.assert_close(g, ref, rtol=rtol, atol=atol, msg=lambda m: f"{name}: {m}")


def test_oracle_edge_cases():
    # empty sequence, single element, N = 1, group count == dim
    A = -torch.rand(4, 1)
    u = torch.randn(1, 4, 0)
    out = so.selective_scan_oracle(u, u.clone(), A, torch.randn(1, 4, 1, 0), torch.randn(1, 4, 1, 0))
    assert out.shape == (1, 4, 0)
    u = torch.randn(2, 4, 1)
    d = torch.rand(2, 4, 1)
    Bm, Cm = torch.randn(2, 4, 1, 1), torch.randn(2, 4, 1, 1)
    out = so.selective_scan_oracle(u, d, A, Bm, Cm)
    # x = delta*B*u ; y = x*C  (single step)
    exp = (d * Bm[:, :, 0] * u) * Cm[:, :, 0]
    torch.testing.assert_close(out, exp, rtol=1e-6, atol=1e-6)


def test_oracle_linearity_in_u_and_chunk_independence():
    """Size-independent properties: linear in u (for fixed delta) and invariant to where
    the 2048-element checkpoints fall (compare one 4100-long scan with a manual restart)."""
    torch.manual_seed(1)
    Bsz, Dm, N, G, L = 1, 6, 4, 2, 4100
    A = -torch.rand(Dm, N)
    u1, u2 = torch.randn(Bsz, Dm, L), torch.randn(Bsz, Dm, L)
    d = 0.3 * torch.rand(Bsz, Dm, L)
    Bm, Cm = torch.randn(Bsz, G, N, L), torch.randn(Bsz, G, N, L)
    f = lambda uu: so.selective_scan_oracle(uu, d, A, Bm, Cm, None, None, True, acc64=True)
    torch.testing.assert_close(f(u1 + 2 * u2), f(u1) + 2 * f(u2), rtol=1e-4, atol=1e-4)
    # restart property: state after l0 summarises the past
    l0 = 2048
    full, _ = so.selective_scan_oracle(u1, d, A, Bm, Cm, None, None, True, acc64=True, return_last_state=True)
    head, st = so.selective_scan_oracle(u1[..., :l0], d[..., :l0], A, Bm[..., :l0], Cm[..., :l0], None, None,
                                        True, acc64=True, return_last_state=True)
    torch.testing.assert_close(full[..., :l0], head, rtol=0, atol=0)
    # replay the tail by hand from the saved state
    dl = torch.nn.functional.softplus(d[..., l0:].double())
    x = st.double().clone()
    ys = []
    rows_per_group = Dm // G
    for i in range(L - l0):
        for dd in range(Dm):
            g = dd // rows_per_group
            a = torch.exp(dl[0, dd, i] * A[dd].double())
            x[0, dd] = a * x[0, dd] + dl[0, dd, i] * Bm[0, g, :, l0 + i].double() * u1[0, dd, l0 + i].double()
        ys.append(torch.stack([(x[0, dd] * Cm[0, dd // rows_per_group, :, l0 + i].double()).sum() for dd in range(Dm)]))
    tail = torch.stack(ys, dim=1).unsqueeze(0).float()
    torch.testing.assert_close(full[..., l0:], tail, rtol=1e-4, atol=1e-4)


def test_torch_restatement_of_the_reference_cpu_fallback_matches_the_c_oracle():
    """oracle/scan_ref_torch.py (the tensor program of selective_scan_interface.py:86-131, timed by
    bench.py's cpu_baseline) against the C oracle, grouped and 3-D B/C, values and gradients."""
    import torch
    from oracle import scan_oracle as so
    from oracle import scan_ref_torch as rt
    g = torch.Generator().manual_seed(3)
    for groups in (0, 2):
        bsz, dim, L, N = 2, 12, 150, 8
        u = torch.randn(bsz, dim, L, generator=g)
        d = 0.5 * torch.rand(bsz, dim, L, generator=g)
        A = -0.5 * torch.rand(dim, N, generator=g)
        shp = (bsz, N, L) if groups == 0 else (bsz, groups, N, L)
        B, C = torch.randn(*shp, generator=g), torch.randn(*shp, generator=g)
        D, bias = torch.randn(dim, generator=g), 0.5 * torch.rand(dim, generator=g)
        dout = torch.randn(bsz, dim, L, generator=g)
        leaves = [t.clone().requires_grad_() for t in (u, d, A, B, C, D, bias)]
        out = rt.selective_scan_ref(*leaves, True)
        out.backward(dout)
        ref = so.selective_scan_oracle(u, d, A, B, C, D, bias, True, acc64=True)
        torch.testing.assert_close(out.detach(), ref, rtol=1e-4, atol=1e-4)
        rg = so.selective_scan_oracle_bwd(u, d, A, B if groups else B.unsqueeze(1), C if groups else C.unsqueeze(1), D, bias, dout, True)
        for t, r in zip(leaves, rg):
            r = r.reshape(t.shape)
            torch.testing.assert_close(t.grad, r, rtol=2e-3, atol=1e-3 + 1e-4 * float(r.abs().max()))
