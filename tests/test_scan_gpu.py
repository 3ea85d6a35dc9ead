"""GPU parity tests of the HIP selective-scan operator (run with -m gpu on an MI355X).

Every call goes python -> ctypes -> C ABI (include/sigma_scan.h) -> HIP kernel.  The
checker is the CPU oracle (oracle/scan_oracle.c) and the committed golden vectors that
were produced by the reference's own selective_scan_ref + autograd.

Tolerances are the reference's (models/encoders/selective_scan/test_selective_scan.py:148-151,
216-224): fwd fp32 rtol 6e-4 / atol 2e-3, fp16 3e-3 / 5e-3, bf16 3e-2 / 5e-2,
weights 1e-3 / 1e-3, with the same 2x / 5x-10x multipliers on the gradients.
"""
import ast
import glob
import itertools
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FILES = sorted(glob.glob(os.path.join(GOLDEN, "scan_*.npz")))


def _tols(dtype):
    if dtype == torch.float32:
        return 6e-4, 2e-3
    if dtype == torch.float16:
        return 3e-3, 5e-3
    return 3e-2, 5e-2


def _core():
    from sigma_amd import selective_scan_cuda_core as core
    return core


def _fn():
    from sigma_amd.selective_scan import selective_scan_fn
    return selective_scan_fn


def _oracle():
    from oracle import scan_oracle as so
    return so


# Per-gradient tolerances of the reference's unit test in fp32 (test_selective_scan.py:216-224: du 2x, ddelta 5x / 10x the
# forward's rtol / atol, dB / dC the forward's, dA 1e-3 / 5e-3, dD / ddelta_bias 1e-3 / 1e-3).  The per-row parameter
# gradients (dA, dD, ddelta_bias) are fp32 sums over the whole sequence, whose absolute rounding error scales with the
# magnitude of the result: they alone keep a floor of 2e-4 of the tensor's largest entry.
GRAD_TOLS = {"u": (1.2e-3, 4e-3), "delta": (3e-3, 2e-2), "A": (1e-3, 5e-3), "B": (6e-4, 2e-3), "C": (6e-4, 2e-3),
             "D": (1e-3, 1e-3), "bias": (1e-3, 1e-3)}


def assert_grads_close(grads, refs):
    """grads / refs in the operator's order (u, delta, A, B, C, D, delta_bias); None entries must match"""
    for name, g, r in zip(["u", "delta", "A", "B", "C", "D", "bias"], grads, refs):
        if r is None:
            assert g is None, f"d{name}: expected no gradient"
            continue
        rt, at = GRAD_TOLS[name]
        if name in ("A", "D", "bias"):
            at = max(at, 2e-4 * float(r.abs().max()))
        torch.testing.assert_close(g.detach().float().cpu(), r.float().cpu(), rtol=rt, atol=at, msg=lambda m, name=name: f"d{name}: {m}")


def test_native_library_and_wave_primitives():
    from sigma_amd import _capi
    lib = _capi.load()
    assert torch.cuda.is_available()
    rc = lib.sigma_scan_selftest(None)
    assert rc == 0, _capi.last_error()


@pytest.mark.parametrize("opts", [{}, {"rl_segs": 1}, {"rl_segs": 2}, {"rl_waves": 8}], ids=["auto", "one-segment", "two-segments", "8-fwd-waves"])
def test_row_lane_load_time_self_test(opts):
    """sigma_scan_rowlane_selftest (include/sigma_scan.h, ABI 9): the row-lane kernels against a host recurrence in
    double precision on a small problem, under the planner's choice and with the segment / wave geometry forced; the
    binding runs it once per device before the first ckpt_pitch-16 launch (selective_scan_cuda_core.rowlane_selftest)."""
    from sigma_amd import _capi
    lib = _capi.load()
    try:
        for k, v in opts.items():
            _capi.set_option(k, v)
        rc = lib.sigma_scan_rowlane_selftest(None)
        assert rc == 0, _capi.last_error()
    finally:
        for k in opts:
            _capi.set_option(k, 0)
    _core().rowlane_selftest("cuda:0")                    # idempotent


def _load_golden(path):
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


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[5:-4] for p in FILES])
def test_against_reference_golden_vectors(path):
    """HIP fwd + all seven grads vs outputs of the REFERENCE's own CPU code."""
    meta, t = _load_golden(path)
    dt = getattr(torch, meta["dtype"])
    rtol, atol = _tols(dt)
    dev = "cuda"
    leaves = {}
    for k in ("u", "delta", "A", "B", "C", "D", "delta_bias"):
        v = t.get("in_" + k)
        leaves[k] = None if v is None else v.to(dev).requires_grad_()
    out = _fn()(leaves["u"], leaves["delta"], leaves["A"], leaves["B"], leaves["C"], leaves["D"],
                leaves["delta_bias"], meta["softplus"], 1)
    assert out.dtype == dt
    torch.testing.assert_close(out.float().cpu(), t["out"].float(), rtol=rtol, atol=atol)
    out.backward(t["in_dout"].to(dev))
    mult = {"u": (2, 2), "delta": (5, 10), "B": (1, 1), "C": (1, 1)}
    for k in ("u", "delta", "B", "C"):
        rm, am = mult[k]
        torch.testing.assert_close(leaves[k].grad.float().cpu(), t["grad_" + k].float(), rtol=rtol * rm,
                                   atol=atol * am, msg=lambda m, k=k: f"d{k}: {m}")
    torch.testing.assert_close(leaves["A"].grad.cpu(), t["grad_A"], rtol=1e-3, atol=5e-3)
    if leaves["D"] is not None:
        torch.testing.assert_close(leaves["D"].grad.cpu(), t["grad_D"], rtol=1e-3, atol=1e-3)
    if leaves["delta_bias"] is not None:
        torch.testing.assert_close(leaves["delta_bias"].grad.cpu(), t["grad_delta_bias"], rtol=1e-3, atol=1e-3)


def _ref_test_inputs(seqlen, itype, groups, has_D, has_bias, batch=2, dim=24, dstate=8, seed=0):
    """Input distributions of the reference unit test (test_selective_scan.py:153-179)."""
    g = torch.Generator().manual_seed(seed)
    A = -0.5 * torch.rand(dim, dstate, generator=g)
    shp = (batch, dstate, seqlen) if groups == 0 else (batch, groups, dstate, seqlen)
    B = torch.randn(*shp, generator=g).to(itype)
    C = torch.randn(*shp, generator=g).to(itype)
    D = torch.randn(dim, generator=g) if has_D else None
    bias = 0.5 * torch.rand(dim, generator=g) if has_bias else None
    u = torch.randn(batch, dim, seqlen, generator=g).to(itype)
    delta = (0.5 * torch.rand(batch, dim, seqlen, generator=g)).to(itype)
    dout = torch.randn(batch, dim, seqlen, generator=g).to(itype)
    return u, delta, A, B, C, D, bias, dout


def _run_hip(u, delta, A, B, C, D, bias, dout, softplus, nrows):
    dev = "cuda"
    L = {k: (None if v is None else v.to(dev).requires_grad_()) for k, v in
         dict(u=u, delta=delta, A=A, B=B, C=C, D=D, bias=bias).items()}
    out = _fn()(L["u"], L["delta"], L["A"], L["B"], L["C"], L["D"], L["bias"], softplus, nrows)
    out.backward(dout.to(dev))
    grads = [None if L[k] is None else L[k].grad for k in ("u", "delta", "A", "B", "C", "D", "bias")]
    return out, grads


def _compare(out, grads, u, delta, A, B, C, D, bias, dout, softplus, itype):
    so = _oracle()
    rtol, atol = _tols(itype)
    ref = so.selective_scan_oracle(u, delta, A, B, C, D, bias, softplus, acc64=True)
    torch.testing.assert_close(out.float().cpu(), ref.float(), rtol=rtol, atol=atol, msg=lambda m: f"out: {m}")
    rg = so.selective_scan_oracle_bwd(u, delta, A, B, C, D, bias, dout, softplus)
    names = ["u", "delta", "A", "B", "C", "D", "delta_bias"]
    tol = {"u": (rtol * 2, atol * 2), "delta": (rtol * 5, atol * 10), "A": (1e-3, 5e-3), "B": (rtol, atol),
           "C": (rtol, atol), "D": (1e-3, 1e-3), "delta_bias": (1e-3, 1e-3)}
    for name, g, r in zip(names, grads, rg):
        if r is None:
            assert g is None
            continue
        rr = r.to(itype).float() if name in ("u", "delta", "B", "C") else r
        rt, at = tol[name]
        # long fp32 reductions scale their absolute error with the magnitude of the result
        at = max(at, 2e-4 * float(rr.abs().max())) if name in ("A", "D", "delta_bias") else at
        torch.testing.assert_close(g.float().cpu(), rr, rtol=rt, atol=at, msg=lambda m, name=name: f"d{name}: {m}")


# The reference's own parametrisation, complete (test_selective_scan.py:137-144): 3 dtypes x 10 lengths x
# {bias} x {softplus} x {D} x varBC_groups {1, 2} x nrows {1, 2, 3, 4} = 1920 cases.  varBC_groups = 1 is the
# 3-D B/C form (reference :159-168), written 0 here.
GRID = list(itertools.product(
    [64, 128, 256, 372, 512, 784, 1024, 1134, 2048, 4096],       # seqlen (test_selective_scan.py:139)
    [torch.float32, torch.float16, torch.bfloat16],
    list(itertools.product([False, True], repeat=3)),             # has_delta_bias, delta_softplus, has_D
    [0, 2],                                                        # varBC groups (0 => 3-D B/C)
    [1, 2, 3, 4],                                                  # nrows
))


def _grid_id(g):
    return f"L{g[0]}-{str(g[1]).split('.')[-1]}-b{int(g[2][0])}s{int(g[2][1])}d{int(g[2][2])}-g{g[3]}-r{g[4]}"


@pytest.mark.parametrize("seqlen,itype,flags,groups,nrows", GRID, ids=[_grid_id(g) for g in GRID])
def test_reference_unit_test_grid(seqlen, itype, flags, groups, nrows):
    """The reference's own parametrisation (batch 2, dim 24, dstate 8), fwd + 7 grads, through the
    autograd front-end (fine checkpoints -> second-generation backward)."""
    has_bias, softplus, has_D = flags
    inp = _ref_test_inputs(seqlen, itype, groups, has_D, has_bias)
    u, delta, A, B, C, D, bias, dout = inp
    out, grads = _run_hip(u, delta, A, B, C, D, bias, dout, softplus, nrows)
    _compare(out, grads, u, delta, A, B, C, D, bias, dout, softplus, itype)


def _run_hip_module(u, delta, A, B, C, D, bias, dout, softplus, nrows):
    """fwd + bwd through the module-level operator entries (reference-shaped x, checkpoint pitch 1280
    -> first-generation backward), exactly the calls the reference's SelectiveScanFn makes."""
    core = _core()
    dev = "cuda"
    if B.dim() == 3:
        B, C = B.unsqueeze(1), C.unsqueeze(1)
    t = [None if v is None else v.to(dev) for v in (u, delta, A, B, C, D, bias)]
    out, x = core.fwd(*t, softplus, nrows)
    grads = core.bwd(*t, dout.to(dev), x, softplus, 1)
    if u.dim() == 3 and grads[3].dim() == 4 and grads[3].shape[1] == 1:
        pass
    return out, list(grads)


GRID_V1 = [g for g in GRID if g[4] in (1, 4) and g[2] in ((False, False, False), (True, True, True), (True, False, True),
                                                            (False, True, False))]


@pytest.mark.parametrize("seqlen,itype,flags,groups,nrows", GRID_V1, ids=[_grid_id(g) for g in GRID_V1])
def test_reference_unit_test_grid_module_entries(seqlen, itype, flags, groups, nrows):
    """A quarter of the grid through ``selective_scan_cuda_core.fwd / bwd`` themselves (the reference
    extension's entry points, x of the reference's shape): covers csrc/scan_bwd.hip."""
    has_bias, softplus, has_D = flags
    u, delta, A, B, C, D, bias, dout = _ref_test_inputs(seqlen, itype, groups, has_D, has_bias)
    out, grads = _run_hip_module(u, delta, A, B, C, D, bias, dout, softplus, nrows)
    if B.dim() == 3:
        grads[3], grads[4] = grads[3].squeeze(1), grads[4].squeeze(1)
    _compare(out, grads, u, delta, A, B, C, D, bias, dout, softplus, itype)


def _model_like(batch, KD, L, N, G, seed=0, itype=torch.float32):
    """Magnitudes of a freshly initialised SS2D block (vmamba.py:729-782): A = -(1..N),
    dt bias = softplus^-1(U_log[1e-3, 1e-1]), D = 1, small dt projections."""
    g = torch.Generator().manual_seed(seed)
    A = -torch.arange(1, N + 1, dtype=torch.float32).repeat(KD, 1) * (1 + 0.05 * torch.rand(KD, N, generator=g))
    B = torch.randn(batch, G, N, L, generator=g).to(itype)
    C = torch.randn(batch, G, N, L, generator=g).to(itype)
    D = 1.0 + 0.1 * torch.randn(KD, generator=g)
    tgt = torch.exp(torch.rand(KD, generator=g) * (np.log(0.1) - np.log(0.001)) + np.log(0.001))
    bias = tgt + torch.log(-torch.expm1(-tgt))
    u = torch.randn(batch, KD, L, generator=g).to(itype)
    delta = (0.5 * torch.randn(batch, KD, L, generator=g)).to(itype)
    dout = torch.randn(batch, KD, L, generator=g).to(itype)
    return u, delta, A, B, C, D, bias, dout


STAGE_SHAPES = [
    # (batch, KD, L, N, G)  -- SURVEY.md Appendix A, sigma_tiny/small @480x640
    (1, 768, 19200, 16, 4),     # encoder stage 0 (the headline kernel shape)
    (1, 1536, 4800, 16, 4),     # encoder stage 1
    (2, 3072, 1200, 16, 4),     # encoder stage 2, batch 2
    (1, 6144, 300, 16, 4),      # encoder stage 3
    (1, 192, 19200, 4, 1),      # CroMB stage 0
    (1, 384, 38400, 4, 2),      # ConMB stage 0 (longest sequence at 480x640)
    (1, 768, 19200, 4, 4),      # decoder @120x160
]


@pytest.mark.parametrize("itype", [torch.float32, torch.float16, torch.bfloat16], ids=["float32", "float16", "bfloat16"])
@pytest.mark.parametrize("shape", STAGE_SHAPES, ids=["x".join(map(str, s)) for s in STAGE_SHAPES])
def test_real_stage_shapes_forward_and_backward(shape, itype):
    """The seven launch shapes of the model at 480x640 in every IO dtype of the operator (16-bit IO takes
    the register staging path)."""
    batch, KD, L, N, G = shape
    u, delta, A, B, C, D, bias, dout = _model_like(batch, KD, L, N, G, itype=itype)
    out, grads = _run_hip(u, delta, A, B, C, D, bias, dout, True, 4 if (KD // G) % 4 == 0 else 1)
    _compare(out, grads, u, delta, A, B, C, D, bias, dout, True, itype)


LONG_SHAPES = [
    # (batch, KD, L, N, G, rev_mask, u_gshift): the sequence lengths of BASELINE configs[4] (720x1280, sigma_base):
    # 180x320 = 57600 (encoder stage 0), 2 x 57600 = 115200 (ConMB stage 0), 45x80 = 3600, 23x40 = 920
    (1, 32, 57600, 16, 4, 0b1010, 1),
    (1, 16, 115200, 4, 2, 0b10, 1),
    (2, 16, 3600, 16, 4, 0b1010, 1),
    (2, 16, 920, 16, 4, 0b1010, 1),
    (1, 8, 1840, 4, 2, 0b10, 1),
]


@pytest.mark.parametrize("shape", LONG_SHAPES, ids=["x".join(map(str, s[:5])) for s in LONG_SHAPES])
@pytest.mark.parametrize("pitch", [0, 640, 320, 160])
def test_long_sequences_of_the_720x1280_configuration(shape, pitch):
    """Operator-level oracle check (fwd + 7 grads) at the longest sequences the model produces, with the
    fused path's addressing (reversed groups, shared u / dout rows), for every checkpoint pitch
    (0 = reference-shaped x / first-generation backward, 160 = quad-row backward).  VERDICT r1 weak #1."""
    batch, KD, L, N, G, mask, ush = shape
    u, delta, A, B, C, D, bias, dout = _model_like(batch, KD, L, N, G, seed=17)
    rpg = KD // G
    keep = lambda t: torch.cat([t[:, g * rpg:(g + 1) * rpg] for g in range(0, G, 1 << ush)], dim=1).contiguous()
    full = lambda t: torch.cat([t[:, (g >> ush) * rpg:((g >> ush) + 1) * rpg] for g in range(G)], dim=1)
    u_h, g_h = keep(u), keep(dout)
    u_f, g_f = full(u_h), full(g_h)
    core = _core()
    dev = "cuda"
    args = [t.to(dev) for t in (u_h, delta, A, B, C, D, bias)]
    out, x = core.fwd_ext(*args, True, rev_mask=mask, u_gshift=ush, ckpt_pitch=pitch)
    grads = core.bwd_ext(*args, g_h.to(dev), x, True, rev_mask=mask, u_gshift=ush, dout_gshift=ush, ckpt_pitch=pitch)
    revs = [(mask >> g) & 1 for g in range(G)]
    fr = lambda t: torch.cat([t[:, g * rpg:(g + 1) * rpg].flip(-1) if revs[g] else t[:, g * rpg:(g + 1) * rpg] for g in range(G)], 1)
    fg = lambda t: torch.stack([t[:, g].flip(-1) if revs[g] else t[:, g] for g in range(G)], 1)
    so = _oracle()
    ref = fr(so.selective_scan_oracle(fr(u_f), fr(delta), A, fg(B), fg(C), D, bias, True, acc64=True))
    torch.testing.assert_close(out.cpu(), ref, rtol=6e-4, atol=2e-3)
    rg = list(so.selective_scan_oracle_bwd(fr(u_f), fr(delta), A, fg(B), fg(C), D, bias, fr(g_f), True))
    rg[0], rg[1], rg[3], rg[4] = fr(rg[0]), fr(rg[1]), fg(rg[3]), fg(rg[4])
    assert_grads_close(grads, rg)


def test_checkpoint_tensor_shape_and_documented_layout():
    """x has the reference's SHAPE (batch, dim, ceil(L/2048), 2N) (selective_scan.cpp:225-228); its
    contents are this library's private fwd->bwd scratch, documented in include/sigma_scan.h:
    checkpoint j = state after element min(L, (j+1)*1280)-1 at x[b, r].flatten()[j*N + n]; with
    fine_ckpt the pitch is 640 and x is (B, dim, ceil(L/640)*N)."""
    so = _oracle()
    batch, KD, L, N, G = 2, 8, 5000, 4, 2
    u, delta, A, B, C, D, bias, _ = _model_like(batch, KD, L, N, G, seed=3)
    dev = "cuda"
    out, x = _core().fwd(u.to(dev), delta.to(dev), A.to(dev), B.to(dev), C.to(dev), D.to(dev), bias.to(dev), True, 1)
    assert x.shape == (batch, KD, 3, 2 * N) and x.dtype == torch.float32
    x = x.cpu().reshape(batch, KD, -1)
    args = [t.to(dev) for t in (u, delta, A, B, C, D, bias)]
    out2, xf = _core().fwd_ext(*args, True, fine_ckpt=True)
    assert xf.shape == (batch, KD, 8 * N) and torch.equal(out2, out)
    out3, xq = _core().fwd_ext(*args, True, ckpt_pitch=320)
    assert xq.shape == (batch, KD, 16 * N) and torch.equal(out3, out)
    out4, xo = _core().fwd_ext(*args, True, ckpt_pitch=160)
    assert xo.shape == (batch, KD, 32 * N)
    # pitch 160 runs the quad-row forward (csrc/scan_fwd4.hip): other summation order, same values to rounding
    torch.testing.assert_close(out4, out, rtol=2e-5, atol=2e-6 * float(out.abs().max()))
    xf, xq, xo = xf.cpu(), xq.cpu(), xo.cpu()
    for pitch, xt in ((1280, x), (640, xf), (320, xq), (160, xo)):
        for j in range((L + pitch - 1) // pitch):
            end = min(L, (j + 1) * pitch)
            _, st = so.selective_scan_oracle(u[..., :end], delta[..., :end], A, B[..., :end], C[..., :end], D, bias, True,
                                             acc64=True, return_last_state=True)
            torch.testing.assert_close(xt[:, :, j * N:(j + 1) * N], st, rtol=1e-3, atol=1e-4)
    # the backward gives the same gradients from either checkpoint tensor
    dout = torch.randn(batch, KD, L).to(dev)
    g1 = _core().bwd_ext(*args, dout, x.to(dev).view(batch, KD, 3, 2 * N), True)
    for xt, pitch in ((xf, 640), (xq, 320), (xo, 160)):
        g2 = _core().bwd_ext(*args, dout, xt.to(dev), True, ckpt_pitch=pitch)
        for a, b in zip(g1, g2):
            torch.testing.assert_close(a, b, rtol=2e-4, atol=1e-5 + 1e-5 * float(b.abs().max()))


def _run_hip_ext(u, delta, A, B, C, D, bias, dout, rev_mask=0, u_gshift=0, dout_gshift=0):
    """fwd + bwd through the extended operator entry (reverse groups / shared u rows)."""
    core = _core()
    dev = "cuda"
    args = [t.to(dev) for t in (u, delta, A, B, C, D, bias)]
    out, x = core.fwd_ext(*args, True, rev_mask=rev_mask, u_gshift=u_gshift)
    grads = core.bwd_ext(*args, dout.to(dev), x, True, rev_mask=rev_mask, u_gshift=u_gshift, dout_gshift=dout_gshift)
    return out.cpu(), [g.cpu() for g in grads]


@pytest.mark.parametrize("L", [300, 640, 1200, 1283, 2564, 4800])
@pytest.mark.parametrize("mask", [0b1100, 0b1010])
def test_reversed_groups_equal_flipped_inputs(L, mask):
    """CrossScan's flip done by addressing: with some of the four groups reversed, the result
    must equal the plain operator applied to explicitly flipped copies (vmamba.py:80-121)."""
    batch, KD, N, G = 2, 32, 16, 4
    u, delta, A, B, C, D, bias, dout = _model_like(batch, KD, L, N, G, seed=11)
    out, grads = _run_hip_ext(u, delta, A, B, C, D, bias, dout, rev_mask=mask)
    rpg = KD // G
    revs = [(mask >> g) & 1 for g in range(G)]

    def flip_rows(t):            # (B, KD, L): rows of reversed groups flipped along L
        return torch.cat([t[:, g * rpg:(g + 1) * rpg].flip(-1) if revs[g] else t[:, g * rpg:(g + 1) * rpg]
                          for g in range(G)], dim=1)

    def flip_groups(t):          # (B, G, N, L)
        return torch.stack([t[:, g].flip(-1) if revs[g] else t[:, g] for g in range(G)], dim=1)

    so = _oracle()
    uf, df, Bf, Cf, gf = flip_rows(u), flip_rows(delta), flip_groups(B), flip_groups(C), flip_rows(dout)
    ref = flip_rows(so.selective_scan_oracle(uf, df, A, Bf, Cf, D, bias, True, acc64=True))
    torch.testing.assert_close(out, ref, rtol=6e-4, atol=2e-3)
    rg = list(so.selective_scan_oracle_bwd(uf, df, A, Bf, Cf, D, bias, gf, True))
    rg[0], rg[1] = flip_rows(rg[0]), flip_rows(rg[1])
    rg[3], rg[4] = flip_groups(rg[3]), flip_groups(rg[4])
    assert_grads_close(grads, rg)


def test_shared_u_and_dout_rows():
    """u_gshift / dout_gshift: groups 2j and 2j+1 share one physical copy of u (and of dout), as the
    forward and the flipped direction of one memory order do in SS2D; dB / dC go straight into views."""
    batch, KD, L, N, G = 1, 32, 900, 4, 4
    u, delta, A, B, C, D, bias, dout = _model_like(batch, KD, L, N, G, seed=13)
    rpg = KD // G
    u_half = torch.cat([u[:, 0:rpg], u[:, 2 * rpg:3 * rpg]], dim=1).contiguous()          # groups 0 and 2
    g_half = torch.cat([dout[:, 0:rpg], dout[:, 2 * rpg:3 * rpg]], dim=1).contiguous()
    u_full = torch.cat([u_half[:, :rpg]] * 2 + [u_half[:, rpg:]] * 2, dim=1)
    g_full = torch.cat([g_half[:, :rpg]] * 2 + [g_half[:, rpg:]] * 2, dim=1)
    core = _core()
    dev = "cuda"
    args = [t.to(dev) for t in (u_half, delta, A, B, C, D, bias)]
    out, x = core.fwd_ext(*args, True, u_gshift=1)
    big = torch.zeros(batch, G, 2 * N + 3, L, device=dev)
    grads = core.bwd_ext(*args, g_half.to(dev), x, True, u_gshift=1, dout_gshift=1,
                         dB_out=big[:, :, 3:3 + N], dC_out=big[:, :, 3 + N:])
    assert grads[3].data_ptr() == big[:, :, 3:3 + N].data_ptr() and float(big[:, :, :3].abs().max()) == 0.0
    so = _oracle()
    ref = so.selective_scan_oracle(u_full, delta, A, B, C, D, bias, True, acc64=True)
    torch.testing.assert_close(out.cpu(), ref, rtol=6e-4, atol=2e-3)
    rg = so.selective_scan_oracle_bwd(u_full, delta, A, B, C, D, bias, g_full, True)
    assert_grads_close(grads, rg)


def test_strided_and_unaligned_inputs():
    """Arbitrary batch/row strides are honoured (selective_scan.cpp:88-103); odd L and
    offset storage take the scalar path."""
    so = _oracle()
    dev = "cuda"
    batch, KD, L, N, G = 2, 12, 777, 4, 3
    u, delta, A, B, C, D, bias, dout = _model_like(batch, KD, L, N, G, seed=5)
    big_u = torch.zeros(batch, KD, 2, L + 3, device=dev)
    big_u[:, :, 1, 1:L + 1] = u.to(dev)
    u_s = big_u[:, :, 1, 1:L + 1]                       # strided in batch/row, storage offset 1 element
    big_d = torch.zeros(KD, batch, L, device=dev)
    big_d.copy_(delta.to(dev).transpose(0, 1))
    d_s = big_d.transpose(0, 1)                          # (batch, KD, L) with swapped strides
    assert u_s.stride(-1) == 1 and d_s.stride(-1) == 1 and not d_s.is_contiguous()
    out, x = _core().fwd(u_s, d_s, A.to(dev), B.to(dev), C.to(dev), D.to(dev), bias.to(dev), True, 1)
    ref = so.selective_scan_oracle(u, delta, A, B, C, D, bias, True, acc64=True)
    torch.testing.assert_close(out.cpu(), ref, rtol=6e-4, atol=2e-3)
    grads = _core().bwd(u_s, d_s, A.to(dev), B.to(dev), C.to(dev), D.to(dev), bias.to(dev), dout.to(dev), x, True, 1)
    rg = so.selective_scan_oracle_bwd(u, delta, A, B, C, D, bias, dout, True)
    assert_grads_close(grads, rg)


@pytest.mark.parametrize("opt", [("fwd_items", 4), ("fwd_items", 5), ("fwd_items", 10), ("fwd_items", 20),
                                 ("fwd_waves", 1), ("fwd_waves", 2), ("fwd_waves", 16), ("fwd_tiles", 1),
                                 ("fwd_tiles", 2), ("fwd_tiles", 4), ("fwd_nb", 1), ("fwd_nb", 2), ("no_glds", 1),
                                 ("bwd_items", 4), ("bwd_items", 5), ("bwd_items", 10), ("bwd_waves", 1),
                                 ("bwd_waves", 8), ("bwd_nb", 1), ("bwd_nb", 2), ("bwd_slab2", 1), ("bwd_slab2", 2),
                                 ("bwd_gen", 1), ("bwd_gen", 2), ("bwd_rb", 2), ("bwd_rb", 4), ("bwd_waves", 16),
                                 ("bwd_waves", 4)])
def test_every_launch_geometry_is_correct(opt):
    """All (items per lane, rows per workgroup) variants compute the same thing."""
    from sigma_amd import _capi
    name, val = opt
    batch, KD, L, N, G = 2, 96, 2500, 16, 2
    u, delta, A, B, C, D, bias, dout = _model_like(batch, KD, L, N, G, seed=7)
    _capi.set_option(name, val)
    try:
        out, grads = _run_hip(u, delta, A, B, C, D, bias, dout, True, 1)
    finally:
        _capi.set_option(name, 0)
    _compare(out, grads, u, delta, A, B, C, D, bias, dout, True, torch.float32)


QUAD_OPTS = [{}, {"bwd_sb": 1}, {"bwd_sb": 4}, {"bwd_waves": 3}, {"bwd_waves": 6, "bwd_rb": 2}, {"bwd_waves": 12}, {"bwd_waves": 16},
             {"bwd_rb": 1}, {"bwd_touch": 2}, {"bwd_waves": 16, "bwd_sb": 1, "bwd_touch": 1}, {"bwd_wgs": 2},
             {"bwd_seg": 2}, {"bwd_seg": 4, "bwd_waves": 3}, {"bwd_seg": 3, "bwd_rb": 1, "bwd_nb": 1}, {"bwd_seg": 1}]


@pytest.mark.parametrize("opts", QUAD_OPTS, ids=[",".join(f"{k}={v}" for k, v in o.items()) or "auto" for o in QUAD_OPTS])
@pytest.mark.parametrize("shape", [(2, 768, 1200, 16, 4, 0b1010, 1), (2, 384, 2564, 4, 2, 0b10, 1), (3, 192, 300, 8, 1, 0, 0),
                                   (1, 96, 4960, 16, 4, 0b0110, 1)],
                         ids=["2x768x1200xN16", "2x384x2564xN4", "3x192x300xN8", "1x96x4960xN16-segments"])
def test_quad_row_backward_geometries_against_oracle(shape, opts):
    """csrc/scan_bwd4.hip (ckpt_pitch 160) in every launch geometry -- waves per workgroup, states per barrier, row
    blocks, L2 touches on/off, two workgroups per CU, sequence segments (forced and automatic: the few-row shape) --
    against the CPU oracle: all seven gradients, reversed groups, shared u / dout rows."""
    batch, KD, L, N, G, mask, ush = shape
    u, delta, A, B, C, D, bias, dout = _model_like(batch, KD, L, N, G, seed=23)
    rpg = KD // G
    keep = lambda t: torch.cat([t[:, g * rpg:(g + 1) * rpg] for g in range(0, G, 1 << ush)], dim=1).contiguous()
    full = lambda t: torch.cat([t[:, (g >> ush) * rpg:((g >> ush) + 1) * rpg] for g in range(G)], dim=1)
    u_h, g_h = keep(u), keep(dout)
    u_f, g_f = full(u_h), full(g_h)
    core = _core()
    dev = "cuda"
    args = [t.to(dev) for t in (u_h, delta, A, B, C, D, bias)]
    assert core.quad_backward_ok(args[0], args[1], args[3], args[4])
    out, x = core.fwd_ext(*args, True, rev_mask=mask, u_gshift=ush, ckpt_pitch=160)
    from sigma_amd import _capi
    try:
        for k, v in opts.items():
            _capi.set_option(k, v)
        grads = core.bwd_ext(*args, g_h.to(dev), x, True, rev_mask=mask, u_gshift=ush, dout_gshift=ush, ckpt_pitch=160)
    finally:
        for k in opts:
            _capi.set_option(k, 0)
    revs = [(mask >> g) & 1 for g in range(G)]
    fr = lambda t: torch.cat([t[:, g * rpg:(g + 1) * rpg].flip(-1) if revs[g] else t[:, g * rpg:(g + 1) * rpg] for g in range(G)], 1)
    fg = lambda t: torch.stack([t[:, g].flip(-1) if revs[g] else t[:, g] for g in range(G)], 1)
    so = _oracle()
    rg = list(so.selective_scan_oracle_bwd(fr(u_f), fr(delta), A, fg(B), fg(C), D, bias, fr(g_f), True))
    rg[0], rg[1], rg[3], rg[4] = fr(rg[0]), fr(rg[1]), fg(rg[3]), fg(rg[4])
    assert_grads_close(grads, rg)


FULL_LAUNCHES = [
    # (batch, KD, L, N, G, rev_mask, u_gshift, kernel family the automatic policy must pick): whole launches of the
    # sigma_small training step at batch 8 (2 x 8 encoder images) -- the dominant one (encoder stage 2: row-lane kernels,
    # pitch 16, 12 row blocks per group through the dB / dC slabs) and the SURVEY headline shape at the same batch
    # (quad-row kernels, pitch 160) -- and the same two stages of the one-image-per-GPU step (row-lane, few rows: sequence
    # segments on the long one)
    (16, 3072, 1200, 16, 4, 0b1010, 1, 16),
    (16, 768, 19200, 16, 4, 0b1010, 1, 160),
    (2, 3072, 1200, 16, 4, 0b1010, 1, 16),
    (1, 768, 19200, 16, 4, 0b1010, 1, 16),
    # round 6: the three largest 4-state launches of the batch-8 step -- decoder stage 0 (MambaDecoder.py:103,
    # d_state 4), ConMB stage 0 (vmamba.py:369-430: K = 2, one direction reversed, the 2 L sequence) and decoder stage 1 --
    # through the automatic policy: 64-lane kernels at pitch 640 for the long rows, quad-row at pitch 160 for (8,1536,4800)
    (8, 768, 19200, 4, 4, 0b1010, 1, 640),
    (8, 384, 38400, 4, 2, 0b10, 1, 640),
    (8, 1536, 4800, 4, 4, 0b1010, 1, 160),
    # round 6: the REST of the 15 distinct launch shapes of the batch-8 step, so that every scan launch the bench times has
    # run at its real size against the oracle with the kernels the policy picks for it: encoder stages 1 and 3 (quad-row),
    # decoder stage 2, ConMB stages 1-3 (64-lane at 9600, quad-row below), CroMB stages 0-3 (one group, plain direction:
    # Cross_Mamba_Attention_SSM through selective_scan_fn, vmamba.py:1407-1545 -- row-lane where rowlane_pays, quad-row at L = 300)
    (16, 1536, 4800, 16, 4, 0b1010, 1, 160),
    (16, 6144, 300, 16, 4, 0b1010, 1, 160),
    (8, 3072, 1200, 4, 4, 0b1010, 1, 160),
    (8, 768, 9600, 4, 2, 0b10, 1, 640),
    (8, 1536, 2400, 4, 2, 0b10, 1, 160),
    (8, 3072, 600, 4, 2, 0b10, 1, 160),
    (8, 192, 19200, 4, 1, 0, 0, 16),
    (8, 384, 4800, 4, 1, 0, 0, 16),
    (8, 768, 1200, 4, 1, 0, 0, 16),
    (8, 1536, 300, 4, 1, 0, 0, 160),
]


@pytest.mark.parametrize("shape", FULL_LAUNCHES, ids=["x".join(map(str, s[:3])) + "xN%d" % s[3] + ("g%d" % s[4] if s[4] != 4 else "") for s in FULL_LAUNCHES])
def test_full_size_step_launches_against_oracle(shape):
    """The benchmarked launches AT THEIR REAL SIZE, with the pitch / kernels the fused model path (SS2DCoreFn) picks
    automatically -- ckpt_pitch_for called exactly as ss2d_fused calls it, with rowlane_ok of the operands and the group
    count: 16 = scan_fwdr + scan_bwdr, 160 = scan_fwd4 + scan_bwd4, each with the planner's own geometry -- forward and
    all seven gradients against the CPU oracle (OpenMP; a few seconds per launch), per-gradient tolerances."""
    from sigma_amd.ss2d_fused import ckpt_pitch_for
    batch, KD, L, N, G, mask, ush, want_pitch = shape
    u, delta, A, B, C, D, bias, dout = _model_like(batch, KD, L, N, G, seed=31)
    rpg = KD // G
    keep = lambda t: torch.cat([t[:, g * rpg:(g + 1) * rpg] for g in range(0, G, 1 << ush)], dim=1).contiguous()
    full = lambda t: torch.cat([t[:, (g >> ush) * rpg:((g >> ush) + 1) * rpg] for g in range(G)], dim=1)
    u_h, g_h = keep(u), keep(dout)
    u_f, g_f = full(u_h), full(g_h)
    core = _core()
    dev = "cuda"
    args = [t.to(dev) for t in (u_h, delta, A, B, C, D, bias)]
    g_dev = g_h.to(dev)
    pitch = ckpt_pitch_for(L, N, batch * KD, core.quad_backward_ok(args[0], args[1], args[3], args[4]),
                           core.rowlane_ok(args[0], args[1], args[3], args[4], g_dev), G)
    assert pitch == want_pitch
    out, x = core.fwd_ext(*args, True, rev_mask=mask, u_gshift=ush, ckpt_pitch=pitch)
    grads = core.bwd_ext(*args, g_dev, x, True, rev_mask=mask, u_gshift=ush, dout_gshift=ush, ckpt_pitch=pitch)
    revs = [(mask >> g) & 1 for g in range(G)]
    fr = lambda t: torch.cat([t[:, g * rpg:(g + 1) * rpg].flip(-1) if revs[g] else t[:, g * rpg:(g + 1) * rpg] for g in range(G)], 1)
    fg = lambda t: torch.stack([t[:, g].flip(-1) if revs[g] else t[:, g] for g in range(G)], 1)
    so = _oracle()
    ref = fr(so.selective_scan_oracle(fr(u_f), fr(delta), A, fg(B), fg(C), D, bias, True, acc64=False))
    torch.testing.assert_close(out.cpu(), ref, rtol=6e-4, atol=2e-3)
    rg = list(so.selective_scan_oracle_bwd(fr(u_f), fr(delta), A, fg(B), fg(C), D, bias, fr(g_f), True))
    rg[0], rg[1], rg[3], rg[4] = fr(rg[0]), fr(rg[1]), fg(rg[3]), fg(rg[4])
    assert_grads_close(grads, rg)


def test_quad_row_backward_refuses_what_it_cannot_take():
    """ckpt_pitch 160 with 16-bit IO / odd lengths must fail loudly (no silent fallback), and quad_backward_ok says so first."""
    core = _core()
    dev = "cuda"
    u, delta, A, B, C, D, bias, dout = _model_like(1, 16, 322, 4, 2, seed=1)
    args = [t.to(dev) for t in (u, delta, A, B, C, D, bias)]
    assert not core.quad_backward_ok(args[0], args[1], args[3], args[4])          # L % 4 != 0
    out, x = core.fwd_ext(*args, True, ckpt_pitch=160)                          # the forward takes any pitch
    with pytest.raises(RuntimeError, match="ckpt_pitch 160"):
        core.bwd_ext(*args, dout.to(dev), x, True, ckpt_pitch=160)
    h = [t.to(dev).half() if i in (0, 1, 3, 4) else t.to(dev) for i, t in enumerate(_model_like(1, 16, 320, 4, 2, seed=1)[:7])]
    assert not core.quad_backward_ok(h[0], h[1], h[3], h[4])


def test_error_behaviour_matches_reference_checks():
    core = _core()
    dev = "cuda"
    u = torch.randn(1, 8, 16, device=dev)
    A = -torch.rand(8, 4, device=dev)
    Bm = torch.randn(1, 1, 4, 16, device=dev)
    with pytest.raises(RuntimeError):                       # dtype mismatch (selective_scan.cpp:179-181)
        core.fwd(u, u.half(), A, Bm, Bm, None, None, False, 1)
    with pytest.raises(RuntimeError):                       # dim % (groups*nrows) (selective_scan.cpp:200)
        core.fwd(u, u, A, torch.randn(1, 3, 4, 16, device=dev), torch.randn(1, 3, 4, 16, device=dev), None, None, False, 1)
    with pytest.raises(RuntimeError):                       # CPU tensor (selective_scan.cpp:183-187)
        core.fwd(u.cpu(), u.cpu(), A.cpu(), Bm.cpu(), Bm.cpu(), None, None, False, 1)
    with pytest.raises(RuntimeError):                       # A must be fp32 (:177)
        core.fwd(u, u, A.half(), Bm, Bm, None, None, False, 1)
    long_u = torch.randn(1, 8, 3000, device=dev)
    long_B = torch.randn(1, 1, 4, 3000, device=dev)
    with pytest.raises(RuntimeError):                       # x required when n_chunks > 1 (:320)
        core.bwd(long_u, long_u, A, long_B, long_B, None, None, long_u, None, False, 1)


def test_empty_and_tiny_sequences():
    core = _core()
    so = _oracle()
    dev = "cuda"
    A = -torch.rand(4, 3)
    out, x = core.fwd(torch.randn(1, 4, 0, device=dev), torch.randn(1, 4, 0, device=dev), A.to(dev),
                      torch.randn(1, 1, 3, 0, device=dev), torch.randn(1, 1, 3, 0, device=dev), None, None, False, 1)
    assert out.shape == (1, 4, 0)
    for L in (1, 2, 3, 5, 63, 65):
        u, d = torch.randn(2, 4, L), torch.rand(2, 4, L)
        Bm, Cm = torch.randn(2, 2, 3, L), torch.randn(2, 2, 3, L)
        out, _ = core.fwd(u.to(dev), d.to(dev), A.to(dev), Bm.to(dev), Cm.to(dev), None, None, False, 1)
        torch.testing.assert_close(out.cpu(), so.selective_scan_oracle(u, d, A, Bm, Cm, acc64=True), rtol=6e-4, atol=2e-3)


def test_full_size_properties_linearity_and_determinism():
    """Size-independent properties at the headline shape (1, 768, 19200), N=16: the scan is
    linear in u for fixed delta, and the forward is bit-deterministic (no atomics)."""
    dev = "cuda"
    batch, KD, L, N, G = 1, 768, 19200, 16, 4
    u, delta, A, B, C, D, bias, _ = [None if t is None else t.to(dev) for t in _model_like(batch, KD, L, N, G, seed=11)]
    u2 = torch.randn_like(u)
    f = lambda uu: _core().fwd(uu, delta, A, B, C, None, bias, True, 4)[0]
    lhs = f(u + 2 * u2)
    rhs = f(u) + 2 * f(u2)
    torch.testing.assert_close(lhs, rhs, rtol=2e-4, atol=2e-4)
    # determinism of the forward (no atomics on the forward path)
    assert torch.equal(f(u), f(u))


# ---- row-lane kernels (csrc/scan_fwdr.hip / scan_bwdr.hip, checkpoint pitch 16; round 4) ------------------------------
ROWLANE_CASES = [
    # (batch, KD, L, N, G, rev_mask, u_gshift), library options
    ((2, 256, 1200, 16, 4, 0b1010, 1), {}),                  # one row block per group: dB/dC written directly
    ((2, 512, 1200, 16, 4, 0b1010, 1), {}),                  # two row blocks per group: workspace slabs + reduce
    ((1, 256, 1204, 16, 4, 0b0110, 1), {}),                  # partial last tile (L % 16 == 4), other flip pattern
    ((3, 192, 300, 8, 1, 0, 0), {}),                         # 8 states, one group, L % 16 == 12
    ((3, 192, 300, 8, 1, 1, 0), {"rl_waves": 8}),            # 8 states, eight forward state waves, reversed
    ((2, 384, 2564, 4, 2, 0b10, 1), {}),                     # 4 states (fusion / decoder), ConMB's flipped second group
    ((2, 256, 1200, 16, 4, 0b1010, 1), {"rl_segs": 3}),      # forced sequence segments (summary pre-passes)
    ((2, 256, 1200, 16, 4, 0b1010, 1), {"rl_segs": 5, "rl_waves": 8}),
    ((2, 256, 1200, 16, 4, 0b1010, 1), {"rl_waves": 16}),    # forward with 16 state waves
    ((1, 256, 4800, 16, 4, 0b1010, 1), {}),                  # few rows: automatic segments
    ((1, 768, 19200, 16, 4, 0b1010, 1), {}),                 # one image per GPU, encoder stage 0 (SURVEY's headline shape)
    ((1, 384, 38400, 4, 2, 0b10, 1), {}),                    # one image per GPU, ConMB stage 0 (the longest sequence at 480x640)
    ((11, 3072, 176, 16, 4, 0b1010, 1), {"rl_chain": 2}),    # 528 row blocks > resident workgroups: chained walk of the backward
    ((11, 3072, 172, 16, 4, 0b0101, 1), {"rl_chain": 2}),    # ... with a partial last tile
    ((2, 64, 16, 4, 1, 0, 0), {}),                           # exactly one tile
    ((2, 64, 8, 4, 1, 1, 0), {}),                            # less than one tile, reversed
]


def _rowlane_id(c):
    shape, opts = c
    return "x".join(map(str, shape[:5])) + (("-" + ",".join(f"{k}={v}" for k, v in opts.items())) if opts else "")


def _rowlane_run(shape, opts, softplus=True, with_D=True, with_bias=True, seed=23):
    batch, KD, L, N, G, mask, ush = shape
    u, delta, A, B, C, D, bias, dout = _model_like(batch, KD, L, N, G, seed=seed)
    if not softplus:
        delta, bias = delta.abs(), bias.abs()             # a negative step size makes the recurrence itself explode
    if not with_D:
        D = None
    if not with_bias:
        bias = None
    rpg = KD // G
    keep = lambda t: torch.cat([t[:, g * rpg:(g + 1) * rpg] for g in range(0, G, 1 << ush)], dim=1).contiguous()
    full = lambda t: torch.cat([t[:, (g >> ush) * rpg:((g >> ush) + 1) * rpg] for g in range(G)], dim=1)
    u_h, g_h = keep(u), keep(dout)
    u_f, g_f = full(u_h), full(g_h)
    core = _core()
    dev = "cuda"
    args = [None if t is None else t.to(dev) for t in (u_h, delta, A, B, C, D, bias)]
    assert core.rowlane_ok(args[0], args[1], args[3], args[4])
    from sigma_amd import _capi
    try:
        for k, v in opts.items():
            _capi.set_option(k, v)
        out, x = core.fwd_ext(*args, softplus, rev_mask=mask, u_gshift=ush, ckpt_pitch=16)
        out_nox, _ = core.fwd_ext(*args, softplus, rev_mask=mask, u_gshift=ush, ckpt_pitch=16, need_x=False)
        grads = core.bwd_ext(*args, g_h.to(dev), x, softplus, rev_mask=mask, u_gshift=ush, dout_gshift=ush, ckpt_pitch=16)
    finally:
        for k in opts:
            _capi.set_option(k, 0)
    assert x.shape == (batch, KD, max((L + 15) // 16, 1) * N)
    revs = [(mask >> g) & 1 for g in range(G)]
    fr = lambda t: torch.cat([t[:, g * rpg:(g + 1) * rpg].flip(-1) if revs[g] else t[:, g * rpg:(g + 1) * rpg] for g in range(G)], 1)
    fg = lambda t: torch.stack([t[:, g].flip(-1) if revs[g] else t[:, g] for g in range(G)], 1)
    so = _oracle()
    ref = fr(so.selective_scan_oracle(fr(u_f), fr(delta), A, fg(B), fg(C), D, bias, softplus, acc64=True))
    torch.testing.assert_close(out.cpu(), ref, rtol=6e-4, atol=2e-3)
    assert torch.equal(out_nox, out)                      # inference (no checkpoints) runs the same arithmetic
    rg = list(so.selective_scan_oracle_bwd(fr(u_f), fr(delta), A, fg(B), fg(C), D, bias, fr(g_f), softplus))
    rg[0], rg[1], rg[3], rg[4] = fr(rg[0]), fr(rg[1]), fg(rg[3]), fg(rg[4])
    assert_grads_close(grads, rg)


@pytest.mark.parametrize("case", ROWLANE_CASES, ids=[_rowlane_id(c) for c in ROWLANE_CASES])
def test_row_lane_kernels_against_oracle(case):
    """csrc/scan_fwdr.hip / scan_bwdr.hip (ckpt_pitch 16: a lane is a channel row, B / C as scalar operands, per-tile
    checkpoints, lane-reduce network for dB / dC) against the CPU oracle: forward with and without checkpoints and all
    seven gradients; reversed groups, shared u / dout rows, partial tiles, workspace slabs, state-wave counts, sequence
    segments (forced and automatic) and the chained walk."""
    shape, opts = case
    _rowlane_run(shape, opts)
    if opts.get("rl_chain") == 2:
        from sigma_amd import _capi
        # no hand-over wait ran out (bwd_ext raises when one does; the count is read-only and resets on read)
        assert _capi.get_option("rl_chain_timeouts") == 0
        with pytest.raises(RuntimeError):
            _capi.set_option("rl_chain_timeouts", 0)


@pytest.mark.parametrize("flags", [dict(softplus=False), dict(with_D=False, with_bias=False), dict(softplus=False, with_D=False)],
                         ids=["no-softplus", "no-D-no-bias", "no-softplus-no-D"])
def test_row_lane_kernels_optional_operands(flags):
    _rowlane_run((2, 256, 1200, 16, 4, 0b1010, 1), {}, **flags)


def test_row_lane_kernels_refuse_what_they_cannot_take():
    """ckpt_pitch 16 with 16-bit IO, rows per group not divisible by 64 or an odd length must fail loudly in BOTH entry
    points (no silent fallback), and rowlane_ok says so first."""
    core = _core()
    u, delta, A, B, C, D, bias, dout = _model_like(2, 96, 640, 16, 2, seed=5)       # 48 rows per group
    args = [t.cuda() for t in (u, delta, A, B, C, D, bias)]
    assert not core.rowlane_ok(args[0], args[1], args[3], args[4])
    with pytest.raises(RuntimeError):
        core.fwd_ext(*args, True, ckpt_pitch=16)
    u, delta, A, B, C, D, bias, dout = _model_like(2, 128, 642, 16, 2, seed=5)      # L % 4 != 0
    args = [t.cuda() for t in (u, delta, A, B, C, D, bias)]
    assert not core.rowlane_ok(args[0], args[1], args[3], args[4])
    with pytest.raises(RuntimeError):
        core.fwd_ext(*args, True, ckpt_pitch=16)
    u, delta, A, B, C, D, bias, dout = _model_like(2, 128, 640, 16, 2, seed=5, itype=torch.bfloat16)
    args = [t.cuda() for t in (u, delta, A, B, C, D, bias)]
    assert not core.rowlane_ok(args[0], args[1], args[3], args[4])
    with pytest.raises(RuntimeError):
        core.fwd_ext(*args, True, ckpt_pitch=16)


def test_row_lane_policy_through_the_autograd_function():
    """selective_scan_fn picks the row-lane kernels by itself on the shapes ckpt_pitch_for lists (here: the one-image
    ConMB launch) and gives the oracle's gradients through autograd, including an unaligned dout."""
    from sigma_amd.ss2d_fused import ckpt_pitch_for
    batch, KD, L, N, G = 1, 384, 2400, 4, 2
    u, delta, A, B, C, D, bias, dout = _model_like(batch, KD, L, N, G, seed=41)
    core = _core()
    dev = "cuda"
    leaves = [t.to(dev).requires_grad_() for t in (u, delta, A, B, C, D, bias)]
    assert ckpt_pitch_for(L, N, batch * KD, False, core.rowlane_ok(*[leaves[i].detach() for i in (0, 1, 3, 4)]), G) == 16
    out = _fn()(*leaves, True, 1)
    pad = torch.zeros(batch, KD, L + 1, device=dev)
    pad[..., 1:] = dout.to(dev)
    out.backward(pad[..., 1:])                             # rows that start 4 bytes off a 16-byte boundary
    so = _oracle()
    ref = so.selective_scan_oracle(u, delta, A, B, C, D, bias, True, acc64=True)
    torch.testing.assert_close(out.detach().cpu(), ref, rtol=6e-4, atol=2e-3)
    rg = so.selective_scan_oracle_bwd(u, delta, A, B, C, D, bias, dout, True)
    assert_grads_close([t.grad for t in leaves], rg)
