"""GPU parity tests of the split-operand bf16 MFMA GEMMs (csrc/gemm_split.hip, include/sigma_gemm.h; run with -m gpu).

Checker: the same product in fp64 (torch on the GPU).  A product of fp32 operands split into (hi, lo) bf16 pairs with the
lo*lo term dropped has a relative error of ~2^-16 per product term at worst and ~4e-6 rms on sums; the bound used here is
3e-5 * sum_k |a||b| per output element (the fp32 library GEMM sits at ~1e-6 of the same scale).  The model-level
tolerance (1e-3 on logits, the reference's gradient tolerances) is checked in tests/test_model_gpu.py with these kernels on.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (scale * torch.randn(*shape, generator=g)).to(DEV)


def _bound(a64, b64):
    """sum_k |a||b| per output element for C = a @ b (both given in product orientation (M, K) x (K, N))"""
    return a64.abs() @ b64.abs()


def _assert_close(got, want64, bound, what):
    err = (got.double() - want64).abs()
    worst = float((err / (bound + 1e-30)).max())
    assert worst < 3e-5, f"{what}: max error / sum|a||b| = {worst:.3e}"
    rms = float(err.pow(2).mean().sqrt() / want64.pow(2).mean().sqrt())
    assert rms < 2e-5, f"{what}: relative rms error {rms:.3e}"


NT_SHAPES = [
    # (M, K, N): in_proj / out_proj / PatchMerging / decoder shapes of sigma_small at 480x640 plus ragged ones
    (19200, 384, 1536), (19200, 768, 384), (4800, 192, 768), (1200, 1536, 768), (2400, 96, 384), (3000, 192, 96),
    (300, 768, 1536), (257, 100, 70), (128, 32, 128), (5, 4, 3), (1, 8, 1), (130, 36, 200),
]


@pytest.mark.parametrize("shape", NT_SHAPES, ids=["x".join(map(str, s)) for s in NT_SHAPES])
@pytest.mark.parametrize("with_bias", [False, True], ids=["nobias", "bias"])
def test_gemm_nt_against_fp64(shape, with_bias):
    from sigma_amd import gemm
    M, K, N = shape
    a, w = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=0.05)
    bias = _rand(N, seed=3) if with_bias else None
    got = gemm.gemm_nt(a, w, bias)
    want = a.double() @ w.double().t() + (bias.double() if with_bias else 0.0)
    _assert_close(got, want, _bound(a.double(), w.double().t()), "nt")


@pytest.mark.parametrize("shape", [(19200, 384, 1536), (300, 768, 1536), (18, 768, 1536), (18, 1536, 768), (6, 768, 1536), (792, 96, 384),
                                   (257, 100, 70), (5, 4, 3)], ids=lambda s: "x".join(map(str, s)))
def test_three_piece_gemms_reach_fp32_accuracy(shape):
    """pieces = 3 (hi, mid, lo: six MFMAs per block): error against fp64 at the level of an fp32 GEMM (~1e-6 of
    sum |a||b|), for the forward (nt), the input gradient (nn) and the weight gradient (tn)."""
    from sigma_amd import gemm
    M, K, N = shape
    a, w = _rand(M, K, seed=21), _rand(N, K, seed=22, scale=0.05)
    bias = _rand(N, seed=23)
    def check(got, want, bound, what):
        worst = float(((got.double() - want).abs() / (bound + 1e-30)).max())
        assert worst < 2e-6, f"{what}: max error / sum|a||b| = {worst:.3e}"
    check(gemm.gemm_nt(a, w, bias, pieces=3), a.double() @ w.double().t() + bias.double(), _bound(a.double(), w.double().t()), "nt p3")
    if N % 4 == 0 and K % 4 == 0:
        dy = _rand(M, N, seed=24)
        check(gemm.gemm_nn(dy, w, pieces=3), dy.double() @ w.double(), _bound(dy.double(), w.double()), "nn p3")
        check(gemm.gemm_tn(dy, a, pieces=3), dy.double().t() @ a.double(), _bound(dy.double().t(), a.double()), "tn p3")


def test_gemm_nt_is_not_transposed_identity_check():
    """A = I with an ASYMMETRIC B: catches a row/column swap in the accumulator write (a symmetric B would not)."""
    from sigma_amd import gemm
    n = 160
    a = torch.eye(n, device=DEV)
    w = (torch.arange(n * n, device=DEV, dtype=torch.float32).view(n, n) % 251) / 16.0      # exactly representable in bf16 x 2
    got = gemm.gemm_nt(a, w)                                      # C = I @ w^T
    assert torch.equal(got, w.t().contiguous())


def test_gemm_nt_strided_operands_accumulate_and_out():
    """A as the x half of an in_proj output (row stride 2K), out as a column slice of a wider buffer, accumulate."""
    from sigma_amd import gemm
    M, K, N = 1000, 192, 96
    xz = _rand(M, 2 * K, seed=4)
    a = xz[:, :K]
    w = _rand(N, K, seed=5, scale=0.1)
    wide = torch.zeros(M, 2 * N, device=DEV)
    out = wide[:, N:]
    gemm.gemm_nt(a, w, out=out)
    want = a.double() @ w.double().t()
    _assert_close(out, want, _bound(a.double(), w.double().t()), "nt strided")
    assert float(wide[:, :N].abs().max()) == 0.0
    gemm.gemm_nt(a, w, out=out, accumulate=True)
    _assert_close(out, 2 * want, 2 * _bound(a.double(), w.double().t()), "nt accumulate")


@pytest.mark.parametrize("shape", [(19200, 1536, 384), (19200, 384, 768), (4800, 96, 192), (300, 1536, 768), (513, 40, 72), (7, 4, 4)],
                         ids=lambda s: "x".join(map(str, s)))
def test_gemm_nn_against_fp64(shape):
    """dx = dy @ W with W (N_out, K_in) read in place"""
    from sigma_amd import gemm
    M, Nout, Kin = shape
    dy, w = _rand(M, Nout, seed=6), _rand(Nout, Kin, seed=7, scale=0.05)
    got = gemm.gemm_nn(dy, w)
    want = dy.double() @ w.double()
    _assert_close(got, want, _bound(dy.double(), w.double()), "nn")


@pytest.mark.parametrize("shape", [(19200, 1536, 384), (19200, 384, 768), (76800, 384, 96), (4800, 96, 192), (300, 1536, 768),
                                   (1001, 68, 36), (33, 4, 4), (31, 128, 128)],
                         ids=lambda s: "x".join(map(str, s)))
def test_gemm_tn_against_fp64(shape):
    """dW = dy^T @ x: reduction over the token dimension, sliced over workgroups, fp32 atomics"""
    from sigma_amd import gemm
    M, Nout, Kin = shape
    dy, x = _rand(M, Nout, seed=8, scale=0.1), _rand(M, Kin, seed=9)
    got = gemm.gemm_tn(dy, x)
    want = dy.double().t() @ x.double()
    _assert_close(got, want, _bound(dy.double().t(), x.double()), "tn")
    acc = torch.ones(Nout, Kin, device=DEV)
    gemm.gemm_tn(dy, x, out=acc, accumulate=True)
    _assert_close(acc, want + 1.0, _bound(dy.double().t(), x.double()) + 1.0, "tn accumulate")


def test_linear_autograd_against_fp64_linear():
    """LinearSplit3Fn: forward, input gradient, weight gradient and bias gradient vs F.linear in fp64."""
    from sigma_amd import gemm
    M, K, N = 2400, 384, 768
    x = _rand(4, M // 4, K, seed=10).requires_grad_()
    w = _rand(N, K, seed=11, scale=0.05).requires_grad_()
    b = _rand(N, seed=12).requires_grad_()
    gy = _rand(4, M // 4, N, seed=13)
    y = gemm.linear(x, w, b)
    y.backward(gy)
    x64, w64, b64 = (t.detach().double().requires_grad_() for t in (x, w, b))
    y64 = torch.nn.functional.linear(x64, w64, b64)
    y64.backward(gy.double())
    for name, got, want in (("y", y, y64), ("dx", x.grad, x64.grad), ("dw", w.grad, w64.grad), ("db", b.grad, b64.grad)):
        rel = float((got.double() - want).abs().max() / want.abs().max())
        assert rel < 5e-5, f"{name}: {rel:.3e}"


def test_gemm_refuses_cpu_tensors_and_bad_shapes():
    from sigma_amd import gemm
    with pytest.raises(RuntimeError):
        gemm.gemm_nt(torch.randn(8, 8), torch.randn(8, 8))
    with pytest.raises(RuntimeError):
        gemm.gemm_nt(_rand(8, 6), _rand(8, 6))                       # K % 4 != 0
    with pytest.raises(RuntimeError):
        gemm.linear(torch.randn(2, 8), torch.randn(4, 8))


# ---- round 4: stacked problems (x_proj / dt_proj of the SS2D core), epilogue addends, outputs shared by problems --------
# (B, d, c, R, L): the four encoder stages of sigma_small at batch 2 (R = dt_rank, c = R + 2 N) and ragged ones
CORE_SHAPES = [(2, 768, 56, 24, 1200), (2, 384, 44, 12, 4800), (1, 1536, 80, 48, 300), (2, 192, 38, 6, 1200), (3, 40, 12, 4, 36)]


@pytest.mark.parametrize("dims", CORE_SHAPES, ids=["x".join(map(str, s)) for s in CORE_SHAPES])
def test_stacked_projections_of_the_scan_core_against_fp64(dims):
    """bgemm_nn with a weight stack shared by groups of problems and strided row-slice operands / outputs (p = W x,
    delta = W_dt p[:R], their input gradients, the last one with the scan's two du as epilogue addends), bgemm_nt_sum
    (weight gradients summed over the batch inside the kernel)."""
    from sigma_amd import gemm
    B, d, c, R, L = dims
    xs = _rand(B, 2, d, L, seed=1)
    Wst = _rand(2, 2 * c, d, seed=2, scale=0.05)
    p4 = torch.empty(B, 4, c, L, device=DEV)
    gemm.bgemm_nn(Wst, xs.view(2 * B, d, L), p4.view(2 * B, 2 * c, L))
    want = torch.matmul(Wst.double().unsqueeze(0), xs.double()).view(B, 4, c, L)
    bound = torch.matmul(Wst.double().abs().unsqueeze(0), xs.double().abs()).view(B, 4, c, L)
    _assert_close(p4, want, bound, "x_proj")
    if R % 4 == 0:
        dtw = _rand(4, d, R, seed=3, scale=0.2)
        delta = torch.full((B, 4, d, L), float("nan"), device=DEV)
        gemm.bgemm_nn(dtw, p4.view(4 * B, c, L)[:, :R], delta.view(4 * B, d, L))
        pr = p4[:, :, :R].double()
        _assert_close(delta, torch.matmul(dtw.double().unsqueeze(0), pr), torch.matmul(dtw.double().abs().unsqueeze(0), pr.abs()), "dt_proj")
        # input gradient of dt_proj written into the first R rows of every group of dp4, the others untouched
        dd = _rand(B, 4, d, L, seed=4)
        dp4 = torch.full((B, 4, c, L), 7.0, device=DEV)
        gemm.bgemm_nn(dtw.transpose(1, 2).contiguous(), dd.view(4 * B, d, L), dp4.view(4 * B, c, L)[:, :R])
        wt = dtw.double().transpose(1, 2).unsqueeze(0)
        _assert_close(dp4[:, :, :R], torch.matmul(wt, dd.double()), torch.matmul(wt.abs(), dd.double().abs()), "dt_proj dgrad")
        assert bool((dp4[:, :, R:] == 7.0).all())
        # weight gradient summed over the batch
        dW = torch.zeros(4, d, R, device=DEV)
        gemm.bgemm_nt_sum(dd.view(4 * B, d, L), p4.view(4 * B, c, L)[:, :R], dW)
        wantW = torch.matmul(dd.double(), pr.transpose(-1, -2)).sum(0)
        boundW = torch.matmul(dd.double().abs(), pr.abs().transpose(-1, -2)).sum(0)
        _assert_close(dW, wantW, boundW, "dt_proj wgrad")
    # x_proj input gradient + the scan's du of both directions of an order, one pass
    dp = _rand(B, 4, c, L, seed=5)
    du = _rand(B, 4, d, L, seed=6)
    du3 = du.view(2 * B, 2, d, L)
    dxs = torch.empty(B, 2, d, L, device=DEV)
    gemm.bgemm_nn(Wst.transpose(1, 2).contiguous(), dp.view(2 * B, 2 * c, L), dxs.view(2 * B, d, L), residual=du3[:, 0], residual2=du3[:, 1])
    wt = Wst.double().transpose(1, 2).unsqueeze(0)
    dp2 = dp.double().view(B, 2, 2 * c, L)
    want = torch.matmul(wt, dp2) + du.double().view(B, 2, 2, d, L).sum(2)
    bound = torch.matmul(wt.abs(), dp2.abs()) + du.double().abs().view(B, 2, 2, d, L).sum(2)
    _assert_close(dxs, want, bound, "x_proj dgrad + du")
    dWst = torch.zeros(2, 2 * c, d, device=DEV)
    gemm.bgemm_nt_sum(dp.view(2 * B, 2 * c, L), xs.view(2 * B, d, L), dWst)
    wantW = torch.matmul(dp2, xs.double().transpose(-1, -2)).sum(0)
    boundW = torch.matmul(dp2.abs(), xs.double().abs().transpose(-1, -2)).sum(0)
    _assert_close(dWst, wantW, boundW, "x_proj wgrad")


@pytest.mark.parametrize("shape", [(19200, 768, 384), (4800, 384, 192), (257, 100, 72), (130, 36, 200), (5, 4, 4)],
                         ids=lambda s: "x".join(map(str, s)))
def test_linear_with_the_residual_added_in_the_kernel(shape):
    """y = x W^T (+ b) + r with r entering through the accumulators: value, and the gradients of x, W, b and r"""
    from sigma_amd import gemm
    M, K, N = shape
    x, w, b, r = _rand(M, K, seed=1), _rand(N, K, seed=2, scale=0.05), _rand(N, seed=3), _rand(M, N, seed=4)
    for bias in (None, b):
        y = gemm.gemm_nt(x, w, bias, residual=r)
        want = x.double() @ w.double().t() + r.double() + (0 if bias is None else bias.double())
        _assert_close(y, want, _bound(x.double(), w.double().t()) + r.double().abs() + 1.0, "nt + residual")
    xa, wa, ba, ra = (t.clone().requires_grad_() for t in (x, w, b, r))
    g = _rand(M, N, seed=5)
    gemm.linear(xa.view(1, M, K), wa, ba, residual=ra.view(1, M, N)).backward(g.view(1, M, N))
    xr, wr, br, rr = (t.double().clone().requires_grad_() for t in (x, w, b, r))
    (torch.nn.functional.linear(xr, wr, br) + rr).backward(g.double())
    assert torch.equal(ra.grad, g)
    torch.testing.assert_close(ba.grad.double(), br.grad, rtol=1e-4, atol=1e-3)
    _assert_close(xa.grad, xr.grad, _bound(g.double().abs(), w.double().abs()), "dx")
    _assert_close(wa.grad, wr.grad, _bound(g.double().t().abs(), x.double().abs()), "dW")


def test_stacked_gemms_refuse_what_they_cannot_take():
    from sigma_amd import gemm
    a, b, o = _rand(2, 8, 16), _rand(4, 16, 24), torch.empty(4, 8, 24, device=DEV)
    gemm.bgemm_nn(a, b, o)
    with pytest.raises(RuntimeError):
        gemm.bgemm_nn(_rand(3, 8, 16), b, o)                       # 3 does not divide 4
    with pytest.raises(RuntimeError):
        gemm.bgemm_nn(a, _rand(4, 16, 22), torch.empty(4, 8, 22, device=DEV))     # N % 4
    with pytest.raises(RuntimeError):
        gemm.bgemm_nn(a.cpu(), b.cpu(), o.cpu())
    with pytest.raises(RuntimeError):
        gemm.bgemm_nn(a, b, o, residual2=o)
    with pytest.raises(RuntimeError):
        gemm.bgemm_nt_sum(_rand(4, 8, 16), _rand(4, 24, 16), torch.zeros(3, 8, 24, device=DEV))
    with pytest.raises(RuntimeError):
        gemm.gemm_nt(_rand(8, 16), _rand(24, 16), residual=_rand(8, 20))


# ---- round 6: in_proj with a channel-major x half (sigma_gemm.h t_cols), sliced reductions of the nn form ------------------

XZ_SHAPES = [(2, 30, 40, 384, 768), (2, 6, 10, 96, 192), (1, 15, 20, 768, 1536), (2, 23, 40, 128, 256), (1, 3, 4, 32, 32)]


@pytest.mark.parametrize("dims", XZ_SHAPES, ids=["x".join(map(str, d)) for d in XZ_SHAPES])
@pytest.mark.parametrize("with_bias", [False, True], ids=["nobias", "bias"])
def test_in_proj_with_channel_major_x_half_against_fp64(dims, with_bias):
    """gemm.linear_xz (vmamba.py:1067-1071: in_proj, chunk, permute + contiguous of the x half): ONE GEMM whose x columns
    leave the kernel transposed.  Forward and the gradients of x, weight and bias against the fp64 formulation; the x half is
    a (B, d, H, W) view of a channel-major (d, B, H, W) buffer whose planes are contiguous."""
    from sigma_amd import gemm
    B, H, W, C, d = dims
    x = _rand(B, H, W, C, seed=31).requires_grad_()
    w = _rand(2 * d, C, seed=32, scale=0.05).requires_grad_()
    b = _rand(2 * d, seed=33, scale=0.1).requires_grad_() if with_bias else None
    assert gemm.xz_ok(x.reshape(-1, C), w)
    xi, z = gemm.linear_xz(x, w, b)
    assert tuple(xi.shape) == (B, d, H, W) and xi.stride()[2:] == (W, 1) and xi.stride(1) == B * H * W and z.is_contiguous()
    x64, w64 = x.detach().double().requires_grad_(), w.detach().double().requires_grad_()
    b64 = b.detach().double().requires_grad_() if with_bias else None
    ref = torch.nn.functional.linear(x64, w64, b64)
    bound = _bound(x64.detach().reshape(-1, C), w64.detach().t())
    _assert_close(xi.permute(0, 2, 3, 1).reshape(-1, d), ref[..., :d].reshape(-1, d).detach(), bound[:, :d], "x half")
    _assert_close(z.reshape(-1, d), ref[..., d:].reshape(-1, d).detach(), bound[:, d:], "z half")
    gx, gz = _rand(*xi.shape, seed=34), _rand(*z.shape, seed=35)
    ((xi * gx).sum() + (z * gz).sum()).backward()
    ((ref[..., :d].permute(0, 3, 1, 2) * gx.double()).sum() + (ref[..., d:] * gz.double()).sum()).backward()
    g2 = torch.cat([gx.permute(0, 2, 3, 1).reshape(-1, d), gz.reshape(-1, d)], 1).double()
    _assert_close(x.grad.reshape(-1, C), x64.grad.reshape(-1, C), _bound(g2, w64.detach()), "dx")
    _assert_close(w.grad, w64.grad, _bound(g2.t(), x64.detach().reshape(-1, C)), "dw")
    if with_bias:
        torch.testing.assert_close(b.grad.double(), b64.grad, rtol=1e-4, atol=1e-4 * float(b64.grad.abs().max()))


def test_transposed_column_range_refuses_what_it_cannot_take():
    """t_cols needs 32-column granularity, M % 4 == 0, plain stores (no accumulate / residual): SIGMA_OPS_ERR_ARG otherwise"""
    from sigma_amd import gemm
    a, w = _rand(64, 32, seed=1), _rand(128, 32, seed=2)
    z, xt = torch.empty(64, 64, device=DEV), torch.empty(64, 64, device=DEV)
    ok = gemm._params(64, 128, 32, a, w, z, None, 32, 32, 64, out_t=xt, ldct=64, t_cols=64)
    gemm._run("sigma_gemm_nt_split3", ok, a.device)                                  # the legal call
    want = a.double() @ w.double().t()
    _assert_close(xt.t(), want[:, :64], _bound(a.double(), w.double().t())[:, :64], "transposed half")
    _assert_close(z, want[:, 64:], _bound(a.double(), w.double().t())[:, 64:], "plain half")
    for kw in (dict(t_cols=48), dict(t_cols=160), dict(ldct=60), dict(accumulate=True)):
        args = dict(out_t=xt, ldct=64, t_cols=64)
        acc = kw.pop("accumulate", False)
        args.update(kw)
        p = gemm._params(64, 128, 32, a, w, z, None, 32, 32, 64, acc, **args)
        with pytest.raises(RuntimeError, match="sigma_gemm_nt_split3 failed"):
            gemm._run("sigma_gemm_nt_split3", p, a.device)
    a2 = _rand(66, 32, seed=3)                                                        # M % 4 != 0
    p = gemm._params(66, 128, 32, a2, w, torch.empty(66, 64, device=DEV), None, 32, 32, 64, out_t=torch.empty(64, 68, device=DEV), ldct=68, t_cols=64)
    with pytest.raises(RuntimeError, match="sigma_gemm_nt_split3 failed"):
        gemm._run("sigma_gemm_nt_split3", p, a.device)


@pytest.mark.parametrize("shape", [(768, 19200, 384), (192, 4800, 96), (1536, 300, 768), (64, 4096, 32), (130, 1000, 36)],
                         ids=lambda s: "x".join(map(str, s)))
def test_gemm_nn_with_a_sliced_reduction_against_fp64(shape):
    """k_slices: few output tiles, a long reduction (dW_x = dx^T x with dx held channel-major): slices summed with atomics
    into a zero-filled output; also into a row-slice view of a larger gradient tensor, and accumulating"""
    from sigma_amd import gemm
    M, K, N = shape
    a, b = _rand(M, K, seed=41), _rand(K, N, seed=42, scale=0.05)
    want = a.double() @ b.double()
    bound = _bound(a.double(), b.double())
    _assert_close(gemm.gemm_nn(a, b, k_slices=True), want, bound, "nn sliced")
    big = torch.full((2 * M, N), 7.0, device=DEV)
    gemm.gemm_nn(a, b, out=big[M:], k_slices=True)
    _assert_close(big[M:], want, bound, "nn sliced into a view")
    assert float((big[:M] - 7.0).abs().max()) == 0.0
    acc = _rand(M, N, seed=43)
    got = gemm.gemm_nn(a, b, out=acc.clone(), accumulate=True, k_slices=True)
    _assert_close(got, want + acc.double(), bound + acc.double().abs(), "nn sliced accumulate")


@pytest.mark.parametrize("shape", [(19200, 1536, 384), (19200, 384, 768), (4800, 96, 192), (1001, 68, 36)],
                         ids=lambda s: "x".join(map(str, s)))
def test_sliced_reductions_sum_in_two_stages_without_a_zero_fill(shape):
    """Round 6: the reduction slices of a weight gradient are stored as partial results in scratch and summed by a second
    kernel (sigma_gemm_workspace_bytes / params.workspace): the output needs no zero fill (here it holds NaN), two runs
    agree bit for bit, `accumulate` adds, and a caller WITHOUT scratch still gets the atomic sums into a zero-filled C."""
    import ctypes
    from sigma_amd import _capi, gemm
    M, Nout, Kin = shape
    dy, x = _rand(M, Nout, seed=8, scale=0.1), _rand(M, Kin, seed=9)
    want = dy.double().t() @ x.double()
    bound = _bound(dy.double().t(), x.double())
    out = torch.full((Nout, Kin), float("nan"), device=DEV)
    gemm.gemm_tn(dy, x, out=out)
    _assert_close(out, want, bound, "tn into NaN")
    again = torch.full((Nout, Kin), float("nan"), device=DEV)
    gemm.gemm_tn(dy, x, out=again)
    assert torch.equal(out, again)                              # fixed summation order
    big = torch.full((2 * Nout, Kin), 3.0, device=DEV)          # a row-slice view of a larger gradient tensor
    gemm.gemm_tn(dy, x, out=big[Nout:])
    assert torch.equal(big[Nout:], out) and float((big[:Nout] - 3.0).abs().max()) == 0.0
    # the C ABI without scratch: atomics into a zero-filled C
    p = gemm._params(M, Nout, Kin, dy, x, out, None, dy.stride(0), x.stride(0), out.stride(0), False)
    need = int(_capi.load().sigma_gemm_workspace_bytes(ctypes.byref(p), 2))
    assert need >= 0
    out.zero_()
    with torch.cuda.device(dy.device):
        rc = _capi.load().sigma_gemm_tn_split3(ctypes.byref(p), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    _assert_close(out, want, bound, "tn atomics")
    if need > 0:                                                 # scratch one byte short: refused as scratch, atomics again
        ws = torch.empty(need, dtype=torch.uint8, device=DEV)
        p.workspace, p.workspace_bytes = ws.data_ptr(), need - 1
        out.zero_()
        with torch.cuda.device(dy.device):
            assert _capi.load().sigma_gemm_tn_split3(ctypes.byref(p), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
        _assert_close(out, want, bound, "tn short scratch")
        p.workspace, p.workspace_bytes = None, 16                # bytes without a pointer
        with torch.cuda.device(dy.device):
            assert _capi.load().sigma_gemm_tn_split3(ctypes.byref(p), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) != 0


def test_shared_outputs_sum_in_two_stages():
    """bgemm_nt_sum (weight gradients of the stacked projections summed over the batch): overwrite into NaN, accumulate,
    bit-identical repeats; a ragged problem count per output keeps the atomic path (zero-filled / accumulated C)"""
    from sigma_amd import gemm
    B, d, c, L = 8, 192, 16, 1200
    dp, xs = _rand(2 * B, 2 * c, L, seed=51), _rand(2 * B, d, L, seed=52)
    want = torch.matmul(dp.double(), xs.double().transpose(-1, -2)).view(B, 2, 2 * c, d).sum(0)
    bound = torch.matmul(dp.double().abs(), xs.double().abs().transpose(-1, -2)).view(B, 2, 2 * c, d).sum(0)
    out = torch.full((2, 2 * c, d), float("nan"), device=DEV)
    gemm.bgemm_nt_sum(dp, xs, out, accumulate=False)
    _assert_close(out, want, bound, "nt_sum overwrite")
    again = torch.full((2, 2 * c, d), float("nan"), device=DEV)
    gemm.bgemm_nt_sum(dp, xs, again, accumulate=False)
    assert torch.equal(out, again)
    acc = torch.ones(2, 2 * c, d, device=DEV)
    gemm.bgemm_nt_sum(dp, xs, acc)
    _assert_close(acc, want + 1.0, bound + 1.0, "nt_sum accumulate")
