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
