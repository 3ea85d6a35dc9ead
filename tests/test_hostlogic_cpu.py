"""Host-side logic added in round 3, checked on CPU tensors (the HIP kernels behind it are covered by -m gpu tests):
layout helpers, stochastic depth folded into the branch, the applicability rules of the fused loss, the evaluator's
window grid, the GEMM wrappers' refusal of CPU tensors."""
import numpy as np
import pytest
import torch
import torch.nn as nn


def test_layout_helpers_on_cpu_are_the_torch_permutes():
    from sigma_amd.layout import channels_first, channels_last, transpose_rows
    x = torch.randn(2, 5, 7, 12)
    assert torch.equal(channels_first(x), x.permute(0, 3, 1, 2)) and channels_first(x).is_contiguous()
    y = torch.randn(2, 12, 5, 7)
    assert torch.equal(channels_last(y), y.permute(0, 2, 3, 1)) and channels_last(y).is_contiguous()
    s = torch.randn(3, 6, 20)
    h = s.split(10, dim=-1)[1]
    assert torch.equal(transpose_rows(h), h.transpose(1, 2)) and transpose_rows(h).is_contiguous()
    with pytest.raises(RuntimeError):
        transpose_rows(torch.randn(4, 4))


def test_drop_path_draw_is_the_mask_add_to_applies():
    import importlib
    vm = importlib.import_module("sigma_amd.models.encoders.vmamba")
    dp = vm.DropPath(0.4).train()
    x, r = torch.randn(16, 3, 4, 5), torch.randn(16, 3, 4, 5)
    torch.manual_seed(5)
    mask = dp.draw(x)
    torch.manual_seed(5)
    out = dp.add_to(r, x)
    assert tuple(mask.shape) == (16, 1, 1, 1)
    assert all(v == 0.0 or abs(v - 1 / 0.6) < 1e-5 for v in mask.flatten().tolist())
    torch.testing.assert_close(out, r + x * mask)
    assert dp.eval().draw(x) is None and vm.DropPath(0.0).train().draw(x) is None
    torch.testing.assert_close(dp.add_to(r, x), r + x)                      # eval: plain residual


def test_vss_and_cvss_blocks_fold_the_mask_into_the_branch(monkeypatch):
    """x + mask * op(norm(x)) (vmamba.py:1716-1722) and x * scale1 + mask * op(norm1(x)) (:1800-1802) with the mask handed
    to the branch: checked with the branch replaced by a linear map (the scan itself needs the GPU)."""
    import importlib
    vm = importlib.import_module("sigma_amd.models.encoders.vmamba")

    class Branch(nn.Module):
        def __init__(self, C):
            super().__init__()
            self.lin = nn.Linear(C, C, bias=False)

        def forward(self, x, branch_scale=None, residual=None):
            y = self.lin(x)
            y = y if branch_scale is None else y * branch_scale
            return y if residual is None else residual + y

    C = 8
    blk = vm.VSSBlock(hidden_dim=C, drop_path=0.5, d_state=4).train()
    blk.op = Branch(C)
    x = torch.randn(6, 3, 4, C)
    torch.manual_seed(2)
    got = blk(x)
    torch.manual_seed(2)
    mask = blk.drop_path.draw(x)
    torch.testing.assert_close(got, x + blk.op(blk.norm(x)) * mask)


    dec = vm.CVSSDecoderBlock(hidden_dim=96, drop_path=0.5, d_state=4).train()
    dec.op = Branch(96)
    with torch.no_grad():
        dec.scale1.copy_(torch.randn(96))
        dec.scale2.copy_(torch.randn(96))
    z = torch.randn(4, 6, 5, 96)
    torch.manual_seed(3)
    got = dec(z)
    torch.manual_seed(3)
    mask = dec.drop_path.draw(z)
    mid = z * dec.scale1 + dec.op(dec.norm1(z)) * mask
    want = mid * dec.scale2 + dec.conv_blk(dec.norm2(mid).permute(0, 3, 1, 2).contiguous()).permute(0, 2, 3, 1)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)


def test_fused_cross_entropy_only_takes_what_it_computes():
    from sigma_amd.pointwise import cross_entropy
    logits = torch.randn(1, 6, 6, 12).permute(0, 3, 1, 2)
    label = torch.zeros(1, 6, 6, dtype=torch.long)
    assert cross_entropy(nn.CrossEntropyLoss(), logits, label) is None                      # CPU tensors: not this path
    assert cross_entropy(nn.CrossEntropyLoss(reduction="sum"), logits, label) is None
    assert cross_entropy(nn.CrossEntropyLoss(weight=torch.ones(12)), logits, label) is None
    assert cross_entropy(nn.CrossEntropyLoss(label_smoothing=0.1), logits, label) is None
    assert cross_entropy(nn.MSELoss(), logits, label) is None


def test_builder_forward_uses_the_criterion_when_the_fused_loss_declines():
    """EncoderDecoder.forward (models/builder.py:146-157): loss = criterion(logits, label) -- on CPU tensors through the
    criterion itself, whatever it is."""
    from sigma_amd.models.builder import EncoderDecoder

    class Stub(EncoderDecoder):
        def __init__(self):
            nn.Module.__init__(self)
            self.criterion = nn.CrossEntropyLoss(reduction="mean", ignore_index=255)

        def encode_decode(self, rgb, modal_x):
            return rgb[:, :3] * 2.0

    m = Stub()
    rgb = torch.randn(2, 3, 4, 5)
    label = torch.randint(0, 3, (2, 4, 5))
    torch.testing.assert_close(m(rgb, rgb, label), nn.functional.cross_entropy(rgb * 2.0, label))
    assert m(rgb, rgb).is_contiguous() and torch.equal(m(rgb, rgb), rgb * 2.0)


def test_evaluator_window_grid_reproduces_the_reference_arithmetic():
    """engine/evaluator.py:464-478, including its use of stride[0] / crop_size[0] for the column direction"""
    from sigma_amd.engine.evaluator_ops import window_grid
    crop, rate, rows, cols = (256, 256), 2 / 3, 600, 800
    stride = (int(np.ceil(crop[0] * rate)), int(np.ceil(crop[1] * rate)))
    r_grid = int(np.ceil((rows - crop[0]) / stride[0])) + 1
    c_grid = int(np.ceil((cols - crop[1]) / stride[1])) + 1
    ref = []
    for gy in range(r_grid):                       # the reference's loop, literally
        for gx in range(c_grid):
            s_x, s_y = gx * stride[0], gy * stride[1]
            e_x, e_y = min(s_x + crop[0], cols), min(s_y + crop[1], rows)
            s_x, s_y = e_x - crop[0], e_y - crop[1]
            ref.append((s_y, e_y, s_x, e_x))
    wins = window_grid(rows, cols, crop, rate)
    assert wins == ref and len(wins) == r_grid * c_grid
    covered = np.zeros((rows, cols), dtype=bool)
    for s_y, e_y, s_x, e_x in wins:
        covered[s_y:e_y, s_x:e_x] = True
    assert covered.all()
    # a non-square crop on an image that is shorter than crop[1] (NYU at scale 1.25): the reference's mixed indices give a
    # negative start, which its numpy / torch slicing wraps to the end of the axis -- reproduced, since a drop-in must
    # score the same pixels (tests/test_evaluator_oracle.py has the window list)
    img = np.arange(600 * 800).reshape(600, 800)
    for (s_y, e_y, s_x, e_x) in window_grid(600, 800, (480, 640), 2 / 3):
        assert s_y >= 0 and np.array_equal(img[s_y:e_y, s_x:e_x], img[e_y - 640:e_y, e_x - 480:e_x])


def test_gemm_and_gate_wrappers_refuse_cpu_tensors():
    from sigma_amd import gemm
    from sigma_amd.pointwise import channel_gate
    a, b = torch.randn(8, 16), torch.randn(4, 16)
    for fn in (gemm.gemm_nt, gemm.gemm_tn):
        with pytest.raises(RuntimeError):
            fn(a, b if fn is gemm.gemm_nt else torch.randn(8, 4))
    with pytest.raises(RuntimeError):
        gemm.linear(torch.randn(2, 3, 16), b)
    with pytest.raises(RuntimeError):
        channel_gate(torch.randn(1, 8, 3, 3), torch.randn(2, 8, 1, 1), torch.randn(8, 2, 1, 1))


def test_gradient_buffer_hand_off_matches_only_the_registered_half():
    """sigma_amd/_handoff.py: claimed once, by the z half itself; clones, other geometries and stale entries do not match"""
    from sigma_amd import _handoff as h
    h._XZ_GRAD_BUFFERS.clear()
    full = torch.zeros(2, 3, 4, 8)
    h.offer_xz_grad_buffer(full, 4)
    dz = full[..., 4:]
    assert h.claim_xz_grad_buffer(dz.clone(), (2, 3, 4, 8)) is None          # another tensor: no match, entry kept
    assert h.claim_xz_grad_buffer(dz, (2, 3, 4, 8)) is full
    assert h.claim_xz_grad_buffer(dz, (2, 3, 4, 8)) is None                  # claimed once
    h.offer_xz_grad_buffer(full, 4)
    assert h.claim_xz_grad_buffer(dz, (2, 3, 2, 16)) is None                 # geometry mismatch consumes and refuses
    for _ in range(3 * h._XZ_GRAD_KEEP):
        h.offer_xz_grad_buffer(torch.zeros(1, 1, 2, 8), 4)
    assert len(h._XZ_GRAD_BUFFERS) <= h._XZ_GRAD_KEEP
    assert h.claim_xz_grad_buffer(None, (1, 1, 2, 8)) is None


def test_derived_parameter_cache_follows_the_parameter_version():
    """ss2d_fused._derived_params: permuted / transposed weight copies and -exp(A_logs), cached per parameter version --
    hit while nothing changed, rebuilt after an in-place update, a data swap, or for another parameter object"""
    from sigma_amd import ss2d_fused as f
    f._DERIVED.clear()
    K, c, d, R, N = 4, 6, 8, 2, 4
    xw, dw, al = (nn.Parameter(torch.randn(K, c, d)), nn.Parameter(torch.randn(K, d, R)), nn.Parameter(torch.randn(K * d, N)))
    a = f._derived_params(xw, dw, al)
    Wst, WstT, dtw, A = a
    perm = [0, 2, 1, 3]
    torch.testing.assert_close(Wst, xw.detach()[perm].reshape(2, 2 * c, d))
    torch.testing.assert_close(WstT, Wst.transpose(1, 2))
    torch.testing.assert_close(dtw, dw.detach()[perm])
    torch.testing.assert_close(A, -torch.exp(al.detach()))
    assert WstT.is_contiguous() and dtw.is_contiguous() and not any(t.requires_grad for t in a)
    b = f._derived_params(xw, dw, al)
    assert all(x is y for x, y in zip(a, b))                                  # hit
    with torch.no_grad():
        al.mul_(0.5)                                                           # what an optimizer step does: version bump
    c2 = f._derived_params(xw, dw, al)
    assert c2[3] is not A
    torch.testing.assert_close(c2[3], -torch.exp(al.detach()))
    xw.data = torch.randn(K, c, d)                                             # no version bump, another storage
    d2 = f._derived_params(xw, dw, al)
    torch.testing.assert_close(d2[0], xw.detach()[perm].reshape(2, 2 * c, d))
    xw2 = nn.Parameter(xw.detach().clone())
    e2 = f._derived_params(xw2, dw, al)
    assert e2[0] is not d2[0]
    assert all(x is y for x, y in zip(d2, f._derived_params(xw, dw, al)))      # the first one is still cached
    # an optimizer that owns one of the parameters drops the entry when it steps (fused AdamW bumps no version counter);
    # one that does not own them (the reference's groups) leaves it alone
    kept = f._derived_params(xw, dw, al)
    other = nn.Parameter(torch.randn(3))
    other.grad = torch.ones(3)
    torch.optim.AdamW([other], lr=0.1).step()
    assert f._derived_params(xw, dw, al)[3] is kept[3]
    al.grad = torch.ones_like(al)
    try:
        opt = torch.optim.AdamW([al], lr=0.1, fused=True)
    except Exception:
        opt = torch.optim.AdamW([al], lr=0.1)
    opt.step()
    fresh = f._derived_params(xw, dw, al)
    assert fresh[3] is not kept[3]
    torch.testing.assert_close(fresh[3], -torch.exp(al.detach()))
    d2 = fresh
    # a write through .data is invisible to the key (documented): the caller invalidates
    al.data.mul_(2.0)
    assert f._derived_params(xw, dw, al)[3] is d2[3]
    f.invalidate_derived_params()
    torch.testing.assert_close(f._derived_params(xw, dw, al)[3], -torch.exp(al.detach()))
