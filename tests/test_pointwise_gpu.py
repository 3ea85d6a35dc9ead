"""GPU parity of csrc/pointwise.hip (ChannelAttention gate, softmax cross entropy) and of the tiled-transpose layout
changes (sigma_amd/layout.py) against the torch formulations of the reference -- run with -m gpu."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref_gate(x, w1, w2):
    """ChannelAttention.forward of the reference (vmamba.py:1725-1741) with plain torch ops under torch's own autograd.
    The two bias-free 1x1 convolutions act on (2B, C, 1, 1) tensors, i.e. they ARE matrix products of the (2B, C) pooled
    rows: written as F.linear so that this checker does not send a 1 x 1-pixel convolution through MIOpen (no tuned
    solution exists for it: first-use fallback paths of a vendor library inside the one process that runs the whole GPU
    suite -- the round-4 suite aborted in the backward of exactly this expression)."""
    B = x.shape[0]
    pooled = torch.cat([x.mean(dim=(2, 3)), x.amax(dim=(2, 3))], dim=0)                       # (2B, C)
    g = F.linear(F.silu(F.linear(pooled, w1.flatten(1))), w2.flatten(1))
    return x * torch.sigmoid(g[:B] + g[B:])[:, :, None, None]


@pytest.mark.parametrize("shape,sq", [((2, 96, 30, 40), 30), ((1, 60, 7, 9), 30), ((3, 32, 1, 1), 16), ((2, 192, 15, 20), 30),
                                      ((1, 96, 120, 160), 30)])
def test_channel_gate_matches_torch(shape, sq):
    from sigma_amd.pointwise import channel_gate
    B, C, H, W = shape
    g = torch.Generator().manual_seed(5)
    x = torch.randn(shape, generator=g).cuda()
    w1 = (torch.randn(max(C // sq, 1), C, 1, 1, generator=g) * 0.3).cuda()
    w2 = (torch.randn(C, max(C // sq, 1), 1, 1, generator=g) * 0.3).cuda()
    gy = torch.randn(shape, generator=g).cuda()
    outs = []
    for fn in (channel_gate, _ref_gate):
        a, b1, b2 = x.clone().requires_grad_(), w1.clone().requires_grad_(), w2.clone().requires_grad_()
        y = fn(a, b1, b2)
        y.backward(gy)
        outs.append((y.detach(), a.grad, b1.grad, b2.grad))
    for got, want, name in zip(outs[0], outs[1], ("y", "dx", "dw1", "dw2")):
        torch.testing.assert_close(got, want, rtol=2e-5, atol=2e-5 * float(want.abs().max()) + 1e-7, msg=lambda m, n=name: f"{n}: {m}")


def test_channel_gate_shares_the_max_gradient_between_ties():
    """torch.amax semantics: tied maxima share the gradient of the pooled maximum"""
    from sigma_amd.pointwise import channel_gate
    g = torch.Generator().manual_seed(6)
    x = torch.randn(1, 32, 6, 8, generator=g)
    x[0, 3, 1, 2] = x[0, 3, 4, 5] = 9.0                       # two maxima in plane 3
    x[0, 7].fill_(0.25)                                       # a constant plane: 48 ties
    x = x.cuda()
    w1 = (torch.randn(2, 32, 1, 1, generator=g) * 0.3).cuda()
    w2 = (torch.randn(32, 2, 1, 1, generator=g) * 0.3).cuda()
    gy = torch.randn(1, 32, 6, 8, generator=g).cuda()
    grads = []
    for fn in (channel_gate, _ref_gate):
        a = x.clone().requires_grad_()
        fn(a, w1, w2).backward(gy)
        grads.append(a.grad)
    torch.testing.assert_close(grads[0], grads[1], rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("rows_shape,nc", [((2, 30, 40), 40), ((1, 7, 9), 12), ((8, 480, 640), 40), ((1, 1, 3), 4), ((2, 9, 11), 64),
                                           ((1, 5, 6), 72)])
def test_softmax_cross_entropy_matches_torch(rows_shape, nc):
    from sigma_amd.pointwise import cross_entropy
    B, H, W = rows_shape
    g = torch.Generator().manual_seed(7)
    nhwc = (torch.randn(B, H, W, nc, generator=g) * 3).cuda()
    label = torch.randint(0, nc, (B, H, W), generator=g)
    label[torch.rand(B, H, W, generator=g) < 0.1] = 255
    label = label.cuda()
    crit = nn.CrossEntropyLoss(reduction="mean", ignore_index=255)
    a = nhwc.clone().requires_grad_()
    loss = cross_entropy(crit, a.permute(0, 3, 1, 2), label)
    assert loss is not None
    (loss * 1.7).backward()
    b = nhwc.clone().requires_grad_()
    want = crit(b.permute(0, 3, 1, 2), label)
    (want * 1.7).backward()
    torch.testing.assert_close(loss, want, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(a.grad, b.grad, rtol=1e-4, atol=1e-6 * float(b.grad.abs().max()) + 1e-12)
    # run-to-run identical (fixed partial layout, no atomics)
    again = cross_entropy(crit, nhwc.permute(0, 3, 1, 2), label)
    assert torch.equal(again, loss.detach())


def test_softmax_cross_entropy_declines_what_it_does_not_take():
    from sigma_amd.pointwise import cross_entropy
    x = torch.randn(1, 6, 6, 12).cuda()
    lab = torch.zeros(1, 6, 6, dtype=torch.long).cuda()
    assert cross_entropy(nn.CrossEntropyLoss(reduction="none"), x.permute(0, 3, 1, 2), lab) is None
    assert cross_entropy(nn.CrossEntropyLoss(), x.permute(0, 3, 1, 2).contiguous(), lab) is None       # NCHW logits
    assert cross_entropy(nn.CrossEntropyLoss(), torch.randn(1, 6, 6, 9).cuda().permute(0, 3, 1, 2), lab) is None   # 9 classes


@pytest.mark.parametrize("shape", [(2, 30, 40, 96), (1, 7, 9, 20), (3, 1, 5, 4), (2, 120, 160, 192)])
def test_tiled_layout_changes_match_permute(shape):
    from sigma_amd.layout import channels_first, channels_last, transpose_rows
    B, H, W, C = shape
    g = torch.Generator().manual_seed(8)
    x = torch.randn(shape, generator=g).cuda()
    gy = torch.randn(B, C, H, W, generator=g).cuda()
    a = x.clone().requires_grad_()
    y = channels_first(a)
    assert y.is_contiguous() and torch.equal(y, x.permute(0, 3, 1, 2))
    y.backward(gy)
    assert torch.equal(a.grad, gy.permute(0, 2, 3, 1))
    b = y.detach().clone().requires_grad_()
    z = channels_last(b)
    assert z.is_contiguous() and torch.equal(z, x)
    z.backward(x)
    assert torch.equal(b.grad, x.permute(0, 3, 1, 2))
    # one half of a longer sequence (ConMB: the RGB and the X tokens of the concatenated scan), gradient of a transposed view
    seq = torch.randn(B, C, 2 * H * W, generator=g).cuda().requires_grad_()
    h1, h2 = seq.split(H * W, dim=-1)
    t = transpose_rows(h2)
    assert t.is_contiguous() and torch.equal(t, h2.transpose(1, 2))
    t.backward(torch.ones_like(t).transpose(1, 2).contiguous().transpose(1, 2))
    assert torch.equal(seq.grad[..., H * W:], torch.ones(B, C, H * W).cuda()) and float(seq.grad[..., :H * W].abs().max()) == 0.0


@pytest.mark.parametrize("shape,gated", [((4, 6, 10, 96), True), ((3, 5, 7, 384), True), ((2, 9, 4, 768), False), ((5, 1, 3, 32), True)])
def test_layernorm_with_per_sample_factor_matches_torch(shape, gated):
    """sigma_layernorm_params.row_scale (include/sigma_ops.h): (LayerNorm(x) [* silu(z)]) * mask[b] and its backward, with z
    the second half of a (..., 2C) tensor whose gradient buffer the backward completes (dgate_row_stride)."""
    from sigma_amd.layernorm import LayerNorm, LayerNormFn
    B, H, W, C = shape
    g = torch.Generator().manual_seed(11)
    x = torch.randn(shape, generator=g).cuda()
    xz = torch.randn(B, H, W, 2 * C, generator=g).cuda()
    mask = (torch.rand(B, 1, 1, 1, generator=g) < 0.7).float().div_(0.7).cuda()
    gy = torch.randn(shape, generator=g).cuda()
    ln = LayerNorm(C).cuda()
    with torch.no_grad():
        ln.weight.copy_(torch.randn(C, generator=g))
        ln.bias.copy_(torch.randn(C, generator=g))
    res = []
    for hip in (True, False):
        a, b = x.clone().requires_grad_(), xz.clone().requires_grad_()
        z = b[..., C:]
        ln.zero_grad()
        if hip:
            y = ln.forward_gated(a, z, mask) if gated else LayerNormFn.apply(a, ln.weight, ln.bias, ln.eps, None, mask)
        else:
            y = F.layer_norm(a, (C,), ln.weight, ln.bias, ln.eps)
            y = (y * F.silu(z) if gated else y) * mask
        y.backward(gy)
        res.append([y.detach(), a.grad, ln.weight.grad.clone(), ln.bias.grad.clone()] + ([b.grad] if gated else []))
    for got, want, name in zip(res[0], res[1], ("y", "dx", "dgamma", "dbeta", "dxz")):
        torch.testing.assert_close(got, want, rtol=2e-5, atol=3e-5 * float(want.abs().max()) + 1e-7, msg=lambda m, n=name: f"{n}: {m}")


def test_vss_block_with_stochastic_depth_equals_masked_branch():
    """VSSBlock in training mode: x + mask[b] / keep * SS2D(LN(x)) (vmamba.py:1716-1722) with the mask folded into the
    gated LayerNorm pass equals the same branch multiplied after out_proj (same random draw)."""
    import importlib
    vm = importlib.import_module("sigma_amd.models.encoders.vmamba")
    torch.manual_seed(3)
    blk = vm.VSSBlock(hidden_dim=32, drop_path=0.4, d_state=4).cuda().train()
    x = torch.randn(6, 8, 12, 32).cuda()
    gy = torch.randn(6, 8, 12, 32).cuda()
    outs = []
    for folded in (True, False):
        torch.manual_seed(17)
        a = x.clone().requires_grad_()
        blk.zero_grad()
        if folded:
            y = blk(a)
        else:
            mask = blk.drop_path.draw(a)
            y = a + blk.op(blk.norm(a)) * mask
        y.backward(gy)
        outs.append([y.detach(), a.grad] + [p.grad.clone() for p in blk.parameters()])
    for i, (got, want) in enumerate(zip(outs[0], outs[1])):
        torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-5 * float(want.abs().max()) + 1e-7, msg=lambda m, i=i: f"tensor {i}: {m}")


@pytest.mark.parametrize("n_outer,inner", [(32, 768 * 300), (3, 1001), (1, 4), (5, 4 * 1024 * 7 + 8)])
def test_pair_sum_add_matches_torch(n_outer, inner):
    """sigma_pair_sum_add (include/sigma_ops.h): acc[o] += src[2o] + src[2o + 1]"""
    from sigma_amd.ss2d_fused import _pair_sum_add
    g = torch.Generator().manual_seed(13)
    src = torch.randn(n_outer, 2, inner, generator=g).cuda()
    acc = torch.randn(n_outer, inner, generator=g).cuda()
    want = acc + src[:, 0] + src[:, 1]
    _pair_sum_add(src, acc, n_outer, inner)
    torch.testing.assert_close(acc, want, rtol=0, atol=1e-6)


@pytest.mark.parametrize("shape", [(4, 30, 40, 384), (2, 7, 9, 96), (3, 5, 1, 2048), (1, 1, 1, 4)], ids=lambda s: "x".join(map(str, s)))
def test_layernorm_that_hands_its_input_through_joins_the_residual_gradient(shape):
    """LayerNorm.forward_with_pass (round 4): (LN(x), x) with d x = LN backward + the gradient of the handed-through
    tensor, added inside the kernel (sigma_layernorm_params.dx_add) -- against torch's two nodes and its add"""
    from sigma_amd.layernorm import LayerNorm
    C = shape[-1]
    g = torch.Generator().manual_seed(C)
    ln = LayerNorm(C).cuda()
    with torch.no_grad():
        ln.weight.copy_(torch.randn(C, generator=g))
        ln.bias.copy_(torch.randn(C, generator=g))
    x = torch.randn(shape, generator=g).cuda()
    gy, gp = torch.randn(shape, generator=g).cuda(), torch.randn(shape, generator=g).cuda()
    a = x.clone().requires_grad_()
    y, p = ln.forward_with_pass(a)
    assert p.data_ptr() == a.data_ptr()
    (y * gy + p * gp).sum().backward()
    b = x.clone().requires_grad_()
    ln2 = torch.nn.LayerNorm(C).cuda()
    ln2.load_state_dict(ln.state_dict())
    (ln2(b) * gy + b * gp).sum().backward()
    torch.testing.assert_close(y, ln2(b), rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(a.grad, b.grad, rtol=2e-5, atol=3e-5 * float(b.grad.abs().max()))
    torch.testing.assert_close(ln.weight.grad, ln2.weight.grad, rtol=1e-4, atol=1e-4 * float(ln2.weight.grad.abs().max()))
    torch.testing.assert_close(ln.bias.grad, ln2.bias.grad, rtol=1e-4, atol=1e-4 * float(ln2.bias.grad.abs().max()))
    # only the handed-through tensor used / only the normalised one used
    c = x.clone().requires_grad_()
    _, p = ln.forward_with_pass(c)
    (p * gp).sum().backward()
    torch.testing.assert_close(c.grad, gp)
    d = x.clone().requires_grad_()
    y, _ = ln.forward_with_pass(d)
    (y * gy).sum().backward()
    e = x.clone().requires_grad_()
    (ln2(e) * gy).sum().backward()
    torch.testing.assert_close(d.grad, e.grad, rtol=2e-5, atol=3e-5 * float(e.grad.abs().max()))
    with torch.no_grad():                                    # no graph: the plain path, the tensor itself
        y, p = ln.forward_with_pass(x)
        assert p is x


@pytest.mark.parametrize("mode", ["single", "second_consumer", "hook"])
def test_in_proj_gradient_buffer_hand_off_takes_the_copying_path_when_z_has_company(mode):
    """split_xz + gated LayerNorm: the LayerNorm backward writes dz into the z half of one (B, H, W, 2d) buffer that
    SplitXZFn.backward completes in place (sigma_amd/_handoff.py).  With a second consumer of z, or a hook on z, the
    gradient reaching SplitXZFn is another tensor: the hand-off must not match and the result must still be right."""
    import torch.nn.functional as F
    from sigma_amd.layernorm import LayerNorm
    from sigma_amd.ss2d_fused import split_xz
    from sigma_amd import _handoff
    B, H, W, d = 2, 6, 10, 64
    g = torch.Generator().manual_seed(21)
    ln = LayerNorm(d).cuda()
    xz0 = torch.randn(B, H, W, 2 * d, generator=g).cuda()
    t0 = torch.randn(B, H, W, d, generator=g).cuda()
    gy, gx, gz = (torch.randn(s, generator=g).cuda() for s in ((B, H, W, d), (B, d, H, W), (B, H, W, d)))
    res = []
    for own in (True, False):
        xz, t = xz0.clone().requires_grad_(), t0.clone().requires_grad_()
        if own:
            x, z = split_xz(xz)
            if mode == "hook":
                z.register_hook(lambda gr: gr * 2.0)
            y = ln.forward_gated(t, z)
        else:
            x, z = xz[..., :d].permute(0, 3, 1, 2), xz[..., d:]
            y = F.layer_norm(t, (d,), ln.weight, ln.bias, ln.eps) * F.silu(z)
        loss = (y * gy).sum() + (x * gx).sum()
        if mode == "second_consumer":
            loss = loss + (z * gz).sum()
        ln.zero_grad()
        loss.backward()
        res.append((xz.grad.clone(), t.grad.clone()))
    want_xz = res[1][0].clone()
    if mode == "hook":
        want_xz[..., d:] *= 2.0
    torch.testing.assert_close(res[0][0], want_xz, rtol=2e-5, atol=2e-5 * float(want_xz.abs().max()))
    torch.testing.assert_close(res[0][1], res[1][1], rtol=2e-5, atol=2e-5 * float(res[1][1].abs().max()))
    assert len(_handoff._XZ_GRAD_BUFFERS) <= _handoff._XZ_GRAD_KEEP


@pytest.mark.parametrize("shape", [(2, 12, 16, 96), (8, 30, 40, 384), (1, 7, 9, 192), (3, 5, 5, 1024), (2, 3, 4, 4), (1, 1, 1, 768)],
                         ids=lambda s: "x".join(map(str, s)))
def test_scale_residual_backward_in_one_pass_matches_addcmul(shape):
    """pointwise.scale_residual (sigma_colscale_bwd): a + x * scale of the decoder block (vmamba.py:1800-1805) -- value and
    the three gradients against torch.addcmul's autograd"""
    from sigma_amd.pointwise import scale_residual, scale_residual_ok
    C = shape[-1]
    g = torch.Generator().manual_seed(C + shape[0])
    a0, x0, s0, gy = (torch.randn(sh, generator=g).cuda() for sh in (shape, shape, (C,), shape))
    res = []
    for fn in (scale_residual, torch.addcmul):
        a, x, s = a0.clone().requires_grad_(), x0.clone().requires_grad_(), s0.clone().requires_grad_()
        assert scale_residual_ok(a, x, s)
        y = fn(a, x, s)
        y.backward(gy)
        res.append((y.detach(), a.grad, x.grad, s.grad))
    for got, want, name in zip(res[0], res[1], ("y", "da", "dx", "dscale")):
        torch.testing.assert_close(got, want, rtol=2e-5, atol=2e-5 * float(want.abs().max()) + 1e-7, msg=lambda m, n=name: f"{n}: {m}")
    # what the kernel does not take goes to addcmul: odd channel counts, no graph
    y = scale_residual(torch.randn(2, 3, 5).cuda(), torch.randn(2, 3, 5).cuda(), torch.randn(5).cuda())
    assert y.shape == (2, 3, 5)
