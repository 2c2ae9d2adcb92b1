"""-m gpu numerics: the MFMA implicit-GEMM convolution (C ABI oadg_conv2d_nhwc_bf16) against a plain PyTorch
fp32 convolution of the same bf16-rounded operands.  fp32 accumulation on both sides: the difference is the
bf16 rounding of the output (<= 2^-8 relative) plus summation order."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref(x, w, b, stride, pad, dil, res=None, relu=False):
    y = F.conv2d(x.float(), w.float(), b, stride, pad, dil)
    if res is not None:
        y = y + res.float()
    return F.relu(y) if relu else y


@pytest.mark.parametrize('N,C,H,W,K,R,stride,pad,dil', [
    (2, 256, 32, 48, 256, 3, 1, 1, 1),      # FPN / RPN 3x3
    (1, 64, 17, 23, 128, 3, 1, 1, 1),       # ragged M tail
    (2, 512, 20, 28, 128, 1, 1, 0, 1),      # bottleneck 1x1
    (1, 128, 33, 31, 128, 3, 2, 1, 1),      # stride-2 3x3
    (1, 512, 16, 16, 512, 3, 1, 2, 2),      # dilated (DC5)
    (1, 256, 14, 18, 1024, 1, 2, 0, 1),     # stride-2 1x1 downsample
    (2, 64, 20, 28, 64, 3, 1, 1, 1),        # 64 output channels: the 128 x 64 tile (ResNet stage 1)
    (1, 256, 17, 23, 64, 1, 1, 0, 1),
    (1, 64, 19, 21, 192, 1, 1, 0, 1),       # K = 3 x 64
])
def test_conv_forward_matches_fp32_reference(dev, N, C, H, W, K, R, stride, pad, dil):
    from oadg_amd import hip_conv
    g = torch.Generator(device=dev).manual_seed(C + K + R)
    x = torch.randn(N, C, H, W, device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(K, C, R, R, device=dev, generator=g) / np.sqrt(C * R * R)).bfloat16() \
        .contiguous(memory_format=torch.channels_last)
    b = torch.randn(K, device=dev, generator=g)
    ref = _ref(x, w, b, stride, pad, dil)
    for variant in (0, 1, 3):            # automatic choice, 128-tile with two LDS stages, 128-tile with one stage
        y = hip_conv.conv_forward(x, w, b, None, stride, pad, dil, False, variant=variant)
        assert y.shape == ref.shape
        err = (y.float() - ref).abs().max().item()
        assert err <= 8e-3 * ref.abs().max().item(), (variant, err)
    # asymmetric check against transposes: a single non-zero weight tap / channel
    w2 = torch.zeros_like(w)
    w2[K - 3, 5, R - 1, 0] = 1.0
    y2 = hip_conv.conv_forward(x, w2, None, None, stride, pad, dil, False)
    assert torch.equal(y2.float(), _ref(x, w2, None, stride, pad, dil).bfloat16().float())


@pytest.mark.parametrize('N,C,H,W,K,R,stride,pad,dil', [
    (2, 256, 32, 48, 256, 3, 1, 1, 1),      # FPN / RPN 3x3
    (1, 64, 17, 23, 256, 3, 1, 1, 1),       # ragged pixel tail, a single 64-channel chunk per tap
    (3, 128, 20, 28, 512, 1, 1, 0, 1),      # 1x1, two column tiles, two K-tiles only
    (1, 64, 9, 11, 256, 1, 1, 0, 1),        # one K-tile: prologue + tail staging only
    (1, 128, 33, 31, 256, 3, 2, 1, 1),      # stride 2
    (1, 512, 16, 16, 512, 3, 1, 2, 2),      # dilated (DC5)
    (1, 128, 64, 96, 256, 3, 2, 1, 1),      # stride 2, full tiles
    (2, 64, 48, 80, 256, 3, 1, 1, 1),       # one chunk per tap, 30 tiles
    (2, 128, 40, 56, 128, 3, 1, 1, 1),      # K = 128: the 256 x 128 instantiation (ResNet layer2's 3x3)
    (1, 128, 37, 29, 128, 3, 2, 1, 1),      # ... stride 2, ragged pixel tail
    (1, 64, 24, 24, 384, 1, 1, 0, 1),       # ... three 128-channel column tiles
])
def test_conv_256_tile_variant_matches_fp32_reference(dev, N, C, H, W, K, R, stride, pad, dil):
    """the phase-pipelined 256x256 kernel (variant 2) on its own: same contract, same tolerance; repeated to
    screen for staging races (results must be bit-identical from run to run and equal to the 128-tile kernel
    up to the summation order inside an MFMA chain, i.e. the same bf16 rounding tolerance)."""
    from oadg_amd import hip_conv
    g = torch.Generator(device=dev).manual_seed(C + K + R + 1)
    x = torch.randn(N, C, H, W, device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(K, C, R, R, device=dev, generator=g) / np.sqrt(C * R * R)).bfloat16() \
        .contiguous(memory_format=torch.channels_last)
    b = torch.randn(K, device=dev, generator=g)
    res = torch.randn(N, K, (H + 2 * pad - dil * (R - 1) - 1) // stride + 1,
                      (W + 2 * pad - dil * (R - 1) - 1) // stride + 1, device=dev, generator=g).bfloat16() \
        .contiguous(memory_format=torch.channels_last)
    y = hip_conv.conv_forward(x, w, b, None, stride, pad, dil, False, variant=2)
    ref = _ref(x, w, b, stride, pad, dil)
    assert y.shape == ref.shape
    err = (y.float() - ref).abs().max().item()
    assert err <= 8e-3 * ref.abs().max().item(), err
    for _ in range(5):
        assert torch.equal(y, hip_conv.conv_forward(x, w, b, None, stride, pad, dil, False, variant=2))
    y2 = hip_conv.conv_forward(x, w, b, res, stride, pad, dil, True, variant=2)
    ref2 = _ref(x, w, b, stride, pad, dil, res, True)
    assert (y2.float() - ref2).abs().max().item() <= 8e-3 * ref2.abs().max().item()
    w2 = torch.zeros_like(w)
    w2[K - 3, 5, R - 1, 0] = 1.0
    y3 = hip_conv.conv_forward(x, w2, None, None, stride, pad, dil, False, variant=2)
    assert torch.equal(y3.float(), _ref(x, w2, None, stride, pad, dil).bfloat16().float())


def test_conv_fused_epilogue(dev):
    from oadg_amd import hip_conv
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(2, 128, 24, 40, device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(256, 128, 3, 3, device=dev, generator=g) / 34).bfloat16().contiguous(memory_format=torch.channels_last)
    b = torch.randn(256, device=dev, generator=g)
    res = torch.randn(2, 256, 24, 40, device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    y = hip_conv.conv_forward(x, w, b, res, 1, 1, 1, True)
    ref = _ref(x, w, b, 1, 1, 1, res, True)
    assert (y.float() - ref).abs().max().item() <= 8e-3 * ref.abs().max().item()
    assert (y >= 0).all()


def test_conv_autograd_matches_reference(dev):
    from oadg_amd import hip_conv
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(2, 256, 20, 28, device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(256, 256, 3, 3, device=dev, generator=g) / 48).bfloat16()
    b = torch.randn(256, device=dev, generator=g)
    gy = torch.randn(2, 256, 20, 28, device=dev, generator=g).bfloat16()
    xa, wa, ba = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    res = torch.randn(2, 256, 20, 28, device=dev, generator=g).bfloat16()
    ra = res.clone().requires_grad_(True)
    y = hip_conv.conv2d(xa, wa, ba, 1, 1, 1, relu=True, residual=ra)
    assert y is not None
    (y.float() * gy.float()).sum().backward()
    xr, wr, br = x.float().requires_grad_(True), w.float().requires_grad_(True), b.clone().requires_grad_(True)
    rr = res.float().requires_grad_(True)
    # same ReLU mask on both sides (taken from the kernel's bf16 output): pre-activations within one bf16 ulp of
    # zero would otherwise flip and dominate the comparison
    mask = (y.detach() > 0).float()
    ((F.conv2d(xr, wr, br, 1, 1, 1) + rr) * mask * gy.float()).sum().backward()
    for a, r, tol in ((xa.grad, xr.grad, 2e-2), (wa.grad, wr.grad, 3e-2), (ba.grad, br.grad, 2e-2),
                      (ra.grad, rr.grad, 2e-2)):
        # the ReLU mask is taken from the bf16 output: a handful of pre-activations within one bf16 ulp of 0
        # flip, so compare in the mean as well as in the max
        d = (a.float() - r).abs()
        assert d.mean().item() <= 1e-2 * r.abs().mean().item() + 1e-6, (d.mean().item(), r.abs().mean().item())
        assert d.max().item() <= 10 * tol * r.abs().max().item(), (d.max().item(), r.abs().max().item())


@pytest.mark.parametrize('N,C,H,W,K,R,stride,pad,dil', [
    (2, 256, 20, 28, 256, 3, 1, 1, 1),
    (1, 128, 19, 21, 128, 3, 1, 1, 1),      # ragged pixel tail
    (2, 512, 16, 24, 128, 1, 1, 0, 1),
    (1, 128, 33, 31, 256, 3, 2, 1, 1),
    (1, 512, 16, 16, 512, 3, 1, 2, 2),
    # >= 64 pixel chunks and K, C multiples of 256: the 256-tile phase-pipelined kernel
    (2, 256, 48, 67, 256, 3, 1, 1, 1),      # ragged pixel tail (6432 pixels)
    (2, 512, 40, 56, 256, 1, 1, 0, 1),
    (4, 256, 65, 67, 256, 3, 2, 1, 1),
    (2, 256, 48, 48, 512, 3, 1, 2, 2),
    (8, 256, 64, 128, 256, 3, 1, 1, 1),     # FPN P4 / layer3 conv2 at the bench size: one round of 252 workgroups (28 splits, contiguous XCD slices)
    (8, 512, 32, 64, 512, 3, 1, 1, 1),      # layer4 conv2: 36 tiles x 7 splits
    (2, 256, 320, 320, 256, 1, 1, 0, 1),    # 1x1 over >= 200k pixels (lateral P2 form) on the 256-tile kernel
    (4, 256, 128, 130, 512, 3, 2, 1, 1),    # stride 2, ragged, 18 tiles x 14 splits
    (8, 256, 256, 512, 512, 1, 2, 0, 1),    # layer2 downsample 1x1 / stride 2 at the bench size
    # round 4, row-strip K-tile order (output width a multiple of 64): splits own whole rows, 64-column strips
    (2, 256, 128, 256, 256, 3, 1, 1, 1),    # P3 form: four strips per row, 256 rows over 28 splits (ragged: 10 / 6 rows)
    (3, 256, 50, 128, 256, 3, 1, 1, 1),     # two strips, rows crossing image boundaries inside a split
    (2, 256, 128, 256, 256, 3, 2, 1, 1),    # stride 2: 64 x 128 outputs of a 128 x 256 input
    (2, 256, 96, 64, 256, 3, 1, 2, 2),      # dilation 2, one strip per row
])
def test_conv_wgrad_matches_fp32_reference(dev, N, C, H, W, K, R, stride, pad, dil):
    from oadg_amd import hip_conv
    g = torch.Generator(device=dev).manual_seed(C + K + R + stride)
    x = torch.randn(N, C, H, W, device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    Ho = (H + 2 * pad - dil * (R - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (R - 1) - 1) // stride + 1
    gy = torch.randn(N, K, Ho, Wo, device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    dw = hip_conv.conv_wgrad(x, gy, K, R, R, stride, pad, dil)
    assert torch.equal(dw, hip_conv.conv_wgrad(x, gy, K, R, R, stride, pad, dil))     # deterministic, race screen
    w = torch.zeros(K, C, R, R, device=dev, requires_grad=True)
    (F.conv2d(x.float(), w, None, stride, pad, dil) * gy.float()).sum().backward()
    assert dw.shape == w.grad.shape
    err = (dw - w.grad).abs().max().item()
    assert err <= 2e-3 * w.grad.abs().max().item(), (err, w.grad.abs().max().item())


@pytest.mark.parametrize('w_channels_last', [False, True])
def test_prepared_weights_fold_bn_forward_backward(dev, w_channels_last):
    """csrc prep_weights kernels: BN fold + layouts and their backward against the unfused torch expression, for a
    contiguous and for a torch.channels_last parameter (read and differentiated in place, gradient in its strides)."""
    import torch.nn as nn
    from oadg_amd import hip_conv
    g = torch.Generator(device=dev).manual_seed(5)
    K, C, R = 128, 64, 3
    w = torch.randn(K, C, R, R, device=dev, generator=g) / 24
    if w_channels_last:
        w = w.contiguous(memory_format=torch.channels_last)
    w.requires_grad_(True)
    bn = nn.BatchNorm2d(K).to(dev).eval()
    with torch.no_grad():
        bn.weight.copy_(torch.rand(K, device=dev, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(K, device=dev, generator=g))
        bn.running_mean.copy_(torch.randn(K, device=dev, generator=g))
        bn.running_var.copy_(torch.rand(K, device=dev, generator=g) + 0.5)
    wf, b, wt = hip_conv.prepared(w, bn, None, True)
    scale = bn.weight.detach() * torch.rsqrt(bn.running_var + bn.eps)
    wref = w.detach() * scale.view(-1, 1, 1, 1)
    assert (wf.float() - wref).abs().max() <= 4e-3 * wref.abs().max()
    assert torch.allclose(b, bn.bias.detach() - bn.running_mean * scale, rtol=1e-6, atol=1e-6)
    assert torch.equal(wt, wf.detach().flip(2, 3).transpose(0, 1).contiguous(memory_format=torch.channels_last))
    gw = torch.randn(K, C, R, R, device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    gb = torch.randn(K, device=dev, generator=g)
    torch.autograd.backward([wf, b], [gw, gb])
    w2 = w.detach().clone().requires_grad_(True)
    gam, bet = bn.weight.detach().clone().requires_grad_(True), bn.bias.detach().clone().requires_grad_(True)
    s2 = gam * torch.rsqrt(bn.running_var + bn.eps)
    torch.autograd.backward([w2 * s2.view(-1, 1, 1, 1), bet - bn.running_mean * s2], [gw.float(), gb])
    assert torch.allclose(w.grad, w2.grad, rtol=1e-5, atol=1e-6)
    assert w.grad.stride() == w.stride()
    assert torch.allclose(bn.weight.grad, gam.grad, rtol=1e-4, atol=1e-4)
    assert torch.allclose(bn.bias.grad, bet.grad, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize('N,K,H,W,f32,mask', [
    (2, 256, 37, 53, False, True),      # bottleneck / RPN relu gradient
    (1, 64, 19, 21, False, True),
    (2, 2048, 9, 11, False, True),      # one channel group per thread
    (1, 4096, 5, 7, False, True),       # more channel groups than threads
    (2, 24, 33, 17, False, True),       # thread count not a multiple of the group count
    (2, 256, 40, 48, True, False),      # fp32 dy (RoIAlign backward output): cast + bias gradient
    (2, 512, 16, 24, True, True),
    (3, 256, 31, 29, False, False),     # bias gradient only, nothing written
])
def test_relu_bias_bwd_matches_torch(dev, N, K, H, W, f32, mask):
    """g bit-exact with torch's cast + threshold_backward; dbias within fp32 summation-order error."""
    from oadg_amd import hip_conv
    gen = torch.Generator(device=dev).manual_seed(K + H)
    dy = torch.randn(N, K, H, W, device=dev, generator=gen).contiguous(memory_format=torch.channels_last)
    if not f32:
        dy = dy.bfloat16()
    y = torch.relu(torch.randn(N, K, H, W, device=dev, generator=gen)).bfloat16() \
        .contiguous(memory_format=torch.channels_last) if mask else None
    g, db = hip_conv.relu_bias_bwd(dy, y, True)
    ref = dy.to(torch.bfloat16)
    if mask:
        ref = torch.ops.aten.threshold_backward(ref, y, 0)
    assert g.dtype == torch.bfloat16 and g.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(g, ref)
    ref_db = ref.double().sum((0, 2, 3))
    scale = ref.double().abs().sum((0, 2, 3)).max().item()
    assert (db.double() - ref_db).abs().max().item() <= 1e-6 * scale
    # deterministic: same bits on a second call
    assert torch.equal(db, hip_conv.relu_bias_bwd(dy, y, True)[1])


@pytest.mark.parametrize('N,C,H,W,Ht,Wt', [(2, 256, 32, 48, 16, 24), (1, 64, 25, 31, 13, 16), (2, 8, 7, 9, 3, 4)])
def test_fpn_topdown_matches_torch(dev, N, C, H, W, Ht, Wt):
    """fused lat + nearest_upsample(top) and its backward (fpn.py:166-175) against torch in fp32 on the same bf16
    operands: one bf16 rounding of the result."""
    from oadg_amd import hip_ops
    gen = torch.Generator(device=dev).manual_seed(H)
    mk = lambda *s: torch.randn(*s, device=dev, generator=gen).bfloat16().contiguous(  # noqa: E731
        memory_format=torch.channels_last)
    lat, top, g = mk(N, C, H, W), mk(N, C, Ht, Wt), mk(N, C, H, W)
    lat.requires_grad_(True)
    top.requires_grad_(True)
    out = hip_ops.fpn_topdown(lat, top)
    out.backward(g)
    lf, tf = lat.detach().float().requires_grad_(True), top.detach().float().requires_grad_(True)
    ref = lf + F.interpolate(tf, size=(H, W), mode='nearest')
    ref.backward(g.float())
    assert torch.equal(out, ref.bfloat16())
    assert torch.equal(lat.grad, g)
    assert torch.equal(top.grad, tf.grad.bfloat16())


def test_grad_tokens_fold_relu_and_identity_backward(dev, monkeypatch):
    """Three bottlenecks with GradTokens (mask / bias gradient / identity add inside the data-gradient epilogues)
    against the same blocks with every epilogue backward as a standalone pass: same gradients up to one bf16
    rounding per tensor, and the dgrad-epilogue options against torch directly."""
    from oadg_amd import hip_conv
    from oadg_amd.backbones import make_res_layer
    hip_conv.enable(True)
    try:
        torch.manual_seed(0)
        layer = make_res_layer(512, 128, 3, 1, 1, 'pytorch', dict(type='BN', requires_grad=True)).to(dev)
        layer.eval()                       # norm_eval: frozen statistics, trainable affine
        for m in layer.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                torch.nn.init.uniform_(m.weight, 0.5, 1.5)
                torch.nn.init.normal_(m.bias, 0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.running_mean.normal_(0, 0.2)
        g = torch.Generator(device=dev).manual_seed(1)
        x0 = torch.randn(2, 512, 24, 40, device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
        gy = torch.randn(2, 512, 24, 40, device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
        res = {}
        for mode in ('tokens', 'plain'):
            if mode == 'plain':
                monkeypatch.setattr(hip_conv, 'tokens_ok', lambda *a, **k: False)
            layer.zero_grad(set_to_none=True)
            x = torch.relu(x0).clone().requires_grad_(True)
            with torch.autocast('cuda', dtype=torch.bfloat16):
                y = layer(x)
            y.backward(gy)
            res[mode] = (y.detach().float(), x.grad.float(), {n: p.grad.float().clone() for n, p in layer.named_parameters()})
        assert torch.equal(res['tokens'][0], res['plain'][0])
        def close(a, b, what):
            d = (a - b).abs()
            assert d.max().item() <= 3e-2 * b.abs().max().item() + 1e-6, (what, d.max().item(), b.abs().max().item())
            assert d.mean().item() <= 5e-3 * b.abs().mean().item() + 1e-7, (what, d.mean().item(), b.abs().mean().item())
        close(res['tokens'][1], res['plain'][1], 'x.grad')
        for n in res['plain'][2]:
            close(res['tokens'][2][n], res['plain'][2][n], n)
    finally:
        hip_conv.enable(False)


def test_conv_dgrad_epilogue_mask_and_colsum(dev):
    from oadg_amd import hip_conv
    g = torch.Generator(device=dev).manual_seed(2)
    for variant, (N, C, H, W, K) in ((1, (2, 128, 20, 28, 128)), (2, (2, 128, 40, 56, 256)), (2, (1, 64, 19, 23, 512)),
                                      (2, (2, 128, 32, 48, 256)), (2, (2, 128, 33, 47, 128)),
                                      (1, (2, 128, 21, 27, 64)), (3, (2, 128, 20, 28, 128)), (3, (2, 256, 21, 27, 64))):
        x = torch.randn(N, C, H, W, device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
        w = (torch.randn(K, C, 3, 3, device=dev, generator=g) / (C * 9) ** 0.5).bfloat16().contiguous(
            memory_format=torch.channels_last)
        res = torch.randn(N, K, H, W, device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
        mask = torch.relu(torch.randn(N, K, H, W, device=dev, generator=g)).bfloat16().contiguous(
            memory_format=torch.channels_last)
        y, cs = hip_conv.conv_forward(x, w, None, res, 1, 1, 1, False, variant=variant, mask=mask, want_colsum=True)
        plain = hip_conv.conv_forward(x, w, None, None, 1, 1, 1, False, variant=variant)
        ref = ((plain.float() + res.float()) * (mask > 0)).bfloat16()
        assert torch.equal(y, ref)
        rs = ref.double().sum((0, 2, 3))
        assert (cs.double() - rs).abs().max().item() <= 1e-5 * ref.double().abs().sum((0, 2, 3)).max().item()
        y2, cs2 = hip_conv.conv_forward(x, w, None, res, 1, 1, 1, False, variant=variant, mask=mask, want_colsum=True)
        assert torch.equal(cs, cs2) and torch.equal(y, y2)
        # the same mask as one bit per element (what the producer's forward launch leaves behind, round 2): bits_out of
        # a ReLU launch equals (y > 0) packed LSB-first over 8 consecutive channels, and mask_bits reproduces mask
        bits = torch.full((N * H * W * K // 8,), 0xAA, dtype=torch.uint8, device=dev)
        yr = hip_conv.conv_forward(x, w, None, res if variant != 1 else None, 1, 1, 1, True, variant=variant, bits_out=bits)
        flags = (yr.permute(0, 2, 3, 1).reshape(-1, 8) > 0).to(torch.uint8)
        packed = (flags << torch.arange(8, device=dev, dtype=torch.uint8)).sum(1).to(torch.uint8)
        assert torch.equal(bits, packed)
        y3, cs3 = hip_conv.conv_forward(x, w, None, res, 1, 1, 1, False, variant=variant, mask=yr, want_colsum=True)
        y4, cs4 = hip_conv.conv_forward(x, w, None, res, 1, 1, 1, False, variant=variant, mask_bits=bits, want_colsum=True)
        assert torch.equal(y3, y4) and torch.equal(cs3, cs4)


@pytest.mark.parametrize('N,C,H,W,K', [
    (4, 64, 128, 256, 256),       # layer1 conv3 form: one workgroup column, 8 sub-tiles per workgroup
    (4, 64, 128, 264, 256),       # uneven: some workgroups run 9 sub-tiles, some 8
    (2, 128, 128, 256, 512),      # layer2 conv3 / conv1 data gradient: two columns share a pixel range
    (2, 256, 64, 128, 1024),      # layer3: 16-pixel sub-tiles, four columns
    (2, 256, 64, 64, 2048),       # eight columns on 64 pixel ranges (exactly 8 sub-tiles each)
    (4, 64, 128, 128, 512),       # 64 input channels (128-byte pixel rows), two columns
    (4, 512, 128, 128, 128),      # round 4, the read-heavy mirror (32 output channels per wave): layer2 conv1 form, one column
    (4, 512, 64, 128, 256),       # two columns of 128 channels share a pixel range
    (2, 512, 32, 64, 2048),       # layer4 conv3 form: sixteen columns on 32 pixel ranges (8 sub-tiles each)
    (9, 512, 72, 128, 128),       # uneven: 5184 sub-tiles over 512 workgroups
])
def test_streaming_pointwise_kernel_equals_the_tile_kernel(dev, N, C, H, W, K):
    """csrc conv_pw_stream_kernel (variant 4: weights in registers, LDS-DMA pixel ring, operands one sub-tile ahead)
    against the 128-tile kernel (variant 3) on every operand combination its callers use - same products, same fp32
    summation order: outputs and mask bits bit-identical, column sums equal up to the order of the partial rows -
    and against the fp32 reference convolution."""
    from oadg_amd import hip_conv, _lib
    L = _lib.lib()
    assert L.oadg_conv2d_auto_variant(N, H, W, C, K, 1, 1, 1, 0, 1) == 4
    g = torch.Generator(device=dev).manual_seed(N * C + K)
    cl = dict(memory_format=torch.channels_last)
    x = torch.randn(N, C, H, W, device=dev, generator=g).bfloat16().contiguous(**cl)
    w = (torch.randn(K, C, 1, 1, device=dev, generator=g) / C ** 0.5).bfloat16().contiguous(**cl)
    b = torch.randn(K, device=dev, generator=g)
    res = torch.randn(N, K, H, W, device=dev, generator=g).bfloat16().contiguous(**cl)
    flags = torch.rand(N * H * W * K // 8, device=dev, generator=g)
    bits = (flags * 256).to(torch.uint8)

    def run(variant, **kw):
        return hip_conv.conv_forward(x, w, kw.pop('bias', None), kw.pop('res', None), 1, 0, 1, kw.pop('relu', False),
                                     variant=variant, **kw)
    ref = F.conv2d(x.float(), w.float(), b)
    y4 = run(4, bias=b)
    assert (y4.float() - ref).abs().max().item() <= 8e-3 * ref.abs().max().item()
    for kw in (dict(), dict(bias=b, relu=True), dict(bias=b, res=res, relu=True)):
        assert torch.equal(run(4, **kw), run(3, **kw)), kw
    bo3, bo4 = (torch.full_like(bits, 0xAA) for _ in range(2))
    assert torch.equal(run(4, bias=b, res=res, relu=True, bits_out=bo4), run(3, bias=b, res=res, relu=True, bits_out=bo3))
    assert torch.equal(bo3, bo4)
    for kw in (dict(mask_bits=bits), dict(res=res, mask_bits=bits)):
        y3, c3 = run(3, want_colsum=True, **kw)
        y4, c4 = run(4, want_colsum=True, **kw)
        assert torch.equal(y3, y4)
        assert (c3 - c4).abs().max().item() <= 1e-4 * y3.float().abs().sum((0, 2, 3)).max().item()
        assert torch.equal(run(4, want_colsum=True, **kw)[1], c4)              # deterministic
    assert L.oadg_conv2d_pixel_tiles(N, H, W, C, K, 1, 1, 1, 0, 1, 4) == 512 // (K // (128 if C >= 512 else 256))   # one row per pixel range
    # the automatic choice falls back to the tile kernel for operands the streaming kernel does not take
    ym = hip_conv.conv_forward(x, w, None, None, 1, 0, 1, False, mask=res)
    assert torch.equal(ym, hip_conv.conv_forward(x, w, None, None, 1, 0, 1, False, variant=3, mask=res))


def test_stage_output_gradients_deposited_in_dgrad_epilogues(dev, monkeypatch):
    """A stage output feeds the next stage's conv1, its stride-2 downsample convolution and the FPN lateral.  With
    hip_conv.DEPOSIT the lateral and the downsample leave their data gradients on the tensor's GradToken (each adding
    what is there in its own epilogue; the 1x1 / stride-2 one in place on its strided grid) and conv1 finishes the sum
    with the ReLU mask and the bias-gradient column sums - against autograd's accumulation passes (DEPOSIT off) and
    against the plain path without any token: same outputs, parameter gradients within bf16 rounding of each other."""
    from oadg_amd import hip_conv
    from oadg_amd.backbones import ResNet
    from oadg_amd.necks import FPN
    hip_conv.enable(True)
    try:
        torch.manual_seed(0)
        bb = ResNet(depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                    norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, style='pytorch')
        neck = FPN(in_channels=[256, 512, 1024, 2048], out_channels=256, num_outs=5)
        neck.init_weights()
        net = torch.nn.ModuleList([bb, neck]).to(dev).to(memory_format=torch.channels_last)
        net.train()
        for m in bb.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                torch.nn.init.uniform_(m.weight, 0.5, 1.5)
                torch.nn.init.normal_(m.bias, 0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.running_mean.normal_(0, 0.2)
        g = torch.Generator(device=dev).manual_seed(1)
        x = torch.randn(2, 3, 128, 192, device=dev, generator=g).contiguous(memory_format=torch.channels_last)
        gys = None
        res = {}
        deposits = []
        orig = hip_conv.conv_dgrad_s2
        monkeypatch.setattr(hip_conv, 'conv_dgrad_s2',
                            lambda *a, **k: (deposits.append(k.get('accumulate') is not None), orig(*a, **k))[1])
        for mode in ('deposit', 'accumulate', 'plain'):
            monkeypatch.setattr(hip_conv, 'DEPOSIT', mode == 'deposit')
            if mode == 'plain':
                monkeypatch.setattr(hip_conv, 'tokens_ok', lambda *a, **k: False)
            net.zero_grad(set_to_none=True)
            with torch.autocast('cuda', dtype=torch.bfloat16):
                outs = neck(bb(x))
            if gys is None:
                gys = [torch.randn(o.shape, device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
                       for o in outs]
            torch.autograd.backward(list(outs), gys)
            res[mode] = ([o.detach().float() for o in outs],
                         {n: p.grad.float().clone() for n, p in net.named_parameters() if p.grad is not None})
        assert sum(deposits[:len(deposits) // 3]) == 2          # layer3 / layer4 downsample: in-place deposits
        assert set(res['deposit'][1]) == set(res['plain'][1])
        for a, b in zip(res['deposit'][0], res['plain'][0]):
            assert torch.equal(a, b)

        def close(a, b, what, tol):
            d = (a - b).abs()
            assert d.max().item() <= tol * b.abs().max().item() + 1e-6, (what, d.max().item(), b.abs().max().item())
            assert d.mean().item() <= tol / 4 * b.abs().mean().item() + 1e-7, (what, d.mean().item(), b.abs().mean().item())
        for n in res['plain'][1]:
            close(res['deposit'][1][n], res['accumulate'][1][n], n, 3e-2)
            close(res['deposit'][1][n], res['plain'][1][n], n, 6e-2)
    finally:
        hip_conv.enable(False)


def test_fpn_level_gradients_finished_by_the_rpn_convolution(dev, monkeypatch):
    """An FPN output level is read by the RPN convolution and by RoIAlign.  With hip_conv.DEPOSIT RoIAlign's backward
    leaves its gradient map on the level's GradToken, the RPN convolution's data gradient adds it in its epilogue and
    hands the FPN convolution its bias gradient as column sums (no ReLU mask: GradToken(masked=False)); against
    autograd's accumulation (DEPOSIT off): same parameter gradients up to bf16 rounding of the sums, and the level the
    extra level is subsampled from (a third consumer) keeps the plain path."""
    from oadg_amd import hip_conv, hip_ops
    from oadg_amd.necks import FPN
    from oadg_amd.dense_heads import RPNHead
    hip_conv.enable(True)
    try:
        torch.manual_seed(0)
        neck = FPN(in_channels=[256, 512, 1024, 2048], out_channels=256, num_outs=5)
        neck.init_weights()
        head = RPNHead(in_channels=256, feat_channels=256,
                       anchor_generator=dict(type='AnchorGenerator', scales=[8], ratios=[0.5, 1.0, 2.0],
                                             strides=[4, 8, 16, 32, 64]),
                       bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[0.0] * 4, target_stds=[1.0] * 4),
                       loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                       loss_bbox=dict(type='L1Loss', loss_weight=1.0))
        head.init_weights()
        net = torch.nn.ModuleList([neck, head]).to(dev).to(memory_format=torch.channels_last)
        g = torch.Generator(device=dev).manual_seed(1)
        H, W = 128, 192
        feats = [torch.randn(2, c, H // s, W // s, device=dev, generator=g).bfloat16().contiguous(
            memory_format=torch.channels_last).requires_grad_(True) for c, s in zip((256, 512, 1024, 2048), (4, 8, 16, 32))]
        rs = np.random.RandomState(0)
        x1 = rs.uniform(0, W - 40, 96); y1 = rs.uniform(0, H - 40, 96)
        rois = torch.tensor(np.stack([rs.randint(0, 2, 96), x1, y1, x1 + rs.uniform(4, 120, 96), y1 + rs.uniform(4, 100, 96)], 1),
                            dtype=torch.float32, device=dev)
        res = {}
        for mode in ('deposit', 'accumulate'):
            monkeypatch.setattr(hip_conv, 'DEPOSIT', mode == 'deposit')
            net.zero_grad(set_to_none=True)
            for f in feats:
                f.grad = None
            with torch.autocast('cuda', dtype=torch.bfloat16):
                outs = neck(feats)
                toks = [getattr(o, '_oadg_token', None) for o in outs]
                cls, reg = head(outs)
                pooled = hip_ops.roi_align_fpn(list(outs[:4]), rois, 7, [1 / 4., 1 / 8., 1 / 16., 1 / 32.])
            loss = sum((c.float() ** 2).mean() + (r.float() ** 2).mean() for c, r in zip(cls, reg)) + \
                (pooled.float() ** 2).mean() * 50
            loss.backward()
            res[mode] = ({n: p.grad.float().clone() for n, p in net.named_parameters()}, [f.grad.float().clone() for f in feats],
                         toks)
        toks = res['deposit'][2]
        assert [t is not None for t in toks] == [True, True, True, False, False]     # P5 feeds P6: plain path
        assert all(t.closed and t.extra is None for t in toks[:3])

        def close(a, b, what):
            d = (a - b).abs()
            assert d.max().item() <= 3e-2 * b.abs().max().item() + 1e-6, (what, d.max().item(), b.abs().max().item())
            assert d.mean().item() <= 8e-3 * b.abs().mean().item() + 1e-7, (what, d.mean().item(), b.abs().mean().item())
        for n in res['accumulate'][0]:
            close(res['deposit'][0][n], res['accumulate'][0][n], n)
        for i, (a, b) in enumerate(zip(res['deposit'][1], res['accumulate'][1])):
            close(a, b, f'C{i + 2}')
    finally:
        hip_conv.enable(False)


def test_rpn_head_fused_cls_reg_matches_separate_convs(dev):
    """rpn_cls + rpn_reg as one zero-padded 1x1 conv on the MFMA kernel (with the GradToken hand-off to rpn_conv)
    against the module-by-module path through the library convolutions: outputs and every parameter gradient."""
    from oadg_amd import hip_conv
    from oadg_amd.dense_heads import RPNHead
    torch.manual_seed(0)
    head = RPNHead(in_channels=256, feat_channels=256,
                   anchor_generator=dict(type='AnchorGenerator', scales=[8], ratios=[0.5, 1.0, 2.0], strides=[4]),
                   loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                   loss_bbox=dict(type='L1Loss', loss_weight=1.0)).to(dev)
    for m in (head.rpn_conv, head.rpn_cls, head.rpn_reg):
        torch.nn.init.normal_(m.weight, 0, 0.05)
        torch.nn.init.normal_(m.bias, 0, 0.1)
    g = torch.Generator(device=dev).manual_seed(1)
    x0 = torch.randn(2, 256, 40, 56, device=dev, generator=g).contiguous(memory_format=torch.channels_last)
    gc = torch.randn(2, 3, 40, 56, device=dev, generator=g)
    gr = torch.randn(2, 12, 40, 56, device=dev, generator=g)
    res = {}
    for mode in ('fused', 'separate'):
        hip_conv.enable(mode == 'fused')
        try:
            head.zero_grad(set_to_none=True)
            x = x0.clone().requires_grad_(True)
            with torch.autocast('cuda', dtype=torch.bfloat16):
                cls, reg = head.forward_single(x)
            assert cls.shape == gc.shape and reg.shape == gr.shape
            ((cls.float() * gc).sum() + (reg.float() * gr).sum()).backward()
            res[mode] = (cls.detach().float(), reg.detach().float(), x.grad.float(),
                         {n: p.grad.float().clone() for n, p in head.named_parameters()})
        finally:
            hip_conv.enable(False)
    # the two paths run rpn_conv on different kernels: bf16 rounding flips the ReLU mask of pre-activations next to
    # zero, which moves single elements of everything upstream of the mask by a full term (both paths are equally
    # far from an fp32 run, tools/probe/rpn_debug.py) - compare those in the mean, the rest tightly
    def close(a, b, what, tol_max, tol_mean):
        d = (a - b).abs()
        assert d.max().item() <= tol_max * b.abs().max().item() + 1e-6, (what, d.max().item(), b.abs().max().item())
        assert d.mean().item() <= tol_mean * b.abs().mean().item() + 1e-7, (what, d.mean().item(), b.abs().mean().item())
    close(res['fused'][0], res['separate'][0], 'cls', 3e-2, 8e-3)
    close(res['fused'][1], res['separate'][1], 'reg', 3e-2, 8e-3)
    close(res['fused'][2], res['separate'][2], 'x.grad', 0.25, 6e-2)
    for n in res['separate'][3]:
        upstream = n.startswith('rpn_conv')
        close(res['fused'][3][n], res['separate'][3][n], n, 0.25 if upstream else 3e-2, 8e-2 if upstream else 1e-2)


def test_shared_weight_gradients_accumulate(dev):
    """A prepared weight used by two convolution calls (the RPN conv is shared by the pyramid levels): the
    WeightGradToken hand-off must step aside (autograd has to add the two gradients) and the result must equal the
    fp32 reference; a single use takes the fused partial-sum path - both are checked."""
    from oadg_amd import hip_conv
    hip_conv.enable(True)
    try:
        g = torch.Generator(device=dev).manual_seed(4)
        conv = torch.nn.Conv2d(128, 128, 3, padding=1).to(dev)
        xs = [torch.randn(2, 128, h, w, device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
              for h, w in ((24, 40), (12, 20))]
        gys = [torch.randn(2, 128, h, w, device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
               for h, w in ((24, 40), (12, 20))]
        for uses in (2, 1):
            conv.zero_grad(set_to_none=True)
            with torch.autocast('cuda', dtype=torch.bfloat16):
                wf, b, wt = hip_conv.prepared(conv.weight, None, conv.bias, False)
                ys = [hip_conv._Conv2dMFMA.apply(x, wf, b, None, wt, 1, 1, 1, False, None, None, None) for x in xs[:uses]]
            torch.autograd.backward(ys, gys[:uses])
            w32 = conv.weight.detach().bfloat16().float().requires_grad_(True)
            b32 = conv.bias.detach().clone().requires_grad_(True)
            refs = [F.conv2d(x.float(), w32, b32, 1, 1) for x in xs[:uses]]
            torch.autograd.backward(refs, [gy.float() for gy in gys[:uses]])
            for got, ref in ((conv.weight.grad, w32.grad), (conv.bias.grad, b32.grad)):
                assert (got - ref).abs().max().item() <= 1e-2 * ref.abs().max().item(), uses
    finally:
        hip_conv.enable(False)


@pytest.mark.parametrize('N,C,H,W', [(2, 64, 64, 96), (1, 64, 37, 53), (1, 8, 5, 7), (2, 128, 2, 2)])
def test_bias_relu_maxpool_is_the_unfused_chain(dev, N, C, H, W):
    """csrc/eltwise.hip bias_relu_maxpool_kernel == max_pool2d(relu(x + b)) in bf16, bit for bit (stem tail)."""
    import torch.nn.functional as F
    from oadg_amd import hip_ops
    g = torch.Generator(device=dev).manual_seed(3)
    x = (torch.randn(N, C, H, W, device=dev, generator=g) * 2).bfloat16().contiguous(memory_format=torch.channels_last)
    b = torch.randn(C, device=dev, generator=g)
    y = x + b.bfloat16().view(1, -1, 1, 1)
    assert y.dtype == torch.bfloat16
    ref = F.max_pool2d(F.relu(y), 3, 2, 1)
    out = hip_ops.bias_relu_maxpool(x, b)
    assert out.shape == ref.shape and out.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(out, ref)
    assert torch.equal(hip_ops.bias_relu_maxpool(x, None), F.max_pool2d(F.relu(x), 3, 2, 1))


def test_resnet_stem_fused_tail_matches_module_chain(dev):
    from oadg_amd import hip_conv
    from oadg_amd.backbones import ResNet
    torch.manual_seed(0)
    net = ResNet(50, frozen_stages=1, norm_eval=True).to(dev).to(memory_format=torch.channels_last).train()
    with torch.no_grad():
        net.bn1.running_mean.normal_()
        net.bn1.running_var.uniform_(0.5, 1.5)
        net.bn1.bias.normal_()
    x = torch.randn(2, 3, 96, 160, device=dev).contiguous(memory_format=torch.channels_last)
    was = hip_conv.ENABLED
    try:
        with torch.autocast('cuda', dtype=torch.bfloat16):
            hip_conv.enable(False)
            ref = net._stem(x)
            hip_conv.enable(True)
            out = net._stem(x)
    finally:
        hip_conv.enable(was)
    # the library convolution and the csrc stem kernel accumulate in different orders: one bf16 rounding apart
    assert out.dtype == ref.dtype == torch.bfloat16 and out.shape == ref.shape
    d = (out.float() - ref.float()).abs()
    assert d.max().item() <= 2e-2 * ref.float().abs().max().item() and d.mean().item() <= 4e-3 * ref.float().abs().mean().item()
    from oadg_amd import backbones
    try:
        backbones.STEM_KERNEL = False
        with torch.autocast('cuda', dtype=torch.bfloat16):
            hip_conv.enable(True)
            assert torch.equal(net._stem(x), ref)          # library conv + fused tail == module chain, bit for bit
    finally:
        backbones.STEM_KERNEL = True
        hip_conv.enable(was)


@pytest.mark.parametrize('N,C,H,W,K,R', [(2, 128, 24, 40, 128, 3), (1, 64, 23, 37, 192, 3), (2, 256, 16, 18, 64, 1),
                                         (1, 128, 9, 7, 128, 1), (1, 64, 2, 2, 64, 3), (2, 128, 31, 64, 256, 3)])
def test_stride2_data_gradient_parity_classes(dev, N, C, H, W, K, R):
    """conv_dgrad_s2 (four parity-class convolutions + strided store) against autograd's data gradient of the stride-2
    convolution in fp32, with and without the fused ReLU mask / column sums; even and odd extents."""
    import torch.nn.functional as F
    from oadg_amd import hip_conv
    pad = 1 if R == 3 else 0
    g = torch.Generator(device=dev).manual_seed(11)
    w = (torch.randn(K, C, R, R, device=dev, generator=g) / (C * R * R) ** 0.5).requires_grad_(True)
    x = torch.randn(N, C, H, W, device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    wf, _, wt = hip_conv.prepared(w, None, None, 2)
    y = F.conv2d(x.float(), wf.detach().float(), None, 2, pad)
    gy = torch.randn(y.shape, device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    xr = x.float().requires_grad_(True)
    F.conv2d(xr, wf.detach().float(), None, 2, pad).backward(gy.float())
    ref = xr.grad
    got = hip_conv.conv_dgrad_s2(gy, wt, x.shape, R)
    assert got.shape == ref.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert (got.float() - ref).abs().max().item() <= 1e-2 * ref.abs().max().item() + 1e-6
    got_m, cs = hip_conv.conv_dgrad_s2(gy, wt, x.shape, R, mask=x, want_colsum=True)
    refm = ref * (x.float() > 0)
    assert (got_m.float() - refm).abs().max().item() <= 1e-2 * ref.abs().max().item() + 1e-6
    assert torch.equal(got_m == 0, (got_m == 0) | (x <= 0))
    assert torch.allclose(cs, got_m.float().sum((0, 2, 3)), rtol=1e-3, atol=1e-2)
    flags = (x.permute(0, 2, 3, 1).reshape(-1, 8) > 0).to(torch.uint8)          # the bit form of the same mask
    bits = (flags << torch.arange(8, device=dev, dtype=torch.uint8)).sum(1).to(torch.uint8)
    got_b, cs_b = hip_conv.conv_dgrad_s2(gy, wt, x.shape, R, want_colsum=True, mask_bits=bits)
    assert torch.equal(got_b, got_m) and torch.equal(cs_b, cs)


def test_downsample_stage_backward_on_own_stride2_kernels(dev, monkeypatch):
    """A stage with a stride-2 first block (Bottleneck.conv2 3x3/s2 + downsample 1x1/s2): gradients with the csrc
    stride-2 data gradients (and the ReLU mask / bias gradient of conv1 fused into conv2's) against the library ones."""
    from oadg_amd import hip_conv
    from oadg_amd.backbones import make_res_layer
    hip_conv.enable(True)
    try:
        torch.manual_seed(0)
        layer = make_res_layer(256, 128, 2, 2, 1, 'pytorch', dict(type='BN', requires_grad=True)).to(dev)
        layer.eval()
        for m in layer.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                torch.nn.init.uniform_(m.weight, 0.5, 1.5)
                torch.nn.init.normal_(m.bias, 0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.running_mean.normal_(0, 0.2)
        g = torch.Generator(device=dev).manual_seed(1)
        x0 = torch.randn(2, 256, 30, 44, device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
        gy = torch.randn(2, 512, 15, 22, device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
        res = {}
        for mode in (True, False):
            monkeypatch.setattr(hip_conv, 'S2_DGRAD', mode)
            layer.zero_grad(set_to_none=True)
            x = torch.relu(x0).clone().requires_grad_(True)
            with torch.autocast('cuda', dtype=torch.bfloat16):
                y = layer(x)
            y.backward(gy)
            res[mode] = (y.detach().float(), x.grad.float(), {n: p.grad.float().clone() for n, p in layer.named_parameters()})
        assert torch.equal(res[True][0], res[False][0])

        def close(a, b, what):
            d = (a - b).abs()
            assert d.max().item() <= 3e-2 * b.abs().max().item() + 1e-6, (what, d.max().item(), b.abs().max().item())
            assert d.mean().item() <= 5e-3 * b.abs().mean().item() + 1e-7, (what, d.mean().item(), b.abs().mean().item())
        close(res[True][1], res[False][1], 'x.grad')
        for n in res[False][2]:
            close(res[True][2][n], res[False][2][n], n)
    finally:
        hip_conv.enable(False)


@pytest.mark.parametrize('N,H,W', [(2, 64, 96), (1, 37, 54), (1, 5, 6), (2, 128, 256), (1, 9, 130), (1, 33, 47)])
def test_stem_conv_kernel_matches_fp32_reference(dev, N, H, W):
    """csrc/stem_conv.hip (7x7 / stride 2 / pad 3, 3 -> 64 on the matrix cores) against fp32 conv2d on the same bf16
    operands: one bf16 rounding of the result."""
    import torch.nn.functional as F
    from oadg_amd import hip_ops
    g = torch.Generator(device=dev).manual_seed(4)
    x = torch.randn(N, 3, H, W, device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    w = torch.randn(64, 3, 7, 7, device=dev, generator=g) / 12
    wp = hip_ops.stem_weights(w)
    y = hip_ops.stem_conv(x, wp)
    ref = F.conv2d(x.float(), w.bfloat16().float(), None, 2, 3)
    assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
    d = (y.float() - ref).abs()
    assert d.max().item() <= 8e-3 * ref.abs().max().item() + 1e-6, (d.max().item(), ref.abs().max().item())


def test_weight_bank_refresh_equals_per_layer_preparation(dev):
    """hip_conv.refresh_prepared() (ONE multi-layer launch after optimizer.step()) must leave exactly the tensors the
    per-layer preparation of the next forward pass would have produced - BN-folded bf16 weights, bias, scale and both
    data-gradient layouts (flipped / transposed for stride 1, parity classes for stride 2) - for plain and channels_last
    parameters, with and without BatchNorm; afterwards a forward pass launches no preparation (entries valid), and any
    other in-place change of a parameter invalidates its entry."""
    import torch.nn as nn
    from oadg_amd import hip_conv, layers
    torch.manual_seed(3)
    hip_conv.enable()
    try:
        specs = [(64, 128, 3, 1, 1, True, True), (128, 64, 1, 1, 0, True, False), (64, 64, 3, 2, 1, True, True),
                 (128, 256, 1, 2, 0, True, False), (64, 128, 3, 1, 1, False, True)]
        mods, params = [], []
        for C, K, R, st, pd, has_bn, cl in specs:
            conv = nn.Conv2d(C, K, R, st, pd, bias=not has_bn).to(dev)
            if cl:
                conv = conv.to(memory_format=torch.channels_last)
            bn = None
            if has_bn:
                bn = nn.BatchNorm2d(K).to(dev).eval()
                with torch.no_grad():
                    bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2)
            mods.append((conv, bn))
            params += list(conv.parameters()) + (list(bn.parameters()) if bn is not None else [])
        opt = torch.optim.SGD(params, lr=0.1, momentum=0.9)

        def forward():
            outs = []
            with torch.autocast('cuda', dtype=torch.bfloat16):
                for conv, bn in mods:
                    x = torch.randn(2, conv.in_channels, 16, 20, device=dev).bfloat16() \
                        .contiguous(memory_format=torch.channels_last).requires_grad_(True)
                    y = layers.conv_bn(x, conv, bn) if bn is not None else layers.conv2d(
                        x, conv.weight, conv.bias, conv.stride, conv.padding, conv.dilation, owner=conv)
                    outs.append(y)
            return outs

        def snapshot():
            out = []
            for conv, _ in mods:
                e = conv._oadg_prep_entry
                assert e.args is not None and e.valid
                out.append([t.clone() if t is not None else None for t in (e.wf, e.wt, e.bias, e.scale)])
            return out
        torch.manual_seed(1)
        sum(o.float().sum() for o in forward()).backward()         # fills the entries, produces gradients
        opt.step()
        assert not any(conv._oadg_prep_entry.valid for conv, _ in mods)        # versions moved
        n = hip_conv.refresh_prepared()
        assert n >= len(mods)            # (the bank is per process: layers of other live models are refreshed too)
        banked = snapshot()
        # reference: the per-layer preparation from the same (updated) parameters
        for (conv, bn), got in zip(mods, banked):
            e = conv._oadg_prep_entry
            if bn is not None:
                ref = hip_conv._PrepWeights.apply(conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps,
                                                  None, e.want_wt, None, None)
            else:
                ref = hip_conv._PrepWeights.apply(conv.weight, None, None, None, None, 0.0, conv.bias, e.want_wt, None, None)
            assert torch.equal(ref[0], got[0])
            assert torch.equal(ref[1], got[2])
            if e.want_wt:
                assert torch.equal(ref[2], got[1])
        # a forward pass on valid entries hands out the banked tensors (same storage)
        torch.manual_seed(1)
        forward()
        for conv, _ in mods:
            assert conv._oadg_prep_entry.valid
        with torch.no_grad():
            mods[0][0].weight.mul_(0.5)
        assert not mods[0][0]._oadg_prep_entry.valid and mods[1][0]._oadg_prep_entry.valid
        torch.manual_seed(1)
        y = forward()[0]
        assert mods[0][0]._oadg_prep_entry.valid            # re-prepared by its forward call
        assert torch.isfinite(y.float()).all()
    finally:
        hip_conv.enable(False)


def test_grad_token_late_depositor_taints_the_token(dev):
    """The out-of-order case of the multi-consumer GradToken: t = relu(P(x)) feeds a FINISHER convolution and
    a DEPOSITOR convolution; the depositor is issued first in the forward pass, so autograd runs its backward AFTER the
    finisher's (higher sequence numbers first).  The depositor then finds the token closed, must return its gradient AND
    taint the token (grad_ptr = None): the producer masks / reduces the SUM itself instead of trusting the finisher's
    pointer - autograd may have added the late gradient in place into that very tensor.  Checked against the same
    network without any token."""
    import torch.nn as nn
    from oadg_amd import hip_conv, layers
    hip_conv.enable(True)
    try:
        torch.manual_seed(2)
        P = nn.Conv2d(64, 128, 3, padding=1, bias=False).to(dev).to(memory_format=torch.channels_last)
        Pbn = nn.BatchNorm2d(128).to(dev).eval()
        F_ = nn.Conv2d(128, 64, 3, padding=1, bias=False).to(dev).to(memory_format=torch.channels_last)
        Fbn = nn.BatchNorm2d(64).to(dev).eval()
        D = nn.Conv2d(128, 128, 1, bias=False).to(dev)
        Dbn = nn.BatchNorm2d(128).to(dev).eval()
        with torch.no_grad():
            for bn in (Pbn, Fbn, Dbn):
                bn.weight.uniform_(0.5, 1.5); bn.bias.normal_(0, 0.3); bn.running_var.uniform_(0.5, 1.5); bn.running_mean.normal_(0, 0.3)
        params = [p for m in (P, Pbn, F_, Fbn, D, Dbn) for p in m.parameters()]
        x = torch.randn(2, 64, 24, 40, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
        g1 = torch.randn(2, 64, 24, 40, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
        g2 = torch.randn(2, 128, 24, 40, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
        res = {}
        for mode in ('tokens', 'plain'):
            for p in params:
                p.grad = None
            xin = x.clone().requires_grad_(True)
            tok = hip_conv.GradToken() if mode == 'tokens' else None
            with torch.autocast('cuda', dtype=torch.bfloat16):
                t = layers.conv_bn(xin, P, Pbn, relu=True, **(dict(out_token=tok) if tok else {}))
                yd = layers.conv_bn(t, D, Dbn, **(dict(dep_token=tok) if tok else {}))        # depositor FIRST
                yf = layers.conv_bn(t, F_, Fbn, **(dict(in_token=tok) if tok else {}))         # finisher second
            torch.autograd.backward([yf, yd], [g1, g2])
            if tok is not None:
                assert tok.closed and tok.grad_ptr is None and tok.extra is None     # tainted by the late depositor
            res[mode] = {i: p.grad.float().clone() for i, p in enumerate(params)}
            res[mode]['x'] = xin.grad.float().clone()
        for k in res['plain']:
            a, b = res['tokens'][k], res['plain'][k]
            assert (a - b).abs().max().item() <= 3e-2 * b.abs().max().item() + 1e-6, k
    finally:
        hip_conv.enable(False)


def test_deferred_column_sums_give_the_same_gradients_bit_for_bit(dev):
    """hip_conv.DEFER_COLSUM (TrainEngine's backward): the per-layer reductions of the bias-gradient partial sums run as
    ONE launch after the backward pass (oadg_colsum_reduce_multi), BN scale gradients are left raw by the weight
    preparation's backward and finished there.  Every parameter gradient of ResNet-50 + FPN + RPN head (shared across
    the five levels: its vectors must NOT wait, autograd sums them on arrival) equals the immediate form bit for bit;
    the deferred pass really defers (one multi launch with many jobs)."""
    from oadg_amd import hip_conv
    from oadg_amd.backbones import ResNet
    from oadg_amd.dense_heads import RPNHead
    from oadg_amd.necks import FPN
    hip_conv.enable(True)
    try:
        torch.manual_seed(0)
        bb = ResNet(depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                    norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, style='pytorch')
        neck = FPN(in_channels=[256, 512, 1024, 2048], out_channels=256, num_outs=5)
        neck.init_weights()
        head = RPNHead(in_channels=256, feat_channels=256,
                       anchor_generator=dict(type='AnchorGenerator', scales=[8], ratios=[0.5, 1.0, 2.0],
                                             strides=[4, 8, 16, 32, 64]),
                       loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                       loss_bbox=dict(type='L1Loss', loss_weight=1.0))
        net = torch.nn.ModuleList([bb, neck, head]).to(dev).to(memory_format=torch.channels_last)
        net.train()
        for m in bb.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                torch.nn.init.uniform_(m.weight, 0.5, 1.5)
                torch.nn.init.normal_(m.bias, 0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.running_mean.normal_(0, 0.2)
        g = torch.Generator(device=dev).manual_seed(1)
        x = torch.randn(2, 3, 128, 192, device=dev, generator=g).contiguous(memory_format=torch.channels_last)
        res, jobs = {}, {}
        for mode in ('immediate', 'deferred', 'deferred_again'):
            net.zero_grad(set_to_none=True)
            hip_conv.begin_step(defer=mode != 'immediate')
            assert hip_conv.DEFER_COLSUM == (mode != 'immediate')
            hip_conv.DEFER_WGRAD = False       # (grouped weight gradients re-split their sums: tested below, not bit-equal)
            try:
                with torch.autocast('cuda', dtype=torch.bfloat16):
                    cls, reg = head(neck(bb(x)))
                loss = sum((c.float() ** 2).mean() for c in cls) + sum(r.float().abs().mean() for r in reg)
                loss.backward()
                jobs[mode] = len(hip_conv._PENDING)
            finally:
                flushed = hip_conv.end_backward()
            assert flushed == jobs[mode] and not hip_conv._PENDING and not hip_conv.DEFER_COLSUM
            torch.cuda.synchronize()
            res[mode] = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
        assert jobs['immediate'] == 0 and jobs['deferred'] >= 30, jobs
        assert set(res['immediate']) == set(res['deferred']) and len(res['immediate']) > 100
        for n, a in res['immediate'].items():
            assert torch.equal(a, res['deferred'][n]), n
            assert torch.equal(a, res['deferred_again'][n]), n
        assert any('bn' in n and n.endswith('weight') and float(v.abs().sum()) > 0 for n, v in res['deferred'].items())
    finally:
        hip_conv.enable(False)
        hip_conv.DEFER_COLSUM = False


_WG_SHAPES = [(2, 256, 16, 24, 256, 3, 1, 1, 1), (2, 1024, 16, 24, 256, 1, 1, 0, 1), (2, 256, 16, 24, 1024, 1, 1, 0, 1),
              (2, 512, 8, 12, 512, 3, 1, 1, 1), (2, 512, 16, 24, 1024, 1, 2, 0, 1), (2, 256, 32, 48, 256, 3, 2, 1, 1),
              (1, 512, 9, 13, 256, 3, 1, 2, 2), (3, 256, 5, 7, 256, 3, 1, 1, 1),
              # row-strip order (output width a multiple of 64)
              (2, 256, 16, 64, 256, 3, 1, 1, 1), (3, 256, 11, 128, 256, 3, 1, 1, 1), (1, 256, 64, 128, 256, 3, 2, 1, 1),
              (2, 512, 12, 64, 512, 3, 1, 2, 2)]


@pytest.mark.parametrize('target', [256, 64, 7])
def test_grouped_weight_gradient_launch_matches_single_layer_launches(dev, target):
    """oadg_conv2d_wgrad_multi: the weight gradients of several layers (3x3 / 1x1, stride 2, dilation, maps smaller than
    one 64-pixel K-tile chunk row, a ragged last chunk) from ONE launch, every job with its own split count from
    oadg_conv2d_wgrad_multi_plan.  The summed partials equal the single-layer kernel's result to fp32 rounding and the
    fp32 reference (same bf16 operands) to 2e-3; ``target`` = the number of workgroups of ONE round the plan aims at (7: fewer
    than the group's weight tiles - one workgroup per tile)."""
    from oadg_amd import hip_conv
    g = torch.Generator(device=dev).manual_seed(5)
    jobs, refs, singles = [], [], []
    for N, C, H, W, K, R, stride, pad, dil in _WG_SHAPES:
        x = torch.randn(N, C, H, W, device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        Ho = (H + 2 * pad - dil * (R - 1) - 1) // stride + 1
        Wo = (W + 2 * pad - dil * (R - 1) - 1) // stride + 1
        gy = torch.randn(N, K, Ho, Wo, device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        jobs.append((x, gy, K, R, R, stride, pad, dil))
        refs.append(torch.nn.grad.conv2d_weight(x.float(), (K, C, R, R), gy.float(), stride=stride, padding=pad, dilation=dil))
        singles.append(hip_conv.conv_wgrad(x, gy, K, R, R, stride, pad, dil).float())
    ws, parts = hip_conv.wgrad_multi(jobs, target_blocks=target)
    torch.cuda.synchronize()
    base = ws.data_ptr()
    total = sum((K // 256) * (C // 256) * R * R * sp for (N, C, H, W, K, R, *_), (_, sp) in zip(_WG_SHAPES, parts))
    tiles = sum((K // 256) * (C // 256) * R * R for (N, C, H, W, K, R, *_) in _WG_SHAPES)
    assert total <= max(3 * target, tiles)       # (round 5: the plan may take a list of up to three rounds when its simulated launch is shorter)
    if target >= 256:
        assert max(sp for _, sp in parts) > 1
    for (N, C, H, W, K, R, *_), (pp, sp), ref, single in zip(_WG_SHAPES, parts, refs, singles):
        n = sp * K * R * R * C
        part = ws[pp - base: pp - base + 4 * n].view(torch.float32).view(sp, K, R, R, C)
        dw = part.sum(0).permute(0, 3, 1, 2)
        scale = ref.abs().max().item()
        assert (dw - ref).abs().max().item() <= 2e-3 * scale, (C, K, R)
        assert (dw - single).abs().max().item() <= 2e-5 * scale, (C, K, R)


def test_deferred_grouped_weight_gradients_in_a_backward_pass(dev, monkeypatch):
    """hip_conv.DEFER_WGRAD (TrainEngine's backward): the weight gradients of the layers with small maps are collected
    and launched in groups (csrc conv_wgrad256_multi_kernel + prep_weights_bwd_parts_multi_kernel); dW, dgamma and the
    deferred column sums / raw BN-scale dot products they interact with arrive in the parameters' .grad.  Against the
    immediate form: equal to fp32 rounding (other split counts, other summation order); run to run: bit-identical."""
    from oadg_amd import hip_conv
    from oadg_amd.backbones import ResNet
    from oadg_amd.necks import FPN
    hip_conv.enable(True)
    calls = []
    orig = hip_conv.wgrad_multi
    monkeypatch.setattr(hip_conv, 'wgrad_multi', lambda jobs, **k: (calls.append(len(jobs)), orig(jobs, **k))[1])
    monkeypatch.setattr(hip_conv, 'WGRAD_GROUP', 600)           # tiny maps here: a group every few layers
    try:
        torch.manual_seed(0)
        bb = ResNet(depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                    norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, style='pytorch')
        neck = FPN(in_channels=[256, 512, 1024, 2048], out_channels=256, num_outs=5)
        neck.init_weights()
        net = torch.nn.ModuleList([bb, neck]).to(dev).to(memory_format=torch.channels_last)
        net.train()
        for m in bb.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                torch.nn.init.uniform_(m.weight, 0.5, 1.5)
                torch.nn.init.normal_(m.bias, 0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.running_mean.normal_(0, 0.2)
        g = torch.Generator(device=dev).manual_seed(1)
        x = torch.randn(2, 3, 256, 384, device=dev, generator=g).contiguous(memory_format=torch.channels_last)
        res, groups = {}, {}
        for mode in ('immediate', 'deferred', 'deferred_again'):
            net.zero_grad(set_to_none=True)
            calls.clear()
            hip_conv.begin_step(defer=mode != 'immediate')
            assert hip_conv.DEFER_WGRAD == (mode != 'immediate')
            try:
                with torch.autocast('cuda', dtype=torch.bfloat16):
                    outs = neck(bb(x))
                loss = sum((o.float() ** 2).mean() for o in outs)
                loss.backward()
            finally:
                hip_conv.end_backward()
            assert not hip_conv._WQ and not hip_conv._PENDING and not hip_conv.DEFER_WGRAD
            torch.cuda.synchronize()
            groups[mode] = list(calls)
            res[mode] = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
        assert groups['immediate'] == [] and len(groups['deferred']) >= 3 and sum(groups['deferred']) >= 30, groups
        assert max(groups['deferred']) >= 4
        assert set(res['immediate']) == set(res['deferred']) and len(res['immediate']) > 100
        worst = 0.0
        for n, a in res['immediate'].items():
            b = res['deferred'][n]
            assert torch.isfinite(b).all(), n
            err = (a - b).abs().max().item() / (a.abs().max().item() + 1e-12)
            worst = max(worst, err)
            # (filters with C * R * S > 3000 - layer4's 3x3: the immediate form hands the weight gradient over in bf16,
            #  oadg_prep_conv_weights_bwd; the grouped form keeps fp32 partials)
            wide = 'layer4' in n and ('conv2' in n or 'bn2' in n)          # 512 x 3 x 3 filters and their BN scale
            assert err <= (5e-3 if wide else 1e-4), (n, err)
            assert torch.equal(b, res['deferred_again'][n]), n
        assert worst > 0.0          # (the groups really took other split counts)
    finally:
        hip_conv.enable(False)
        hip_conv.DEFER_COLSUM = hip_conv.DEFER_WGRAD = False


@pytest.mark.parametrize('C,M', [(256, 2 * 40 * 56), (256, 4 * 13 * 7 + 3), (128, 64 * 9 + 1), (256, 8 * 256 * 512 // 16)])
def test_narrow_head_kernels_match_fp32(dev, C, M):
    """csrc/narrow_head.hip: the 16-output-channel 1x1 forward, data gradient (+ ReLU mask bits + column sums) and weight
    gradient (+ bias gradient) against fp32 matmuls on the same bf16 operands - ragged pixel counts included."""
    from oadg_amd import _lib
    from oadg_amd.hip_conv import ptr, stream_ptr, check, _zeros
    L = _lib.lib()
    g = torch.Generator(device=dev).manual_seed(C + M)
    x = torch.randn(M, C, device=dev, generator=g).to(torch.bfloat16)
    w = torch.zeros(16, C, device=dev)
    w[:15] = torch.randn(15, C, device=dev, generator=g) * 0.05
    w16 = w.to(torch.bfloat16)
    b = torch.zeros(16, device=dev)
    b[:15] = torch.randn(15, device=dev, generator=g)
    y = torch.full((M, 16), float('nan'), device=dev, dtype=torch.bfloat16)
    check(L.oadg_conv1x1_n16_fwd(ptr(x), ptr(w16), ptr(b), ptr(y), M, C, stream_ptr()), 'fwd')
    ref = x.float() @ w16.float().t() + b
    assert torch.equal(y, ref.to(torch.bfloat16)) or (y.float() - ref).abs().max().item() <= 2 ** -7 * ref.abs().max().item()
    assert (y[:, 15] == 0).all()

    dy = torch.randn(M, 16, device=dev, generator=g).to(torch.bfloat16)
    dy[:, 15] = 0
    wt = w16.t().contiguous()
    mask = torch.rand(M, C, device=dev, generator=g) < 0.6
    bits = (mask.view(M, C // 8, 8).to(torch.int32) << torch.arange(8, device=dev, dtype=torch.int32)).sum(-1).to(torch.uint8)
    rows = L.oadg_conv1x1_n16_dgrad_rows(M)
    for masked in (False, True):
        dx = torch.full((M, C), float('nan'), device=dev, dtype=torch.bfloat16)
        part = torch.full((rows, C), float('nan'), device=dev)
        check(L.oadg_conv1x1_n16_dgrad(ptr(dy), ptr(wt), ptr(dx), ptr(bits) if masked else None, ptr(part), M, C,
                                       stream_ptr()), 'dgrad')
        r = dy.float() @ w16.float()
        if masked:
            r = r * mask
        assert (dx.float() - r).abs().max().item() <= 2 ** -7 * r.abs().max().item() + 1e-6
        cs = part.sum(0)
        rc = r.to(torch.bfloat16).float().sum(0)
        assert (cs - rc).abs().max().item() <= 2e-3 * rc.abs().max().item() + 1e-3, (cs - rc).abs().max().item()
    # (no column sums asked for)
    dx2 = torch.empty_like(dx)
    check(L.oadg_conv1x1_n16_dgrad(ptr(dy), ptr(wt), ptr(dx2), ptr(bits), None, M, C, stream_ptr()), 'dgrad')
    assert torch.equal(dx2, dx)

    rows = L.oadg_conv1x1_n16_wgrad_rows(M)
    part = torch.full((rows, 16, C), float('nan'), device=dev)
    bpart = torch.full((rows, 16), float('nan'), device=dev)
    check(L.oadg_conv1x1_n16_wgrad(ptr(x), ptr(dy), ptr(part), ptr(bpart), ptr(_zeros(dev)), M, C, stream_ptr()), 'wgrad')
    dw = part.sum(0)
    rw = dy.float().t() @ x.float()
    assert (dw - rw).abs().max().item() <= 1e-3 * rw.abs().max().item() + 1e-4, (dw - rw).abs().max().item()
    rb = dy.float().sum(0)
    assert (bpart.sum(0) - rb).abs().max().item() <= 1e-3 * rb.abs().max().item() + 1e-4


def test_rpn_head_narrow_matches_padded_tile(dev, monkeypatch):
    """The RPN head on 16-channel maps (csrc/narrow_head.hip, all pyramid levels one autograd node) against the same head
    on the zero-padded 128-channel tile, level by level: outputs equal up to the summation order (same bf16 operands,
    fp32 accumulation), gradients of every parameter (summed over the levels) and of the inputs within bf16 rounding."""
    from oadg_amd import hip_conv
    from oadg_amd.dense_heads import RPNHead
    torch.manual_seed(0)
    head = RPNHead(in_channels=256, feat_channels=256,
                   anchor_generator=dict(type='AnchorGenerator', scales=[8], ratios=[0.5, 1.0, 2.0], strides=[4, 8, 16]),
                   loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                   loss_bbox=dict(type='L1Loss', loss_weight=1.0)).to(dev)
    for m in (head.rpn_conv, head.rpn_cls, head.rpn_reg):
        torch.nn.init.normal_(m.weight, 0, 0.05)
        torch.nn.init.normal_(m.bias, 0, 0.1)
    g = torch.Generator(device=dev).manual_seed(1)
    shapes = [(40, 56), (20, 28), (10, 14)]
    x0 = [torch.randn(2, 256, h, w, device=dev, generator=g).contiguous(memory_format=torch.channels_last) for h, w in shapes]
    gc = [torch.randn(2, 3, h, w, device=dev, generator=g) for h, w in shapes]
    gr = [torch.randn(2, 12, h, w, device=dev, generator=g) for h, w in shapes]
    res = {}
    hip_conv.enable(True)
    try:
        for mode in (True, False):
            monkeypatch.setattr(hip_conv, 'NARROW_HEAD', mode)
            head.zero_grad(set_to_none=True)
            xs = [x.clone().requires_grad_(True) for x in x0]
            with torch.autocast('cuda', dtype=torch.bfloat16):
                cls, reg = head(tuple(xs))
            assert all(c._oadg_y.shape[1] == (16 if mode else 128) for c in cls)
            sum((c.float() * a).sum() + (r.float() * b).sum() for c, r, a, b in zip(cls, reg, gc, gr)).backward()
            res[mode] = ([c.detach().float() for c in cls] + [r.detach().float() for r in reg], [x.grad.float() for x in xs],
                         {n: p.grad.float().clone() for n, p in head.named_parameters()})
    finally:
        hip_conv.enable(False)
    for a, b in zip(res[True][0], res[False][0]):
        assert (a - b).abs().max().item() <= 2 ** -7 * b.abs().max().item()
    def close(a, b, what, tol):
        d = (a - b).abs()
        assert d.max().item() <= tol * b.abs().max().item() + 1e-6, (what, d.max().item(), b.abs().max().item())
    for l, (a, b) in enumerate(zip(res[True][1], res[False][1])):
        close(a, b, f'x{l}.grad', 1e-2)
    for n in res[False][2]:
        close(res[True][2][n], res[False][2][n], n, 1e-2)


@pytest.mark.parametrize('first', [False, True])
@pytest.mark.parametrize('shape', [(2, 48, 80), (1, 16, 16), (3, 32, 16), (2, 23, 40), (1, 184, 320), (1, 5, 7)])
def test_frozen_bottleneck_one_launch_matches_the_three_convolutions(dev, monkeypatch, shape, first):
    """csrc/bottleneck_frozen.hip (a frozen identity block of ResNet stage 1 as ONE launch: conv1 on the tile's halo,
    conv2 and conv3 out of LDS, x read once) against the block's three convolution launches and against fp32 arithmetic on
    the same folded operands: tile borders, image borders (zero padding of conv2's INPUT, not of x), several images."""
    from oadg_amd import hip_conv
    from oadg_amd.backbones import Bottleneck
    torch.manual_seed(3)
    N, H, W = shape
    if first:       # the stage's first block: 64 input channels, 1x1 downsample convolution on the shortcut
        from oadg_amd.backbones import make_res_layer
        blk = make_res_layer(64, 64, 1, 1, 1, 'pytorch', dict(type='BN'))[0].to(dev).eval()
    else:
        blk = Bottleneck(256, 64).to(dev).eval()
    for bn in (blk.bn1, blk.bn2, blk.bn3) + ((blk.downsample[1],) if first else ()):
        bn.weight.data.uniform_(0.5, 1.5)
        bn.bias.data.normal_(0, 0.2)
        bn.running_mean.normal_(0, 0.2)
        bn.running_var.uniform_(0.5, 1.5)
    for p in blk.parameters():
        p.requires_grad_(False)
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(N, 64 if first else 256, H, W, device=dev, generator=g).to(torch.bfloat16).contiguous(
        memory_format=torch.channels_last)
    out = {}
    hip_conv.enable(True)
    try:
        for fused in (True, False):
            monkeypatch.setattr(hip_conv, 'FUSED_FROZEN_BLOCK', fused)
            with torch.autocast('cuda', dtype=torch.bfloat16):
                out[fused] = blk(x).float()
    finally:
        hip_conv.enable(False)
    assert hip_conv.frozen_bottleneck(x, blk) is None            # (disabled again: the caller's three launches)
    with torch.no_grad():
        ref = blk.float()(x.float())                             # fp32 arithmetic, unfolded BN
    scale = ref.abs().max().item()
    for fused in (True, False):
        assert (out[fused] - ref).abs().max().item() <= 3e-2 * scale, fused
        assert (out[fused] - ref).abs().mean().item() <= 4e-3 * ref.abs().mean().item() + 1e-4, fused
    d = (out[True] - out[False]).abs()
    assert d.max().item() <= 2e-2 * scale and (d > 0).float().mean().item() < 0.2, (d.max().item(), (d > 0).float().mean().item())


@pytest.mark.parametrize('base', [(128, 256), (96, 320)])
def test_fpn_topdown_add_in_the_lateral_epilogue_is_bit_identical(dev, monkeypatch, base):
    """necks.FPN with the top-down add `laterals[i] += upsample(laterals[i + 1])` (fpn.py:166-175) folded into the lateral
    convolution's epilogue (streaming pointwise kernel, ConvArgs.res_up: the residual row of pixel (n, y, x) is row
    (n, y / 2, x / 2) of the coarser map) against lateral launch + fpn_topdown launch: every output level and every
    gradient (inputs, weights, biases) bit for bit - the same roundings in the same order."""
    from oadg_amd import hip_conv
    from oadg_amd.necks import FPN
    torch.manual_seed(0)
    fpn = FPN(in_channels=[256, 512, 1024, 2048], out_channels=256, num_outs=5).to(dev).to(memory_format=torch.channels_last)
    g = torch.Generator(device=dev).manual_seed(2)
    sizes = [(base[0] >> i, base[1] >> i) for i in range(4)]        # (power-of-two maps: shifts; others: divisions)
    x0 = [torch.randn(8, c, h, w, device=dev, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
          for c, (h, w) in zip((256, 512, 1024, 2048), sizes)]
    go = None
    res = {}
    taken = {True: 0, False: 0}
    real = hip_conv.conv_forward

    def counting(*a, **k):
        taken[bool(k.get('res_up'))] += 1
        return real(*a, **k)
    monkeypatch.setattr(hip_conv, 'conv_forward', counting)
    hip_conv.enable(True)
    try:
        for fused in (True, False):
            monkeypatch.setattr(hip_conv, 'TOPDOWN_FUSED', fused)
            before = taken[True]
            fpn.zero_grad(set_to_none=True)
            xs = [x.clone().requires_grad_(True) for x in x0]
            with torch.autocast('cuda', dtype=torch.bfloat16):
                outs = fpn(xs)
            if go is None:
                go = [torch.randn(o.shape, device=dev, generator=g).to(o.dtype) for o in outs]
            torch.autograd.backward(list(outs), go)
            res[fused] = ([o.detach().clone() for o in outs], [x.grad.clone() for x in xs],
                          {n_: p.grad.clone() for n_, p in fpn.named_parameters()})
            assert taken[True] - before == (2 if fused else 0)        # the two finest laterals took the add with them
    finally:
        hip_conv.enable(False)
    assert hip_conv.topdown_ok(x0[0], 256, res[True][0][1]) is False            # (kernels disabled again)
    for a, b in zip(res[True][0], res[False][0]):
        assert torch.equal(a, b)
    for a, b in zip(res[True][1], res[False][1]):
        assert torch.equal(a, b)
    for n_ in res[False][2]:
        assert torch.equal(res[True][2][n_], res[False][2][n_]), n_
