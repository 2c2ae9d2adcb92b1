"""-m gpu: the fp32 convolution kernels (csrc/conv_f32.hip, fp32 MFMA) against torch's convolution evaluated in fp64 on the
CPU - forward, data gradient (stride 1 and 2, dilation, padding), weight and bias gradient, for the shapes of the path:
3x3 / 1x1 / 7x7 stem (3 input channels), stride 2, dilated, the 3- and 12-channel RPN heads, ragged pixel counts."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

CASES = [  # N, C, H, W, K, R, stride, pad, dil
    (2, 64, 20, 24, 64, 3, 1, 1, 1),
    (1, 32, 17, 19, 96, 1, 1, 0, 1),
    (2, 3, 37, 41, 64, 7, 2, 3, 1),        # stem
    (2, 64, 18, 22, 128, 3, 2, 1, 1),      # stride-2 3x3
    (1, 128, 16, 16, 256, 1, 2, 0, 1),     # downsample 1x1 / 2
    (1, 64, 15, 21, 64, 3, 1, 2, 2),       # dilated (DC5)
    (2, 256, 9, 13, 3, 1, 1, 0, 1),        # rpn_cls
    (2, 256, 9, 13, 12, 1, 1, 0, 1),       # rpn_reg
    (1, 36, 8, 8, 40, 3, 1, 1, 1),         # channel counts that are not multiples of 32
]


@pytest.mark.parametrize('case', CASES)
def test_fp32_convolution_forward_and_gradients(dev, case):
    from oadg_amd.hip_conv_f32 import conv2d_f32
    N, C, H, W, K, R, stride, pad, dil = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(K, C, R, R, generator=g) * 0.1
    b = torch.randn(K, generator=g)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = F.conv2d(xr, wr, br, stride, pad, dil)
    gy = torch.randn(yr.shape, generator=g)
    (yr * gy.double()).sum().backward()
    xd, wd, bd = (t.to(dev).requires_grad_(True) for t in (x, w, b))
    y = conv2d_f32(xd, wd, bd, stride, pad, dil)
    assert y is not None and y.shape == yr.shape
    (y * gy.to(dev)).sum().backward()

    def close(a, ref, name):
        err = (a.detach().cpu().double() - ref.detach()).abs().max().item()
        assert err <= 2e-5 * ref.detach().abs().max().item() + 1e-7, (name, err)
    close(y, yr, 'y')
    close(xd.grad, xr.grad, 'dx')
    close(wd.grad, wr.grad, 'dw')
    close(bd.grad, br.grad, 'db')


def test_fp32_model_forward_uses_no_library_convolution(dev, monkeypatch):
    """a ResNet-50 + FPN forward / backward in fp32 on the device: every convolution goes through csrc/conv_f32.hip
    (F.conv2d is never reached) and the result equals the library path's to fp32 rounding"""
    import oadg_amd  # noqa: F401
    from oadg_amd import hip_conv_f32
    from oadg_amd.backbones import ResNet
    from oadg_amd.necks import FPN
    torch.manual_seed(0)
    bb = ResNet(depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1, norm_eval=True).to(dev).train()
    neck = FPN([256, 512, 1024, 2048], 256, num_outs=5).to(dev)
    x = torch.randn(1, 3, 96, 128, device=dev)
    outs = {}
    for mode in (True, False):
        monkeypatch.setattr(hip_conv_f32, 'ENABLED', mode)
        calls = []
        orig = F.conv2d
        monkeypatch.setattr(F, 'conv2d', lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
        for p in list(bb.parameters()) + list(neck.parameters()):
            p.grad = None
        ys = neck(bb(x))
        sum(y.square().mean() for y in ys).backward()
        monkeypatch.setattr(F, 'conv2d', orig)
        outs[mode] = ([y.detach().clone() for y in ys], bb.layer2[0].conv1.weight.grad.clone(), neck.lateral_convs[0].conv.weight.grad.clone()
                      if hasattr(neck.lateral_convs[0], 'conv') else neck.lateral_convs[0].weight.grad.clone(), len(calls))
    assert outs[True][3] == 0 and outs[False][3] > 50
    for a, b in zip(outs[True][0], outs[False][0]):
        assert (a - b).abs().max().item() <= 1e-4 * b.abs().max().item()
    for i in (1, 2):
        assert (outs[True][i] - outs[False][i]).abs().max().item() <= 2e-4 * outs[False][i].abs().max().item()
