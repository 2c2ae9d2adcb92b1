"""fp32 convolutions on the hand-written fp32-MFMA kernels (csrc/conv_f32.hip): the convolution implementation of
:mod:`oadg_amd.layers` for fp32 CUDA tensors outside autocast - the fp32 parity path (``bench.py --dtype fp32``, the fp32
whole-step tests), which rounds 1-4 ran on MIOpen through ``F.conv2d``.  Forward, data gradient (any stride: the kernel's
transposed gather mode) and weight gradient; exact fp32 products and sums in another order than a library's.

The benchmarked configuration (bf16 autocast) never comes here: :mod:`oadg_amd.hip_conv` takes it.
"""
import os

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr

ENABLED = os.environ.get('OADG_CONV_F32', '1') == '1'      # 0: fp32 convolutions through F.conv2d (MIOpen) again
_ZEROS = {}


def _zeros(device):
    z = _ZEROS.get(device)
    if z is None:
        z = _ZEROS[device] = torch.zeros(64, dtype=torch.uint8, device=device)
    return z


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def _nhwc(t):
    """fp32 NHWC memory (a channels_last NCHW tensor) with the channel count padded to a multiple of 4 by zeros (the
    kernels read 16-byte pieces along the channels: the 3-channel image of the stem, the 3 / 12 channels of rpn_cls / rpn_reg)"""
    C = t.shape[1]
    if C % 4:
        t = torch.nn.functional.pad(t, (0, 0, 0, 0, 0, 4 - C % 4))
    return t.contiguous(memory_format=torch.channels_last)


def _krsc(w):
    """[K][C][R][S] -> [K][R][S][C4] fp32 contiguous, C zero-padded to a multiple of 4"""
    C = w.shape[1]
    if C % 4:
        w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 4 - C % 4))
    return w.permute(0, 2, 3, 1).contiguous()


def _conv(x4, w_krsc, bias, stride, pad, dil, transposed=False, out_hw=(0, 0)):
    N, C, H, W = x4.shape
    K, R, S, _ = w_krsc.shape
    if transposed:
        Ho, Wo = out_hw
    else:
        Ho = (H + 2 * pad - dil * (R - 1) - 1) // stride + 1
        Wo = (W + 2 * pad - dil * (S - 1) - 1) // stride + 1
    y = torch.empty((N, K, Ho, Wo), dtype=torch.float32, device=x4.device, memory_format=torch.channels_last)
    check(_lib.lib().oadg_conv2d_f32(ptr(x4), ptr(w_krsc), ptr(bias), ptr(y), ptr(_zeros(x4.device)), N, H, W, C, K, R, S,
                                     stride, pad, dil, int(transposed), Ho, Wo, stream_ptr()), 'oadg_conv2d_f32')
    return y


class _Conv2dF32(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, dil):
        x4 = _nhwc(x.detach())
        wd = w.detach()
        y = _conv(x4, _krsc(wd), bias.detach().contiguous() if bias is not None else None, stride, pad, dil)
        ctx.save_for_backward(x4, wd)
        ctx.cfg = (stride, pad, dil, bias is not None, x.shape[1])
        return y

    @staticmethod
    def backward(ctx, gy):
        x4, w = ctx.saved_tensors
        stride, pad, dil, has_bias, C = ctx.cfg
        K, _, R, S = w.shape
        g4 = _nhwc(gy)                       # [N][Ho][Wo][K4]
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            # dx = the transposed gather over dy with the weights as [C][R][S][K4]
            wt = w.permute(1, 2, 3, 0)
            if K % 4:
                wt = torch.nn.functional.pad(wt, (0, 4 - K % 4))
            gx = _conv(g4, wt.contiguous(), None, stride, pad, dil, transposed=True, out_hw=(x4.shape[2], x4.shape[3]))
            if gx.shape[1] != C:
                gx = gx[:, :C]
        if ctx.needs_input_grad[1]:
            L = _lib.lib()
            N, C4, H, W = x4.shape
            K4, Ho, Wo = g4.shape[1], g4.shape[2], g4.shape[3]
            splits = L.oadg_conv2d_wgrad_f32_splits(N, Ho, Wo, C4, K4, R, S)
            ws = torch.empty(splits * K4 * R * S * C4, dtype=torch.float32, device=x4.device)
            dw = torch.empty((K4, R, S, C4), dtype=torch.float32, device=x4.device)
            check(L.oadg_conv2d_wgrad_f32(ptr(x4), ptr(g4), ptr(dw), ptr(ws), ws.numel() * 4, N, H, W, C4, K4, R, S, stride, pad,
                                          dil, stream_ptr()), 'oadg_conv2d_wgrad_f32')
            gw = dw[:K, :, :, :C].permute(0, 3, 1, 2)
        if has_bias and ctx.needs_input_grad[2]:
            gb = gy.sum((0, 2, 3))
        return gx, gw, gb, None, None, None


def conv2d_f32(x, weight, bias, stride, padding, dilation):
    """``F.conv2d`` for fp32 CUDA tensors on the csrc kernels, or None when the call is outside their domain (groups,
    asymmetric geometry, other dtypes, autocast) - the caller then uses the library"""
    if not ENABLED or not x.is_cuda or x.dtype != torch.float32 or weight.dtype != torch.float32 or x.dim() != 4 or \
            torch.is_autocast_enabled():
        return None
    st, pd, dl = _pair(stride), _pair(padding), _pair(dilation)
    if st[0] != st[1] or pd[0] != pd[1] or dl[0] != dl[1] or weight.shape[1] != x.shape[1] or isinstance(padding, str):
        return None
    if x.shape[0] * x.shape[2] * x.shape[3] >= 2 ** 31 or x.numel() >= 2 ** 31:
        return None
    return _Conv2dF32.apply(x, weight, bias, st[0], pd[0], dl[0])
