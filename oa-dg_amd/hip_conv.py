"""autograd wrapper of the MFMA implicit-GEMM convolution (csrc/conv_mfma.hip) and its registration as the
convolution implementation of :mod:`oadg_amd.layers`.

Forward and the stride-1 data gradient run on the hand-written kernel; the weight/bias gradients (a GEMM reduced
over the pixel dimension) still go through ``aten.convolution_backward`` (MIOpen) this round.
"""
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr

_ZEROS = {}
# The hand-written weight-gradient kernel is correct (tests/test_hip_conv.py) but at 0.8-1.0x of MIOpen's on the
# large-pixel-count layers (tools/bench_conv.py), so it is opt-in until it wins.
USE_HIP_WGRAD = False
# live HIP-event timing of the kernel launches inside bench.py's timed region: list of (start, end, flops)
TIMERS = None


def _zeros(device):
    z = _ZEROS.get(device)
    if z is None:
        z = _ZEROS[device] = torch.zeros(64, dtype=torch.uint8, device=device)
    return z


def supported(x, weight, stride, padding, dilation):
    K, C, R, S = weight.shape
    return (x.is_cuda and x.dim() == 4 and C % 64 == 0 and K % 128 == 0 and stride[0] == stride[1] and
            padding[0] == padding[1] and dilation[0] == dilation[1] and R == S)


def conv_forward(x, w, bias, residual, stride, pad, dil, relu):
    """x [N,C,H,W] bf16 channels_last, w [K,C,R,S] bf16 channels_last -> y [N,K,Ho,Wo] bf16 channels_last."""
    L = _lib.lib()
    N, C, H, W = x.shape
    K, _, R, S = w.shape
    Ho = (H + 2 * pad - dil * (R - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (S - 1) - 1) // stride + 1
    y = torch.empty((N, K, Ho, Wo), dtype=torch.bfloat16, device=x.device, memory_format=torch.channels_last)
    if TIMERS is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(L.oadg_conv2d_nhwc_bf16(ptr(x), ptr(w), ptr(bias), ptr(residual), ptr(y), ptr(_zeros(x.device)), N, H, W,
                                  C, K, R, S, stride, pad, dil, int(bool(relu)), stream_ptr()),
          'oadg_conv2d_nhwc_bf16')
    if TIMERS is not None:
        e1.record()
        TIMERS.append((e0, e1, 2.0 * N * Ho * Wo * K * C * R * S))
    return y


def conv_wgrad(x16, gy16, K, R, S, stride, pad, dil):
    """dW [K,C,R,S] (fp32, channels_last strides) of y = conv(x, W): csrc/conv_mfma.hip conv_wgrad_kernel."""
    L = _lib.lib()
    N, C, H, W = x16.shape
    Ho, Wo = gy16.shape[2], gy16.shape[3]
    nbytes = L.oadg_conv2d_wgrad_workspace_bytes(N, Ho, Wo, C, K, R, S)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x16.device)
    dw = torch.empty((K, R, S, C), dtype=torch.float32, device=x16.device)
    check(L.oadg_conv2d_wgrad_nhwc_bf16(ptr(x16), ptr(gy16), ptr(dw), ptr(_zeros(x16.device)), ptr(ws), nbytes, N,
                                        H, W, C, K, R, S, stride, pad, dil, stream_ptr()),
          'oadg_conv2d_wgrad_nhwc_bf16')
    return dw.permute(0, 3, 1, 2)


def _nhwc_bf16(t):
    return t.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)


class _Conv2dMFMA(torch.autograd.Function):
    """y = [relu]( conv(x, w) + b [+ residual] ) in one kernel; backward = ReLU mask (one element-wise pass),
    stride-1 data gradient on the same kernel, weight/bias gradients through aten (MIOpen) for now."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, stride, pad, dil, relu):
        x16, w16 = _nhwc_bf16(x), _nhwc_bf16(weight)
        b32 = bias.float().contiguous() if bias is not None else None
        r16 = _nhwc_bf16(residual) if residual is not None else None
        y = conv_forward(x16, w16, b32, r16, stride, pad, dil, relu)
        ctx.save_for_backward(x16, w16, y if relu else None)
        ctx.cfg = (stride, pad, dil, bias is not None, x.dtype, weight.dtype,
                   bias.dtype if bias is not None else None, residual.dtype if residual is not None else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x16, w16, y = ctx.saved_tensors
        stride, pad, dil, has_bias, xdt, wdt, bdt, rdt = ctx.cfg
        gy = _nhwc_bf16(gy)
        if y is not None:
            gy = torch.ops.aten.threshold_backward(gy, y, 0)
        K, C, R, S = w16.shape
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        gx = None
        pad_t = dil * (R - 1) - pad
        if need_x and stride == 1 and pad_t >= 0 and K % 64 == 0 and C % 128 == 0:
            # dx = conv(dy, W^T rotated by 180 degrees): the same kernel, channels swapped
            wt = _nhwc_bf16(w16.flip(2, 3).transpose(0, 1))
            gx = conv_forward(gy, wt, None, None, 1, pad_t, dil, False)
            need_x = False
        want_b = has_bias and ctx.needs_input_grad[2]
        gw = gb = None
        if USE_HIP_WGRAD and need_w and K % 128 == 0 and C % 128 == 0:
            gw = conv_wgrad(x16, gy, K, R, S, stride, pad, dil).to(wdt)
            need_w = False
            if want_b:
                gb = gy.float().sum((0, 2, 3)).to(bdt)
                want_b = False
        if need_x or need_w or want_b:
            outs = torch.ops.aten.convolution_backward(
                gy, x16, w16, [K] if has_bias else None, [stride, stride], [pad, pad], [dil, dil], False, [0, 0],
                1, [need_x, need_w, want_b])
            if gx is None:
                gx = outs[0]
            if outs[1] is not None:
                gw = outs[1].to(wdt)
            if outs[2] is not None:
                gb = outs[2].to(bdt)
        gres = gy.to(rdt) if (rdt is not None and ctx.needs_input_grad[3]) else None
        return (gx.to(xdt) if gx is not None else None), gw, gb, gres, None, None, None, None


def _norm3(stride, padding, dilation):
    t = lambda v: tuple(v) if isinstance(v, (tuple, list)) else (v, v)  # noqa: E731
    return t(stride), t(padding), t(dilation)


def conv2d(x, weight, bias, stride, padding, dilation, relu=False, residual=None):
    """layers.conv2d implementation hook: returns None for shapes the kernel does not cover."""
    stride, padding, dilation = _norm3(stride, padding, dilation)
    if not supported(x, weight, stride, padding, dilation):
        return None
    if not (x.dtype == torch.bfloat16 or torch.is_autocast_enabled()):
        return None      # fp32 parity runs keep fp32 arithmetic
    return _Conv2dMFMA.apply(x, weight, bias, residual, stride[0], padding[0], dilation[0], bool(relu))


def enable(on=True):
    from . import layers
    layers.set_conv_impl(conv2d if on else None)
