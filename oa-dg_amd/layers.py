"""Convolution / normalisation bricks with mmcv-compatible parameter names.

``conv2d`` is the single dispatch point of every convolution on the path: shapes covered by the hand-written
MFMA implicit-GEMM kernels go to the HIP library, everything else to torch's convolution (MIOpen), which is
library plumbing for plain convolutions exactly like the reference's cuDNN calls.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

_CONV_IMPL = None      # set by hip_conv.enable(): (x, w, b, stride, padding, dilation, relu, residual) -> y | None
_CONV_BN_IMPL = None   # set by hip_conv.enable(): (x, conv, bn, relu, residual) -> y | None


def set_conv_impl(fn, fn_bn=None):
    global _CONV_IMPL, _CONV_BN_IMPL
    _CONV_IMPL, _CONV_BN_IMPL = fn, fn_bn


def conv2d(x, weight, bias=None, stride=1, padding=0, dilation=1, relu=False, residual=None, **tokens):
    """[relu]( conv2d(x, weight, bias) [+ residual] ): fused in the MFMA kernel's epilogue when it applies.
    ``tokens``: hip_conv.GradToken hand-offs, only seen by the MFMA implementation."""
    if _CONV_IMPL is not None and x.is_cuda:
        y = _CONV_IMPL(x, weight, bias, stride, padding, dilation, relu, residual, **tokens)
        if y is not None:
            return y
    y = None
    if x.is_cuda and x.dtype == torch.float32:
        # the fp32 parity path: csrc/conv_f32.hip (fp32 MFMA) - no library convolution on either path (round 5)
        from .hip_conv_f32 import conv2d_f32
        y = conv2d_f32(x, weight, bias, stride, padding, dilation)
    if y is None:
        y = F.conv2d(x, weight, bias, stride, padding, dilation)
    if residual is not None:
        y = y + residual
    return F.relu(y, inplace=True) if relu else y


class Conv2d(nn.Conv2d):
    """nn.Conv2d (same state_dict keys) routed through :func:`conv2d`."""

    def forward(self, x):
        assert self.groups == 1 and self.padding_mode == 'zeros'
        return conv2d(x, self.weight, self.bias, self.stride, self.padding, self.dilation, owner=self)


# Frozen-statistics BatchNorm (norm_eval=True: every BN of the detector, resnet.py:648-657) is a per-channel
# affine map, so it is folded into the preceding convolution: y = conv(x, W * s) + t with
# s = gamma / sqrt(var + eps), t = beta - mean * s.  The activation-sized BN forward/backward passes (and the
# two extra reads of dy and x that autograd's batch_norm backward needs for d(gamma), d(beta)) disappear;
# d(gamma) and d(beta) come out of the weight-sized products and the convolution's bias gradient.
FOLD_EVAL_BN = True


def conv_bn(x, conv, bn, relu=False, residual=None, **tokens):
    """[relu]( bn(conv(x)) [+ residual] ) for a BatchNorm in eval mode, folded into the convolution; falls back
    to the module-by-module form otherwise.  ``tokens`` (in_token / out_token / res_token, hip_conv.GradToken) only
    reach the MFMA implementation; callers create them after hip_conv.tokens_ok()."""
    if not FOLD_EVAL_BN or bn.training or conv.bias is not None:
        y = bn(conv(x))
        if residual is not None:
            y = y + residual
        return F.relu(y, inplace=True) if relu else y
    if _CONV_BN_IMPL is not None and x.is_cuda:
        y = _CONV_BN_IMPL(x, conv, bn, relu, residual, **tokens)
        if y is not None:
            return y
    frozen = not (conv.weight.requires_grad or bn.weight.requires_grad or bn.bias.requires_grad)
    key = None
    if frozen:   # constant until the parameters/buffers are overwritten: fold once
        key = (conv.weight._version, bn.weight._version, bn.bias._version, bn.running_mean._version,
               bn.running_var._version, conv.weight.device, conv.weight.data_ptr())
        cached = getattr(conv, '_folded', None)
        if cached is not None and cached[0] == key:
            return conv2d(x, cached[1], cached[2], conv.stride, conv.padding, conv.dilation, relu, residual)
    scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
    w = conv.weight * scale.view(-1, 1, 1, 1)
    b = bn.bias - bn.running_mean * scale
    if frozen:
        w, b = w.detach(), b.detach()
        conv._folded = (key, w, b)
    return conv2d(x, w, b, conv.stride, conv.padding, conv.dilation, relu, residual)


def folded_frozen(conv, bn):
    """(w * s, beta - mean * s) of a frozen conv + eval-mode BN pair, cached on the conv until a tensor is overwritten."""
    key = (conv.weight._version, bn.weight._version, bn.bias._version, bn.running_mean._version,
           bn.running_var._version, conv.weight.device, conv.weight.data_ptr())
    cached = getattr(conv, '_folded', None)
    if cached is None or cached[0] != key:
        with torch.no_grad():
            scale = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
            cached = conv._folded = (key, conv.weight * scale.view(-1, 1, 1, 1), bn.bias - bn.running_mean * scale)
    return cached[1], cached[2]


def build_norm_layer(cfg, num_features, postfix=''):
    """mmcv.cnn.build_norm_layer for BN: returns (name, layer); ``requires_grad`` from the cfg."""
    cfg = dict(cfg)
    t = cfg.pop('type')
    if t not in ('BN', 'BN2d'):
        raise NotImplementedError(f'norm type {t} is not used by the named configs')
    requires_grad = cfg.pop('requires_grad', True)
    cfg.setdefault('eps', 1e-5)
    layer = nn.BatchNorm2d(num_features, **cfg)
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return 'bn' + str(postfix), layer


class ConvModule(nn.Module):
    """conv [+ BN] [+ ReLU] with mmcv.cnn.ConvModule's attribute names (``conv``, ``bn``, ``activate``)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, bias='auto',
                 conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'), inplace=True):
        super().__init__()
        assert conv_cfg is None or conv_cfg.get('type') in ('Conv2d', None)
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if bias == 'auto':
            bias = not self.with_norm
        self.conv = Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, bias=bias)
        if self.with_norm:
            _, self.bn = build_norm_layer(norm_cfg, out_channels)
        if self.with_activation:
            assert act_cfg['type'] == 'ReLU'
            self.activate = nn.ReLU(inplace=inplace)
        kaiming_init(self.conv)

    def forward(self, x):
        x = self.conv(x)
        if self.with_norm:
            x = self.bn(x)
        if self.with_activation:
            x = self.activate(x)
        return x


def kaiming_init(m, a=0, mode='fan_out', nonlinearity='relu', bias=0):
    nn.init.kaiming_normal_(m.weight, a=a, mode=mode, nonlinearity=nonlinearity)
    if getattr(m, 'bias', None) is not None:
        nn.init.constant_(m.bias, bias)


def xavier_init(m, gain=1, bias=0, distribution='uniform'):
    (nn.init.xavier_uniform_ if distribution == 'uniform' else nn.init.xavier_normal_)(m.weight, gain=gain)
    if getattr(m, 'bias', None) is not None:
        nn.init.constant_(m.bias, bias)


def normal_init(m, mean=0, std=1, bias=0):
    nn.init.normal_(m.weight, mean, std)
    if getattr(m, 'bias', None) is not None:
        nn.init.constant_(m.bias, bias)


def constant_init(m, val, bias=0):
    if getattr(m, 'weight', None) is not None:
        nn.init.constant_(m.weight, val)
    if getattr(m, 'bias', None) is not None:
        nn.init.constant_(m.bias, bias)
