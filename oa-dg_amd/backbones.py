"""ResNet-50/101 backbone with mmdet's state_dict layout (mmdet/models/backbones/resnet.py:97-302,306-657,
mmdet/models/utils/res_layer.py).  Frozen stem + stage 1, BN always in eval mode (norm_eval)."""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.modules.batchnorm import _BatchNorm

import os

from . import hip_conv
from .layers import Conv2d, build_norm_layer, constant_init, conv_bn, kaiming_init
from .registry import BACKBONES


STEM_KERNEL = os.environ.get('OADG_STEM_KERNEL', '1') == '1'      # own 7x7/s2 stem convolution (else MIOpen)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None, style='pytorch',
                 norm_cfg=dict(type='BN')):
        super().__init__()
        assert style in ('pytorch', 'caffe')
        # resnet.py:154-159: the stride-2 layer is the 3x3 (pytorch) or the first 1x1 (caffe)
        s1, s2 = (1, stride) if style == 'pytorch' else (stride, 1)
        self.conv1 = Conv2d(inplanes, planes, 1, stride=s1, bias=False)
        _, self.bn1 = build_norm_layer(norm_cfg, planes)
        self.conv2 = Conv2d(planes, planes, 3, stride=s2, padding=dilation, dilation=dilation, bias=False)
        _, self.bn2 = build_norm_layer(norm_cfg, planes)
        self.conv3 = Conv2d(planes, planes * self.expansion, 1, bias=False)
        _, self.bn3 = build_norm_layer(norm_cfg, planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        # GradTokens (hip_conv): the backward of each fused bias+ReLU epilogue - mask, bias gradient, and for the
        # block input the identity-path gradient add - is folded into the data-gradient kernel of the tensor's
        # consumer.  t_in rides on x when the previous block produced it.  In a stage's first block x has several
        # consumers: conv1 finishes its gradient, the downsample convolution (run AFTER conv2 here, so that autograd
        # runs its backward BEFORE conv1's) and the FPN lateral deposit theirs on the token (hip_conv.GradToken).
        out = hip_conv.frozen_bottleneck(x, self)           # a frozen identity block of stage 1: one launch
        if out is not None:
            return out
        t_in = getattr(x, '_oadg_token', None)
        if hip_conv.tokens_ok(x, self.conv1, self.conv2, self.conv3) and not (self.bn1.training or self.bn2.training
                                                                              or self.bn3.training):
            t_a, t_b, t_out = hip_conv.GradToken(), hip_conv.GradToken(), hip_conv.GradToken()
            if not x.requires_grad or (self.downsample is not None and
                                       not hip_conv.tokens_ok(x, self.downsample[0])):
                t_in = None
            out = conv_bn(x, self.conv1, self.bn1, relu=True, in_token=t_in, out_token=t_a)
            out = conv_bn(out, self.conv2, self.bn2, relu=True, in_token=t_a, out_token=t_b)
            if self.downsample is not None:
                identity = conv_bn(x, self.downsample[0], self.downsample[1], dep_token=t_in)
                out = conv_bn(out, self.conv3, self.bn3, relu=True, residual=identity, in_token=t_b, out_token=t_out)
            else:
                out = conv_bn(out, self.conv3, self.bn3, relu=True, residual=x, in_token=t_b, out_token=t_out,
                              res_token=t_in)
            out._oadg_token = t_out
            return out
        identity = x
        if self.downsample is not None:
            identity = conv_bn(x, self.downsample[0], self.downsample[1])
        out = conv_bn(x, self.conv1, self.bn1, relu=True)
        out = conv_bn(out, self.conv2, self.bn2, relu=True)
        return conv_bn(out, self.conv3, self.bn3, relu=True, residual=identity)   # relu(bn3(conv3) + identity)


def make_res_layer(inplanes, planes, num_blocks, stride, dilation, style, norm_cfg):
    downsample = None
    if stride != 1 or inplanes != planes * Bottleneck.expansion:
        downsample = nn.Sequential(
            Conv2d(inplanes, planes * Bottleneck.expansion, 1, stride=stride, bias=False),
            build_norm_layer(norm_cfg, planes * Bottleneck.expansion)[1])
    blocks = [Bottleneck(inplanes, planes, stride, dilation, downsample, style, norm_cfg)]
    inplanes = planes * Bottleneck.expansion
    for _ in range(1, num_blocks):
        blocks.append(Bottleneck(inplanes, planes, 1, dilation, None, style, norm_cfg))
    return nn.Sequential(*blocks)


@BACKBONES.register_module()
class ResNet(nn.Module):
    arch_settings = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}

    def __init__(self, depth, in_channels=3, stem_channels=None, base_channels=64, num_stages=4,
                 strides=(1, 2, 2, 2), dilations=(1, 1, 1, 1), out_indices=(0, 1, 2, 3), style='pytorch',
                 deep_stem=False, avg_down=False, frozen_stages=-1, conv_cfg=None,
                 norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, dcn=None,
                 stage_with_dcn=(False, False, False, False), plugins=None, with_cp=False,
                 zero_init_residual=True, pretrained=None, init_cfg=None):
        super().__init__()
        if depth not in self.arch_settings:
            raise KeyError(f'invalid depth {depth} for resnet (bottleneck depths only)')
        assert not deep_stem and not avg_down and dcn is None and plugins is None and not with_cp
        assert 1 <= num_stages <= 4 and len(strides) == len(dilations) == num_stages
        assert max(out_indices) < num_stages
        self.depth, self.out_indices, self.frozen_stages = depth, out_indices, frozen_stages
        self.norm_eval, self.zero_init_residual, self.init_cfg = norm_eval, zero_init_residual, init_cfg
        stem_channels = stem_channels or base_channels
        self.conv1 = Conv2d(in_channels, stem_channels, 7, stride=2, padding=3, bias=False)
        _, self.bn1 = build_norm_layer(norm_cfg, stem_channels)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.res_layers = []
        inplanes = stem_channels
        for i, nb in enumerate(self.arch_settings[depth][:num_stages]):
            planes = base_channels * 2 ** i
            layer = make_res_layer(inplanes, planes, nb, strides[i], dilations[i], style, norm_cfg)
            inplanes = planes * Bottleneck.expansion
            name = f'layer{i + 1}'
            self.add_module(name, layer)
            self.res_layers.append(name)
        self.feat_dim = inplanes
        self._freeze_stages()

    def init_weights(self, allow_missing_pretrained=None):
        """resnet.py:405-424 defaults: Kaiming(fan_out, relu) convs, BN gamma 1, last BN of a block 0 - then
        ``init_cfg=dict(type='Pretrained', checkpoint=...)`` (BaseModule.init_weights -> PretrainedInit): the checkpoint
        is loaded into the backbone (a detector checkpoint contributes its ``backbone.`` keys).  A configured
        checkpoint that cannot be resolved raises, unless ``allow_missing_pretrained`` (or OADG_ALLOW_RANDOM_INIT=1:
        synthetic benchmarks and tests, which are defined on random-init weights) downgrades it to a loud warning."""
        self._random_init()
        cfg = self.init_cfg
        if isinstance(cfg, (list, tuple)):
            cfg = next((c for c in cfg if c.get('type') == 'Pretrained'), None)
        if cfg is None or cfg.get('type') != 'Pretrained' or not cfg.get('checkpoint'):
            return
        from .checkpoint import load_checkpoint, resolve_checkpoint
        try:
            path = resolve_checkpoint(cfg['checkpoint'])
        except FileNotFoundError as e:
            if allow_missing_pretrained is None:
                allow_missing_pretrained = os.environ.get('OADG_ALLOW_RANDOM_INIT', '0') == '1'
            if not allow_missing_pretrained:
                raise
            import warnings
            warnings.warn(f'[oadg] backbone init_cfg: {e} -> RANDOM initialisation (frozen stem / stage 1 included)')
            return
        import torch as _t
        keys = _t.load(path, map_location='cpu', weights_only=False)
        keys = keys.get('state_dict', keys)
        prefix = cfg.get('prefix') or ('backbone' if any(k.startswith('backbone.') for k in keys) else None)
        load_checkpoint(self, path, prefix=prefix)

    def _random_init(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                kaiming_init(m)
            elif isinstance(m, _BatchNorm):
                constant_init(m, 1)
        if self.zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    constant_init(m.bn3, 0)

    def _freeze_stages(self):
        if self.frozen_stages >= 0:
            self.bn1.eval()
            for m in (self.conv1, self.bn1):
                for p in m.parameters():
                    p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            m = getattr(self, f'layer{i}')
            m.eval()
            for p in m.parameters():
                p.requires_grad = False

    def _stem(self, x):
        """maxpool(relu(bn1(conv1(x)))) (resnet.py:631-637).  Frozen stem on the MI355X in bf16: the library convolution
        without bias, then bias + ReLU + max-pool in one pass (csrc/eltwise.hip) - same bits as the unfused chain."""
        from . import hip_conv, hip_ops, layers
        c, bn, mp = self.conv1, self.bn1, self.maxpool
        frozen = not (c.weight.requires_grad or bn.weight.requires_grad or bn.bias.requires_grad or x.requires_grad)
        if hip_conv.ENABLED and x.is_cuda and frozen and not bn.training and layers.FOLD_EVAL_BN and c.bias is None and \
                (x.dtype == torch.bfloat16 or torch.is_autocast_enabled()) and c.out_channels % 8 == 0 and \
                (mp.kernel_size, mp.stride, mp.padding, mp.dilation, mp.ceil_mode) == (3, 2, 1, 1, False):
            w, b = layers.folded_frozen(c, bn)
            x = x.to(torch.bfloat16)
            if STEM_KERNEL and tuple(w.shape) == (64, 3, 7, 7) and (c.stride, c.padding, c.dilation) == \
                    ((2, 2), (3, 3), (1, 1)) and x.numel() < (1 << 31):
                wp = getattr(c, '_stem_wp', None)
                if wp is None or wp[0] is not w:
                    wp = c._stem_wp = (w, hip_ops.stem_weights(w))
                y = hip_ops.stem_conv(x, wp[1])          # csrc/stem_conv.hip: 7x7/s2 on the matrix cores
            else:
                y = F.conv2d(x, w.to(torch.bfloat16), None, c.stride, c.padding, c.dilation)
            return hip_ops.bias_relu_maxpool(y, b)
        return mp(conv_bn(x, c, bn, relu=True))

    def forward(self, x):
        x = self._stem(x)
        outs = []
        for i, name in enumerate(self.res_layers):
            x = getattr(self, name)(x)
            if i in self.out_indices:
                # a stage output has other consumers (the neck): they may deposit on its token (necks.FPN), but the LAST
                # stage's output has no convolution here to finish its gradient
                if hasattr(x, '_oadg_token') and i == len(self.res_layers) - 1:
                    x._oadg_token = None
                outs.append(x)
        return tuple(outs)

    def train(self, mode=True):
        super().train(mode)
        self._freeze_stages()
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, _BatchNorm):
                    m.eval()
        return self
