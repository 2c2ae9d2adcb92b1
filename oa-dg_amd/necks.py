"""FPN neck (mmdet/models/necks/fpn.py:12-205) for the configurations the named configs use:
add_extra_convs=False, no norm/activation on the lateral/output convs, nearest-neighbour top-down path,
extra levels by stride-2 subsampling (max_pool2d with kernel 1)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hip_ops, layers

from .layers import ConvModule, xavier_init
from .registry import NECKS


@NECKS.register_module()
class FPN(nn.Module):

    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1,
                 add_extra_convs=False, relu_before_extra_convs=False, no_norm_on_lateral=False,
                 conv_cfg=None, norm_cfg=None, act_cfg=None, upsample_cfg=dict(mode='nearest'),
                 init_cfg=dict(type='Xavier', layer='Conv2d', distribution='uniform')):
        super().__init__()
        assert isinstance(in_channels, list)
        if add_extra_convs:
            raise NotImplementedError('add_extra_convs is not used by the named configs')
        assert norm_cfg is None and act_cfg is None
        self.in_channels, self.out_channels = in_channels, out_channels
        self.num_ins, self.num_outs = len(in_channels), num_outs
        self.upsample_cfg = dict(upsample_cfg)
        if end_level == -1:
            self.backbone_end_level = self.num_ins
            assert num_outs >= self.num_ins - start_level
        else:
            self.backbone_end_level = end_level
            assert end_level <= len(in_channels) and num_outs == end_level - start_level
        self.start_level, self.end_level = start_level, end_level
        self.lateral_convs = nn.ModuleList()
        self.fpn_convs = nn.ModuleList()
        for i in range(self.start_level, self.backbone_end_level):
            self.lateral_convs.append(ConvModule(in_channels[i], out_channels, 1, act_cfg=None, inplace=False))
            self.fpn_convs.append(ConvModule(out_channels, out_channels, 3, padding=1, act_cfg=None,
                                             inplace=False))
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                xavier_init(m, distribution='uniform')

    @staticmethod
    def _lateral(conv, x, out_token=None):
        # a backbone stage output carries the GradToken of the next stage's first convolution, which finishes its
        # gradient: the lateral's data gradient is deposited there instead of returned (hip_conv.GradToken).
        # out_token (finest level): the 3x3 output convolution is the only reader of lateral + top-down, so its data
        # gradient IS this convolution's output gradient and its column sums are this convolution's bias gradient.
        tok = getattr(x, '_oadg_token', None)
        if (tok is not None or out_token is not None) and not (conv.with_norm or conv.with_activation):
            c = conv.conv
            return layers.conv2d(x, c.weight, c.bias, c.stride, c.padding, c.dilation, dep_token=tok, out_token=out_token,
                                 owner=c)
        return conv(x)

    def _merge(self, lat, top):
        """lat + upsample(top) (fpn.py:166-175) for the levels whose lateral convolution did not take the add with it"""
        if 'scale_factor' in self.upsample_cfg:
            return lat + F.interpolate(top, **self.upsample_cfg)
        if top.is_cuda and top.dtype == torch.bfloat16 and lat.dtype == torch.bfloat16 and \
                self.upsample_cfg.get('mode') == 'nearest' and top.shape[1] % 8 == 0:
            return hip_ops.fpn_topdown(lat, top)                          # fused, csrc/eltwise.hip
        return lat + F.interpolate(top, size=lat.shape[2:], **self.upsample_cfg)

    @staticmethod
    def _fpn_conv(conv, x, token, in_token=None):
        # an output level is read by the RPN convolution and by RoIAlign: the RPN convolution's data gradient finishes
        # the level's gradient (RoIAlign deposits its part on the token, hip_ops.roi_align_fpn) and hands this
        # convolution its bias gradient as the column sums of that launch.  Not for the level the extra levels are
        # subsampled from (a third consumer).
        from . import hip_conv
        if token and not (conv.with_norm or conv.with_activation) and hip_conv.ENABLED and x.is_cuda and \
                torch.is_grad_enabled() and (x.dtype == torch.bfloat16 or torch.is_autocast_enabled()):
            c = conv.conv
            tok = hip_conv.GradToken(masked=False)
            y = layers.conv2d(x, c.weight, c.bias, c.stride, c.padding, c.dilation, out_token=tok, in_token=in_token,
                              owner=c)
            if getattr(y.grad_fn, 'name', lambda: '')().startswith('_Conv2dMFMA'):
                y._oadg_token = tok
            return y
        return conv(x)

    def forward(self, inputs):
        assert len(inputs) == len(self.in_channels)
        from . import hip_conv
        x0 = inputs[self.start_level]
        t_lat0 = hip_conv.GradToken(masked=False) if (hip_conv.ENABLED and x0.is_cuda and torch.is_grad_enabled() and
                                                      (x0.dtype == torch.bfloat16 or torch.is_autocast_enabled())) else None
        n = len(self.lateral_convs)
        laterals = [None] * n
        for i in range(n - 1, -1, -1):
            # coarsest level first: level i adds the MERGED level i + 1 (fpn.py:166-175).  Where the lateral convolution runs
            # on the streaming pointwise kernel with power-of-two maps (P2 / P3 of the 1024 x 2048 configs) the add happens
            # in ITS epilogue - the un-merged lateral map is never written and re-read (hip_conv ConvArgs.res_up)
            conv, x = self.lateral_convs[i], inputs[i + self.start_level]
            top = laterals[i + 1] if i < n - 1 else None
            out_token = t_lat0 if i == 0 else None
            y = None
            if top is not None and self.upsample_cfg.get('mode') == 'nearest' and 'scale_factor' not in self.upsample_cfg \
                    and not (conv.with_norm or conv.with_activation) and \
                    (x.dtype == torch.bfloat16 or torch.is_autocast_enabled()) and \
                    hip_conv.topdown_ok(x, conv.conv.out_channels, top):
                c = conv.conv
                y = hip_conv.conv2d(x, c.weight, c.bias, c.stride, c.padding, c.dilation, residual=top, owner=c,
                                    dep_token=getattr(x, '_oadg_token', None), out_token=out_token, res_up=True)
            if y is None:
                y = self._lateral(conv, x, out_token)
                if top is not None:
                    y = self._merge(y, top)
            laterals[i] = y
        outs = [self._fpn_conv(self.fpn_convs[i], laterals[i], token=i < n - 1 or self.num_outs == n,
                               in_token=t_lat0 if (i == 0 and n > 1) else None) for i in range(n)]
        for _ in range(self.num_outs - len(outs)):   # fpn.py:184-188
            # fpn.py:177-181 `F.max_pool2d(outs[-1], 1, stride=2)`: a kernel-1 pool is a strided subsample - the same
            # values without the pooling library (whose kernels are compiled per shape)
            last = outs[-1][:, :, ::2, ::2]
            outs.append(last.contiguous(memory_format=torch.channels_last) if last.dim() == 4 and last.is_cuda else last)
        return tuple(outs)
