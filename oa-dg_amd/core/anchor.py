"""Anchor grid generation (mmdet/core/anchor/anchor_generator.py AnchorGenerator,
mmdet/core/anchor/utils.py:19-43,5-16)."""
import numpy as np
import torch

from ..registry import PRIOR_GENERATORS


def _pair(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


@PRIOR_GENERATORS.register_module()
class AnchorGenerator:
    """Multi-level anchors.  RPN of the named configs: scales=[8], ratios=[.5,1,2],
    strides=[4,8,16,32,64] -> 3 anchors per location, 523,776 per 1024x2048 image."""

    def __init__(self, strides, ratios, scales=None, base_sizes=None, scale_major=True,
                 octave_base_scale=None, scales_per_octave=None, centers=None, center_offset=0.):
        if center_offset != 0:
            assert centers is None
        assert 0 <= center_offset <= 1
        self.strides = [_pair(s) for s in strides]
        self.base_sizes = [min(s) for s in self.strides] if base_sizes is None else list(base_sizes)
        assert len(self.base_sizes) == len(self.strides)
        assert (octave_base_scale is not None and scales_per_octave is not None) ^ (scales is not None)
        if scales is not None:
            self.scales = torch.Tensor(scales)
        else:
            octave = np.array([2 ** (i / scales_per_octave) for i in range(scales_per_octave)])
            self.scales = torch.Tensor(octave * octave_base_scale)
        self.octave_base_scale = octave_base_scale
        self.scales_per_octave = scales_per_octave
        self.ratios = torch.Tensor(ratios)
        self.scale_major = scale_major
        self.centers = centers
        self.center_offset = center_offset
        self.base_anchors = [self._level_base(b, None if centers is None else centers[i])
                             for i, b in enumerate(self.base_sizes)]

    @property
    def num_base_anchors(self):
        return [b.size(0) for b in self.base_anchors]

    num_base_priors = num_base_anchors

    @property
    def num_levels(self):
        return len(self.strides)

    def _level_base(self, base_size, center):
        w = h = base_size
        if center is None:
            xc, yc = self.center_offset * w, self.center_offset * h
        else:
            xc, yc = center
        hr = torch.sqrt(self.ratios)
        wr = 1 / hr
        if self.scale_major:
            ws = (w * wr[:, None] * self.scales[None, :]).view(-1)
            hs = (h * hr[:, None] * self.scales[None, :]).view(-1)
        else:
            ws = (w * self.scales[:, None] * wr[None, :]).view(-1)
            hs = (h * self.scales[:, None] * hr[None, :]).view(-1)
        return torch.stack([xc - 0.5 * ws, yc - 0.5 * hs, xc + 0.5 * ws, yc + 0.5 * hs], dim=-1)

    def single_level_grid_priors(self, featmap_size, level_idx, dtype=torch.float32, device='cuda'):
        """Anchors only depend on (feature-map size, level): computed once per shape and kept on the device
        (the reference rebuilds them, with a host->device copy of the base anchors, on every call)."""
        key = (tuple(int(v) for v in featmap_size), level_idx, dtype, str(device))
        cache = self.__dict__.setdefault('_prior_cache', {})
        if key not in cache:
            cache[key] = self._single_level_grid_priors(featmap_size, level_idx, dtype, device)
        return cache[key]

    def _single_level_grid_priors(self, featmap_size, level_idx, dtype, device):
        base = self.base_anchors[level_idx].to(device).to(dtype)
        fh, fw = featmap_size
        sw, sh = self.strides[level_idx]
        sx = (torch.arange(0, fw, device=device).to(dtype) * sw)
        sy = (torch.arange(0, fh, device=device).to(dtype) * sh)
        xx = sx.repeat(fh)
        yy = sy.view(-1, 1).repeat(1, fw).view(-1)
        shifts = torch.stack([xx, yy, xx, yy], dim=-1)
        # row-major over (y, x), base anchors innermost: index = (y*fw + x)*A + a
        return (base[None, :, :] + shifts[:, None, :]).view(-1, 4)

    def grid_priors(self, featmap_sizes, dtype=torch.float32, device='cuda'):
        assert self.num_levels == len(featmap_sizes)
        return [self.single_level_grid_priors(featmap_sizes[i], i, dtype, device)
                for i in range(self.num_levels)]

    grid_anchors = lambda self, featmap_sizes, device='cuda': self.grid_priors(featmap_sizes, device=device)  # noqa: E731

    def valid_flags(self, featmap_sizes, pad_shape, device='cuda'):
        key = (tuple(tuple(int(v) for v in f) for f in featmap_sizes), tuple(pad_shape[:2]), str(device))
        cache = self.__dict__.setdefault('_flag_cache', {})
        if key not in cache:
            cache[key] = self._valid_flags(featmap_sizes, pad_shape, device)
        return cache[key]

    def _valid_flags(self, featmap_sizes, pad_shape, device):
        flags = []
        for i in range(self.num_levels):
            fh, fw = featmap_sizes[i]
            h, w = pad_shape[:2]
            sw, sh = self.strides[i]
            vh = min(int(np.ceil(h / sh)), fh)
            vw = min(int(np.ceil(w / sw)), fw)
            vx = torch.zeros(fw, dtype=torch.bool, device=device)
            vy = torch.zeros(fh, dtype=torch.bool, device=device)
            vx[:vw] = 1
            vy[:vh] = 1
            xx = vx.repeat(fh)
            yy = vy.view(-1, 1).repeat(1, fw).view(-1)
            valid = xx & yy
            A = self.num_base_anchors[i]
            flags.append(valid[:, None].expand(valid.size(0), A).contiguous().view(-1))
        return flags


def anchor_inside_flags(flat_anchors, valid_flags, img_shape, allowed_border=0):
    """utils.py:19-43."""
    h, w = img_shape[:2]
    if allowed_border >= 0:
        return valid_flags & (flat_anchors[:, 0] >= -allowed_border) & \
            (flat_anchors[:, 1] >= -allowed_border) & (flat_anchors[:, 2] < w + allowed_border) & \
            (flat_anchors[:, 3] < h + allowed_border)
    return valid_flags


def images_to_levels(target, num_levels):
    """per-image [sum_l n_l, ...] list -> per-level [N, n_l, ...] list (utils.py:5-16)."""
    target = torch.stack(target, 0)
    out, start = [], 0
    for n in num_levels:
        out.append(target[:, start:start + n])
        start += n
    return out
