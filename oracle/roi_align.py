"""CPU restatement of mmcv.ops.RoIAlign (avg pooling, aligned, adaptive sampling grid) and of
SingleRoIExtractor's level mapping.  Test infrastructure only (see oracle/__init__).

PARITY UNPINNED: the arithmetic is mmcv-full's (pinned by the reference only as the range
1.3.17 <= mmcv-full <= 1.5.0, mmdet/__init__.py:19-27; not vendored, not installed).  Restated from the
published algorithm (mmcv/ops/csrc/common/cuda/roi_align_cuda_kernel.cuh), anchored on the
reference's call sites: mmdet/models/roi_heads/roi_extractors/base_roi_extractor.py:54-59 (construction:
output_size=7, sampling_ratio=0, spatial_scale=1/stride, mmcv defaults pool_mode='avg', aligned=True) and
single_level_roi_extractor.py:36-55 (levels), :110,134 (calls).  The reference's tests assert shapes only
(tests/test_models/test_roi_heads/test_roi_extractor.py:41-42).

Coordinates and weights are computed in float32 exactly as the kernel source does; feature gathers go through
torch so autograd provides the (atomic-free) backward.
"""
import math

import numpy as np
import torch

f32 = np.float32


def _axis_samples(start, bin_size, pooled, grid, size):
    """Per output bin p and sample i: (low, high, w_low, w_high, inside) along one axis."""
    lo = np.zeros((pooled, max(grid, 0)), np.int64)
    hi = np.zeros_like(lo)
    wl = np.zeros(lo.shape, np.float32)
    wh = np.zeros(lo.shape, np.float32)
    inside = np.zeros(lo.shape, bool)
    for p in range(pooled):
        for i in range(grid):
            c = f32(start + f32(p) * bin_size) + f32(f32(i + 0.5) * bin_size) / f32(grid)
            c = f32(c)
            ok = not (c < -1.0 or c > size)
            if c <= 0:
                c = f32(0)
            low = int(c)
            if low >= size - 1:
                high = low = size - 1
                c = f32(low)
            else:
                high = low + 1
            l = f32(c - f32(low))
            h = f32(f32(1.0) - l)
            lo[p, i], hi[p, i], wl[p, i], wh[p, i], inside[p, i] = low, high, l, h, ok
    return lo, hi, wl, wh, inside


def roi_align(feat, rois, out_size, spatial_scale, sampling_ratio=0, aligned=True):
    """feat [N,C,H,W] float tensor, rois [K,5] (batch,x1,y1,x2,y2) -> [K,C,PH,PW] float32."""
    PH, PW = (out_size, out_size) if isinstance(out_size, int) else out_size
    N, C, H, W = feat.shape
    feat = feat.float()
    r = rois.detach().cpu().numpy().astype(np.float32)
    outs = []
    sc = f32(spatial_scale)
    off = f32(0.5 if aligned else 0.0)
    for k in range(r.shape[0]):
        b = int(r[k, 0])
        sw, sh = f32(r[k, 1] * sc - off), f32(r[k, 2] * sc - off)
        ew, eh = f32(r[k, 3] * sc - off), f32(r[k, 4] * sc - off)
        rw, rh = f32(ew - sw), f32(eh - sh)
        if not aligned:
            rw, rh = max(rw, f32(1)), max(rh, f32(1))
        bh, bw = f32(rh / f32(PH)), f32(rw / f32(PW))
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(f32(rh / f32(PH))))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(f32(rw / f32(PW))))
        count = f32(max(gh * gw, 1))
        if gh <= 0 or gw <= 0 or b < 0 or b >= N:
            outs.append(feat.new_zeros((C, PH, PW)))
            continue
        yl, yh, wyl, wyh, iny = _axis_samples(sh, bh, PH, gh, H)
        xl, xh, wxl, wxh, inx = _axis_samples(sw, bw, PW, gw, W)
        fm = feat[b]                                               # [C,H,W]
        YL = torch.as_tensor(yl).view(PH, gh, 1, 1); YH = torch.as_tensor(yh).view(PH, gh, 1, 1)
        XL = torch.as_tensor(xl).view(1, 1, PW, gw); XH = torch.as_tensor(xh).view(1, 1, PW, gw)
        hy = torch.as_tensor(wyh).view(PH, gh, 1, 1); ly = torch.as_tensor(wyl).view(PH, gh, 1, 1)
        hx = torch.as_tensor(wxh).view(1, 1, PW, gw); lx = torch.as_tensor(wxl).view(1, 1, PW, gw)
        ins = (torch.as_tensor(iny).view(PH, gh, 1, 1) & torch.as_tensor(inx).view(1, 1, PW, gw)).float()
        v1, v2 = fm[:, YL, XL], fm[:, YL, XH]
        v3, v4 = fm[:, YH, XL], fm[:, YH, XH]
        val = (hy * hx) * v1 + (hy * lx) * v2 + (ly * hx) * v3 + (ly * lx) * v4   # [C,PH,gh,PW,gw]
        val = val * ins
        # reference order: iy outer, ix inner, sequential fp32 accumulation
        acc = feat.new_zeros((C, PH, PW))
        for iy in range(gh):
            for ix in range(gw):
                acc = acc + val[:, :, iy, :, ix]
        outs.append(acc / float(count))
    if not outs:
        return feat.new_zeros((0, C, PH, PW))
    return torch.stack(outs)


def map_roi_levels(rois, num_levels, finest_scale=56):
    """single_level_roi_extractor.py:36-55."""
    scale = torch.sqrt((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2]))
    lvl = torch.floor(torch.log2(scale / finest_scale + 1e-6))
    return lvl.clamp(min=0, max=num_levels - 1).long()


def roi_align_fpn(feats, rois, out_size, strides, finest_scale=56, sampling_ratio=0, aligned=True):
    """SingleRoIExtractor.forward, single_level_roi_extractor.py:89-146 (roi_scale_factor=None)."""
    PH, PW = (out_size, out_size) if isinstance(out_size, int) else out_size
    K = rois.shape[0]
    C = feats[0].shape[1]
    if len(feats) == 1:
        return roi_align(feats[0], rois, (PH, PW), 1.0 / strides[0], sampling_ratio, aligned)
    lv = map_roi_levels(rois, len(feats), finest_scale)
    out = [None] * K
    for l, f in enumerate(feats):
        idx = (lv == l).nonzero(as_tuple=False).view(-1)
        if idx.numel():
            o = roi_align(f, rois[idx], (PH, PW), 1.0 / strides[l], sampling_ratio, aligned)
            for j, i in enumerate(idx.tolist()):
                out[i] = o[j]
    if K == 0:
        return feats[0].new_zeros((0, C, PH, PW))
    return torch.stack(out)
