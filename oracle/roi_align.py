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


def _axis_samples_vec(start, bin_size, pooled, grid, size):
    """_axis_samples for a batch of RoIs sharing the same `grid` (all float32, elementwise identical)."""
    p = np.arange(pooled, dtype=np.float32)[None, :, None]
    i = (np.arange(grid, dtype=np.float32) + np.float32(0.5))[None, None, :]
    start = start.astype(np.float32)[:, None, None]
    b = bin_size.astype(np.float32)[:, None, None]
    c = ((start + p * b).astype(np.float32) + ((i * b).astype(np.float32) / np.float32(grid)).astype(np.float32))
    c = c.astype(np.float32)
    ok = ~((c < -1.0) | (c > size))
    c = np.where(c <= 0, np.float32(0), c)
    low = c.astype(np.int64)
    edge = low >= size - 1
    low = np.where(edge, size - 1, low)
    high = np.where(edge, size - 1, low + 1)
    c = np.where(edge, low.astype(np.float32), c)
    l = (c - low.astype(np.float32)).astype(np.float32)
    h = (np.float32(1.0) - l).astype(np.float32)
    return low, high, l, h, ok


def _roi_groups(feat_shape, rois, out_size, spatial_scale, sampling_ratio, aligned):
    """Geometry shared by forward and backward: for every (grid_h, grid_w) group the RoI indices, flattened
    corner indices into [N*H*W], the four bilinear weights (already masked by `inside`) and the counts."""
    PH, PW = (out_size, out_size) if isinstance(out_size, int) else out_size
    N, C, H, W = feat_shape
    r = rois.detach().cpu().numpy().astype(np.float32)
    K = r.shape[0]
    sc = f32(spatial_scale)
    off = f32(0.5 if aligned else 0.0)
    b = r[:, 0].astype(np.int64)
    sw, sh = (r[:, 1] * sc - off).astype(np.float32), (r[:, 2] * sc - off).astype(np.float32)
    ew, eh = (r[:, 3] * sc - off).astype(np.float32), (r[:, 4] * sc - off).astype(np.float32)
    rw, rh = (ew - sw).astype(np.float32), (eh - sh).astype(np.float32)
    if not aligned:
        rw, rh = np.maximum(rw, f32(1)), np.maximum(rh, f32(1))
    bh, bw = (rh / f32(PH)).astype(np.float32), (rw / f32(PW)).astype(np.float32)
    if sampling_ratio > 0:
        gh = np.full(K, sampling_ratio, np.int64)
        gw = np.full(K, sampling_ratio, np.int64)
    else:
        gh = np.ceil((rh / f32(PH)).astype(np.float32)).astype(np.int64)
        gw = np.ceil((rw / f32(PW)).astype(np.float32)).astype(np.int64)
    count = np.maximum(gh * gw, 1).astype(np.float32)
    valid = (gh > 0) & (gw > 0) & (b >= 0) & (b < N)
    groups = []
    for g_h, g_w in sorted(set(zip(gh[valid].tolist(), gw[valid].tolist()))):
        idx = np.nonzero(valid & (gh == g_h) & (gw == g_w))[0]
        yl, yh, ly, hy, iny = _axis_samples_vec(sh[idx], bh[idx], PH, g_h, H)     # [n,PH,gh]
        xl, xh, lx, hx, inx = _axis_samples_vec(sw[idx], bw[idx], PW, g_w, W)     # [n,PW,gw]
        base = (b[idx] * H * W)[:, None, None, None, None]

        def lin(yy, xx):
            return torch.as_tensor(base + yy[:, :, :, None, None] * W + xx[:, None, None, :, :])
        ins = (iny[:, :, :, None, None] & inx[:, None, None, :, :]).astype(np.float32)
        T = lambda a: torch.as_tensor(np.ascontiguousarray(a))  # noqa: E731
        ws = [T(hy[:, :, :, None, None] * hx[:, None, None, :, :]), T(hy[:, :, :, None, None] * lx[:, None, None, :, :]),
              T(ly[:, :, :, None, None] * hx[:, None, None, :, :]), T(ly[:, :, :, None, None] * lx[:, None, None, :, :])]
        ids = [lin(yl, xl), lin(yl, xh), lin(yh, xl), lin(yh, xh)]
        groups.append((idx, g_h, g_w, ids, ws, T(ins), torch.as_tensor(count[idx])))
    return K, (PH, PW), groups


class _RoIAlignOracle(torch.autograd.Function):
    """forward: the float32 arithmetic of the kernel source, vectorised over the RoIs that share a sampling
    grid; backward: g/count times the same four weights scattered with index_add_ (what the CUDA kernel's
    atomicAdd does, without the atomics)."""

    @staticmethod
    def forward(ctx, feat, rois, out_size, spatial_scale, sampling_ratio, aligned):
        N, C, H, W = feat.shape
        K, (PH, PW), groups = _roi_groups(feat.shape, rois, out_size, spatial_scale, sampling_ratio, aligned)
        flat = feat.detach().float().permute(0, 2, 3, 1).reshape(N * H * W, C)
        out = feat.new_zeros((K, C, PH, PW), dtype=torch.float32)
        for idx, g_h, g_w, ids, ws, ins, cnt in groups:
            val = ws[0][..., None] * flat[ids[0]] + ws[1][..., None] * flat[ids[1]] + \
                ws[2][..., None] * flat[ids[2]] + ws[3][..., None] * flat[ids[3]]         # [n,PH,gh,PW,gw,C]
            val = val * ins[..., None]
            acc = feat.new_zeros((len(idx), PH, PW, C), dtype=torch.float32)
            for iy in range(g_h):               # reference order: iy outer, ix inner, sequential fp32 adds
                for ix in range(g_w):
                    acc = acc + val[:, :, iy, :, ix, :]
            out[torch.as_tensor(idx)] = (acc / cnt[:, None, None, None]).permute(0, 3, 1, 2)
        ctx.geom = (feat.shape, groups)
        return out

    @staticmethod
    def backward(ctx, gout):
        (N, C, H, W), groups = ctx.geom
        gflat = gout.new_zeros((N * H * W, C), dtype=torch.float32)
        for idx, g_h, g_w, ids, ws, ins, cnt in groups:
            g = gout[torch.as_tensor(idx)].float().permute(0, 2, 3, 1) / cnt[:, None, None, None]     # [n,PH,PW,C]
            g = g[:, :, None, :, None, :] * ins[..., None]                                           # [n,PH,gh,PW,gw,C]
            for k in range(4):
                gflat.index_add_(0, ids[k].reshape(-1), (g * ws[k][..., None]).reshape(-1, C))
        return gflat.view(N, H, W, C).permute(0, 3, 1, 2), None, None, None, None, None


def roi_align(feat, rois, out_size, spatial_scale, sampling_ratio=0, aligned=True):
    """feat [N,C,H,W], rois [K,5] (batch,x1,y1,x2,y2) -> [K,C,PH,PW] float32.  Bit-identical to
    :func:`roi_align_scalar` (the line-by-line restatement; cross-checked in tests/test_oracle_ops.py)."""
    if rois.shape[0] == 0:
        PH, PW = (out_size, out_size) if isinstance(out_size, int) else out_size
        return feat.new_zeros((0, feat.shape[1], PH, PW))
    return _RoIAlignOracle.apply(feat, rois, out_size, spatial_scale, sampling_ratio, aligned)


def roi_align_scalar(feat, rois, out_size, spatial_scale, sampling_ratio=0, aligned=True):
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
    if K == 0:
        return feats[0].new_zeros((0, C, PH, PW))
    lv = map_roi_levels(rois, len(feats), finest_scale)
    out = feats[0].new_zeros((K, C, PH, PW), dtype=torch.float32)
    for l, f in enumerate(feats):
        idx = (lv == l).nonzero(as_tuple=False).view(-1)
        if idx.numel():   # roi_feats[inds] = roi_feats_t (single_level_roi_extractor.py:133-135)
            out = out.index_copy(0, idx, roi_align(f, rois[idx], (PH, PW), 1.0 / strides[l], sampling_ratio,
                                                   aligned))
    return out
