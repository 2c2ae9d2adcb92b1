"""Closed-form known answers for the leaves whose arithmetic lives in dependencies absent from the reference tree
(mmcv.ops.RoIAlign / nms, cv2.warpAffine / resize / GaussianBlur; SURVEY.md A.3-A.5, call sites
mmdet/models/roi_heads/roi_extractors/base_roi_extractor.py:54-59, mmdet/models/dense_heads/rpn_head.py:231,
mmdet/datasets/pipelines/oa_mix.py:74-120, augmix.py:83-188).

The oracle cannot be PINNED to those libraries here (parity unpinned); these tests remove the
"both sides share one misreading" risk instead: every expected value below is derived by hand from the published
definition, not from the oracle, and is asserted on the oracle (CPU) AND on the HIP kernels (-m gpu, through the C ABI).
"""
import ctypes

import numpy as np
import pytest
import torch

from oracle import cvleaves as cv
from oracle import nms as ONMS
from oracle import roi_align as R


# ------------------------------------------------------------------------------------------------ RoIAlign
def _ramp(C, H, W):
    """f[c, y, x] = a_c * x + b_c * y + c_c: bilinear interpolation reproduces it exactly inside the map"""
    a = np.array([1.0, -0.5, 0.25, 2.0][:C], np.float32)
    b = np.array([0.5, 3.0, -1.0, 0.125][:C], np.float32)
    c = np.array([10.0, -4.0, 0.5, 7.0][:C], np.float32)
    y, x = np.mgrid[0:H, 0:W].astype(np.float32)
    return (a[:, None, None] * x + b[:, None, None] * y + c[:, None, None])[None], a, b, c


RAMP_ROIS = np.array([[0, 8.0, 12.0, 64.0, 52.0],       # 14 x 10 map pixels at scale 1/4: grid 2 x 2
                      [0, 3.0, 2.0, 150.0, 100.0],      # 36.75 x 24.5: grid ceil(5.25) = 6 x ceil(3.5) = 4
                      [0, 20.5, 30.25, 48.5, 58.25],    # exactly 7 x 7 map pixels: grid 1 x 1
                      [0, 40.0, 40.0, 41.0, 40.5]],     # smaller than one bin: grid 1 x 1, bins 1/28 px
                     np.float32)


def _ramp_expected(a, b, c, rois, scale, P=7):
    """mmcv aligned=True: x1' = x1 * scale - 0.5; bin = roi / P; the samples of a bin are symmetric about its centre,
    so the mean of a LINEAR function over them is its value at the bin centre - whatever the sampling grid is."""
    out = np.zeros((len(rois), len(a), P, P), np.float64)
    for k, (_, x1, y1, x2, y2) in enumerate(rois.astype(np.float64)):
        xs, ys = x1 * scale - 0.5, y1 * scale - 0.5
        bw, bh = (x2 - x1) * scale / P, (y2 - y1) * scale / P
        xc = xs + (np.arange(P) + 0.5) * bw
        yc = ys + (np.arange(P) + 0.5) * bh
        out[k] = a[:, None, None] * xc[None, None, :] + b[:, None, None] * yc[None, :, None] + c[:, None, None]
    return out


def _roi_align_impls(dev):
    impls = [('oracle', lambda f, r, s: R.roi_align(f, r, 7, s))]
    if dev is not None:
        from oadg_amd import hip_ops
        impls.append(('hip', lambda f, r, s: hip_ops.roi_align_fpn(
            [f.to(dev).contiguous(memory_format=torch.channels_last)], r.to(dev), 7, [s]).cpu()))
    return impls


def _check_roi_align(dev):
    f, a, b, c = _ramp(4, 40, 60)
    rois = torch.tensor(RAMP_ROIS)
    exp = _ramp_expected(a, b, c, RAMP_ROIS, 0.25)
    for name, fn in _roi_align_impls(dev):
        got = fn(torch.tensor(f), rois, 0.25).double().numpy()
        assert np.abs(got - exp).max() <= 2e-4, (name, np.abs(got - exp).max())     # fp32 sums of values ~ 1e2

    # the sampling grid is ceil(roi / pooled) per axis (sampling_ratio = 0): f = x^2 is NOT reproduced by bilinear
    # interpolation - between the integers k, k+1 the interpolant is P(x) = k^2 + (2k + 1)(x - k) - so the bin mean
    # depends on where the samples sit.
    H = W = 40
    xx = np.arange(W, dtype=np.float32)
    sq = np.broadcast_to((xx * xx)[None, None, None, :], (1, 4, H, W)).copy()      # (the kernel takes C % 4 == 0)

    def P_(x):
        k = np.floor(x)
        return k * k + (2 * k + 1) * (x - k)
    for x1, x2, grid in ((0.5, 14.5, 2), (0.5, 15.5, 3), (2.5, 9.5, 1)):
        roi = torch.tensor([[0, x1, 4.5, x2, 18.5]], dtype=torch.float32)
        bw = (x2 - x1) / 7.0
        assert int(np.ceil(bw)) == grid
        exp_x = np.array([np.mean([P_(x1 - 0.5 + pw * bw + (i + 0.5) * bw / grid) for i in range(grid)])
                          for pw in range(7)])
        wrong = np.array([P_(x1 - 0.5 + (pw + 0.5) * bw) for pw in range(7)])          # what a 1-sample grid would give
        if grid > 1:
            assert np.abs(wrong - exp_x).max() > 0.1
        for name, fn in _roi_align_impls(dev):
            got = fn(torch.tensor(sq), roi, 1.0)[0, 0].double().numpy()
            assert np.abs(got - exp_x[None, :]).max() <= 1e-3, (name, grid, got[0], exp_x)

    # borders: samples with coordinate < -1 or > size contribute 0, samples in [-1, 0] are clamped to pixel 0.
    # Constant map 3.0, RoI x1' = -3, seven 1-px bins (grid 1): sample x = -2.5, -1.5 -> 0; -0.5 -> clamped -> 3; rest 3
    const = torch.full((1, 4, 20, 20), 3.0)
    roi = torch.tensor([[0, -2.5, 4.5, 4.5, 11.5]])
    for name, fn in _roi_align_impls(dev):
        got = fn(const, roi, 1.0)[0, 0].numpy()
        assert np.allclose(got, np.array([0, 0, 3, 3, 3, 3, 3], np.float32)[None, :], atol=1e-6), (name, got[0])
    far = torch.tensor([[0, -500.0, -500.0, -400.0, -400.0]])
    for name, fn in _roi_align_impls(dev):
        assert float(fn(const, far, 1.0).abs().max()) == 0.0, name


def _check_roi_align_backward(dev):
    """adjoint of the ramp property: for a RoI whose samples all lie inside the map, the gradient that reaches the map
    has  sum = sum(g)  (the bilinear weights of a sample add up to 1, every bin is a mean) and first moments
    sum(grad * x) = sum_bins g * x_centre, sum(grad * y) likewise."""
    H, W, scale = 40, 60, 0.25
    rois = torch.tensor(RAMP_ROIS[:3])
    g = torch.tensor(np.random.RandomState(0).standard_normal((3, 4, 7, 7)).astype(np.float32))
    impls = [('oracle', lambda f: R.roi_align(f, rois, 7, scale), 'cpu')]
    if dev is not None:
        from oadg_amd import hip_ops
        impls.append(('hip', lambda f: hip_ops.roi_align_fpn([f], rois.to(dev), 7, [scale]), dev))
    yy, xx = np.mgrid[0:H, 0:W].astype(np.float64)
    for name, fn, d in impls:
        f = torch.zeros((1, 4, H, W), device=d, requires_grad=True)
        fn(f).backward(g.to(d))
        gr = f.grad.detach().cpu().double().numpy()[0]
        for ch in range(4):
            gs = g[:, ch].double().numpy()
            exp0 = gs.sum()
            exp_x = exp_y = 0.0
            for k, (_, x1, y1, x2, y2) in enumerate(RAMP_ROIS[:3].astype(np.float64)):
                xc = x1 * scale - 0.5 + (np.arange(7) + 0.5) * (x2 - x1) * scale / 7
                yc = y1 * scale - 0.5 + (np.arange(7) + 0.5) * (y2 - y1) * scale / 7
                exp_x += (gs[k] * xc[None, :]).sum()
                exp_y += (gs[k] * yc[:, None]).sum()
            assert abs(gr[ch].sum() - exp0) <= 1e-4 * (1 + abs(exp0)), (name, gr[ch].sum(), exp0)
            assert abs((gr[ch] * xx).sum() - exp_x) <= 1e-3 * (1 + abs(exp_x)), name
            assert abs((gr[ch] * yy).sum() - exp_y) <= 1e-3 * (1 + abs(exp_y)), name


def test_roi_align_closed_forms_oracle():
    _check_roi_align(None)
    _check_roi_align_backward(None)


@pytest.mark.gpu
def test_roi_align_closed_forms_hip(dev):
    _check_roi_align(dev)
    _check_roi_align_backward(dev)


# ------------------------------------------------------------------------------------------------ NMS
NMS_CASES = [
    # boxes in descending-score order, threshold, expected keep
    # chain A-B-C: IoU(A,B) = IoU(B,C) = 90/110, IoU(A,C) = 80/120 = 0.667: B falls to A and can no longer suppress C
    ([[0, 0, 10, 10], [1, 0, 11, 10], [2, 0, 12, 10]], 0.7, [0, 2]),
    # the test is IoU > thr (strict): 50 / 100 = 0.5 exactly survives at thr 0.5 and falls at 0.49
    ([[0, 0, 10, 10], [0, 0, 10, 5]], 0.5, [0, 1]),
    ([[0, 0, 10, 10], [0, 0, 10, 5]], 0.49, [0]),
    # areas without the legacy "+1": IoU of unit squares offset by half = 0.5 / 1.5 = 1/3 (with +1 it would be 0.5)
    ([[0, 0, 1, 1], [0.5, 0, 1.5, 1]], 0.4, [0, 1]),
    # identical boxes: IoU 1; a zero-area box has IoU 0/0 = NaN with itself and 0 with others: never suppressed
    ([[5, 5, 9, 9], [5, 5, 9, 9], [7, 7, 7, 7], [7, 7, 7, 7]], 0.3, [0, 2, 3]),
]


def _check_nms(dev):
    for boxes, thr, expect in NMS_CASES:
        b = np.asarray(boxes, np.float32)
        assert ONMS.nms_sorted(b, thr).tolist() == expect, (boxes, thr)
        if dev is not None:
            from oadg_amd import hip_ops
            k, kc = hip_ops.nms_sorted_batched(torch.tensor(b, device=dev)[None],
                                               torch.tensor([len(b)], device=dev, dtype=torch.int32), thr)
            assert k[0, :int(kc[0])].tolist() == expect, (boxes, thr)
    # batched_nms: the same box under two ids is kept twice (coordinates are offset by id * (max + 1))
    bx = torch.tensor([[0, 0, 10, 10], [0, 0, 10, 10], [0, 0, 10, 10]], dtype=torch.float32)
    dets, keep = ONMS.batched_nms(bx, torch.tensor([0.9, 0.8, 0.7]), torch.tensor([0, 1, 0]), dict(type='nms', iou_threshold=0.5))
    assert keep.tolist() == [0, 1] and torch.equal(dets[:, :4], bx[:2]) and dets[:, 4].tolist() == pytest.approx([0.9, 0.8])


def test_nms_hand_built_cases_oracle():
    _check_nms(None)


@pytest.mark.gpu
def test_nms_hand_built_cases_hip(dev):
    _check_nms(dev)


# ------------------------------------------------------------------------------------------------ warpAffine / resize
def _translated(img, tx, ty):
    """dst(x, y) = src(x - tx, y - ty), zeros outside (cv2.warpAffine with M = [[1,0,tx],[0,1,ty]], BORDER_CONSTANT)"""
    H, W = img.shape[:2]
    out = np.zeros_like(img)
    xs0, xs1 = max(0, tx), min(W, W + tx)
    ys0, ys1 = max(0, ty), min(H, H + ty)
    if xs1 > xs0 and ys1 > ys0:
        out[ys0:ys1, xs0:xs1] = img[ys0 - ty:ys1 - ty, xs0 - tx:xs1 - tx]
    return out


def _hip_warp(img, M, dev):
    """the product's warp: one bbox-only step whose blend mask is 1 everywhere (b = 1 -> the warped pixel is stored)"""
    from oadg_amd import _lib
    from oadg_amd.pipelines.oa_mix import invert_affine
    L = _lib.lib()
    H, W = img.shape[:2]
    t = torch.tensor(img, device=dev).contiguous()
    scratch = torch.empty_like(t)
    ones_y = torch.ones((H,), device=dev)
    ones_x = torch.ones((W,), device=dev)
    minv = (ctypes.c_double * 6)(*invert_affine(M))
    _lib.check(L.oadg_oamix_bbox_step(_lib.ptr(t), H, W, minv, 0, 0, W, H, _lib.ptr(ones_y), _lib.ptr(ones_x),
                                      _lib.ptr(scratch), _lib.stream_ptr()), 'oadg_oamix_bbox_step')
    return t.cpu().numpy()


def _check_warp(dev):
    rs = np.random.RandomState(3)
    img = rs.randint(0, 256, (37, 53, 3)).astype(np.uint8)
    warps = [('oracle', lambda M: cv.warp_affine(img, M))]
    if dev is not None:
        warps.append(('hip', lambda M: _hip_warp(img, M, dev)))
    for name, fn in warps:
        assert np.array_equal(fn(np.float32([[1, 0, 0], [0, 1, 0]])), img), name           # identity is exact
        for tx, ty in ((5, 0), (0, -7), (-11, 3), (60, 0), (0, 40)):
            got = fn(np.float32([[1, 0, tx], [0, 1, ty]]))
            assert np.array_equal(got, _translated(img, tx, ty)), (name, tx, ty)           # integer shifts are exact
        # half-pixel shift: the mean of horizontal neighbours, rounded half up ((a + b) * 2^14 + 2^14) >> 15
        got = fn(np.float64([[1, 0, 0.5], [0, 1, 0]])).astype(int)
        a = img[:, :-1].astype(int)
        b = img[:, 1:].astype(int)
        assert np.array_equal(got[:, 1:], (a + b + 1) >> 1), name
        # 180 degree rotation about the centre of the pixel grid is an exact flip of both axes
        H, W = img.shape[:2]
        M = cv.get_rotation_matrix_2d(((W - 1) / 2.0, (H - 1) / 2.0), 180, 1.0)
        assert np.array_equal(fn(M), img[::-1, ::-1]), name


def test_warp_affine_exact_cases_oracle():
    _check_warp(None)


@pytest.mark.gpu
def test_warp_affine_exact_cases_hip(dev):
    _check_warp(dev)


def test_resize_identity_and_integer_upscale_oracle():
    rs = np.random.RandomState(4)
    img = rs.randint(0, 256, (19, 23, 3)).astype(np.uint8)
    assert np.array_equal(cv.resize_u8_cv2(img, (23, 19)), img)                              # x1 is exact
    flat = np.full((8, 8, 3), 77, np.uint8)
    assert np.array_equal(cv.resize_u8_cv2(flat, (20, 13)), np.full((13, 20, 3), 77, np.uint8))
    # x2 of a horizontal ramp 0, 10, 20, ...: half-pixel centres give 1/4 - 3/4 mixes, edges clamp
    ramp = np.tile((np.arange(8, dtype=np.uint8) * 10)[None, :, None], (4, 1, 3))
    up = cv.resize_u8_cv2(ramp, (16, 4))[0, :, 0].astype(int)
    inner = np.array([round(v) for v in np.interp((np.arange(16) + 0.5) / 2 - 0.5, np.arange(8), np.arange(8) * 10.0)])
    assert np.abs(up - inner).max() <= 1                                                     # 11-bit coefficient rounding
    assert up[0] == 0 and up[-1] == 70


@pytest.mark.gpu
def test_resize_identity_hip(dev):
    from oadg_amd.pipelines.geometric import Resize
    rs = np.random.RandomState(4)
    img = rs.randint(0, 256, (64, 96, 3)).astype(np.uint8)
    out, _, _ = Resize(img_scale=(96, 64), keep_ratio=False)(torch.tensor(img, device=dev), np.zeros((0, 4), np.float32))
    assert np.array_equal(out.cpu().numpy(), img)
    flat = torch.full((40, 40, 3), 77, dtype=torch.uint8, device=dev)
    out, _, _ = Resize(img_scale=(100, 65), keep_ratio=False)(flat, np.zeros((0, 4), np.float32))
    assert out.shape[:2] == (65, 100) and bool((out == 77).all())


# ------------------------------------------------------------------------------------------------ blurred box mask
def _check_mask(profiles):
    """_get_mask (oa_mix.py:74-93): quarter-resolution box indicator, GaussianBlur(sigma = 0.2 * side), x4 bilinear.
    The Gaussian kernel is normalised and the bilinear x4 resize spreads every source sample over four outputs with
    weights summing to 4, so for a box whose blur support stays inside the image  sum(M) = 4 * (side / 4)  per axis;
    the profile is symmetric about the box centre and falls monotonically away from it; far from the box it is 0."""
    H, W = 512, 640
    box = np.array([200.0, 160.0, 360.0, 288.0], np.float32)         # 160 x 128 px: quarter-res 40 x 32
    my, mx = profiles(box, H, W)
    assert abs(my.sum() - 128.0) <= 1e-3 * 128 and abs(mx.sum() - 160.0) <= 1e-3 * 160
    cy, cx = (160 + 288) / 2.0, (200 + 360) / 2.0
    for m, c in ((my, cy), (mx, cx)):
        i = np.arange(len(m))
        lo, hi = m[i < c], m[i >= c]
        assert np.allclose(lo[::-1][:len(hi)][:200], hi[:200], atol=2e-6)            # symmetric about the centre
        assert (np.diff(hi) <= 1e-7).all() and (np.diff(lo) >= -1e-7).all()           # monotone on either side
        assert m.max() <= 1.0 + 1e-6 and m.min() >= 0.0
        assert m[0] == 0.0 and m[-1] == 0.0                                           # far from the box
    # the centre of a box much wider than the kernel is (almost) fully inside; the rim value is ~ 1/2
    assert my[int(cy)] > 0.98 and abs(my[160] - 0.5) < 0.06 and abs(mx[360] - 0.5) < 0.06


def test_blurred_mask_profile_properties_oracle():
    _check_mask(lambda box, H, W: cv.box_mask_profiles(box, H, W, 4, 0.3))


@pytest.mark.gpu
def test_blurred_mask_profile_properties_hip(dev):
    from oadg_amd.pipelines.oa_mix import _ImageState

    def profiles(box, H, W):
        st = _ImageState(torch.zeros((H, W, 3), dtype=torch.uint8, device=dev), box[None], 4, 0.3)
        return st.My[0].cpu().numpy(), st.Mx[0].cpu().numpy()
    _check_mask(profiles)
