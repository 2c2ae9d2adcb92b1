"""CPU: the oracle's restatements of the un-vendored leaves against INDEPENDENT implementations that do exist in this
image - scipy.ndimage (affine resampling, Gaussian filtering, pixel-centre-aligned zoom), torch's grid_sample (bilinear
sampling) and numpy's FFT - written from the same published definitions by other people.  They are not the libraries
the reference calls (OpenCV / mmcv are absent), so the tolerances cover what legitimately differs (OpenCV's 1/32-px
coordinate quantisation and integer weights, float32 vs float64), and the oracle stays formally "unpinned"; but a shared
misreading of the geometry conventions (pixel centres, matrix direction, border rule, kernel size) would show here.
"""
import numpy as np
import pytest
import scipy.ndimage as ndi
import torch
import torch.nn.functional as F

from inputs import lowpass_image
from oracle import cvleaves as cv
from oracle import roi_align as R


@pytest.mark.parametrize('kind', ['rotate', 'shear_x', 'translate_frac', 'scale'])
def test_warp_affine_against_scipy_affine_transform(kind):
    """cv2.warpAffine(src, M): dst(x, y) = src(M^-1 (x, y)), bilinear, zeros outside.  scipy's affine_transform maps
    OUTPUT coordinates through the matrix it is given, in (row, col) order - i.e. it wants M^-1 with the axes swapped."""
    img = lowpass_image(np.random.RandomState(0), 96, 128, 6)
    H, W = img.shape[:2]
    if kind == 'rotate':
        M = cv.get_rotation_matrix_2d((W / 2, H / 2), 17, 1.0)
    elif kind == 'shear_x':
        M = np.float64([[1, -0.21, 0.21 * 40], [0, 1, 0]])
    elif kind == 'translate_frac':
        M = np.float64([[1, 0, 7.25], [0, 1, -3.5]])
    else:
        M = np.float64([[1.1, 0, -4], [0, 0.9, 3]])
    got = cv.warp_affine(img, M).astype(np.float64)
    Mi = cv.invert_affine(M)                                   # source = Mi @ (x, y, 1)
    A = np.array([[Mi[1, 1], Mi[1, 0]], [Mi[0, 1], Mi[0, 0]]])  # (row, col) order
    off = np.array([Mi[1, 2], Mi[0, 2]])
    ref = np.stack([ndi.affine_transform(img[..., c].astype(np.float64), A, offset=off, order=1, mode='constant', cval=0.0)
                    for c in range(3)], -1)
    # interior: the only differences are OpenCV's 1/32-px source coordinates and 15-bit weights (<= 2 grey levels on
    # this smooth image); at the zero border scipy cuts the last half pixel differently: compare away from it
    valid = ndi.affine_transform(np.ones((H, W)), A, offset=off, order=1, mode='constant', cval=0.0) > 0.999
    valid = ndi.binary_erosion(valid, iterations=2)
    diff = np.abs(got - ref)[valid]
    assert valid.mean() > 0.5 and diff.max() <= 2.5 and diff.mean() <= 0.6, (diff.max(), diff.mean())


@pytest.mark.parametrize('sigma,n', [(1.5, 40), (6.4, 80), (0.7, 16), (12.0, 200)])
def test_gaussian_blur_profile_against_scipy(sigma, n):
    """cv2.GaussianBlur(ksize=(0,0)) on float32: kernel size round(sigma * 8 + 1) | 1, normalised samples of
    exp(-x^2 / 2 sigma^2), BORDER_REFLECT_101 (= scipy 'mirror')."""
    rs = np.random.RandomState(int(sigma * 10))
    a = np.zeros(n, np.float32)
    lo = rs.randint(0, n // 2)
    a[lo:lo + rs.randint(1, n // 2)] = 1.0                       # a box indicator, as OA-Mix blurs
    got = cv.blur_profile_f32(a, sigma)
    radius = ((int(np.rint(sigma * 8 + 1)) | 1) - 1) // 2
    ref = ndi.gaussian_filter1d(a.astype(np.float64), sigma, mode='mirror', truncate=(radius + 0.25) / sigma)
    assert len(cv.gaussian_kernel_f32(sigma)) == 2 * radius + 1
    assert np.abs(got - ref).max() <= 2e-6


@pytest.mark.parametrize('src_n,dst_n', [(32, 128), (50, 200), (17, 68)])
def test_bilinear_resize_profile_against_scipy_zoom(src_n, dst_n):
    """cv2.resize INTER_LINEAR uses pixel-centre alignment (dst centre (d + .5) * scale - .5) and clamps at the edges:
    scipy.ndimage.zoom(order=1, grid_mode=True, mode='nearest') is the same definition."""
    p = np.random.RandomState(src_n).rand(src_n).astype(np.float32)
    got = cv.resize_profile_f32(p, dst_n)
    ref = ndi.zoom(p.astype(np.float64), dst_n / src_n, order=1, grid_mode=True, mode='nearest')
    assert len(ref) == dst_n and np.abs(got - ref).max() <= 1e-6


def test_u8_resize_against_scipy_zoom():
    img = lowpass_image(np.random.RandomState(2), 60, 90, 4)
    got = cv.resize_u8_cv2(img, (135, 80)).astype(np.float64)          # cv2 dsize = (width, height)
    ref = np.stack([ndi.zoom(img[..., c].astype(np.float64), (80 / 60, 135 / 90), order=1, grid_mode=True, mode='nearest')
                    for c in range(3)], -1)
    assert got.shape == ref.shape and np.abs(got - ref).max() <= 1.0     # 11-bit coefficients + rounding: one grey level


def test_roi_align_interior_rois_against_grid_sample():
    """mmcv RoIAlign (aligned=True) = average of bilinear samples at start + ph * bin + (i + .5) * bin / grid with
    start = x1 * scale - 0.5, grid = ceil(roi / pooled).  torch's grid_sample(align_corners=True) interpolates at
    arbitrary pixel-centre coordinates: an independent bilinear sampler for RoIs that stay inside the map."""
    rs = np.random.RandomState(1)
    f = torch.tensor(rs.standard_normal((1, 5, 30, 44)).astype(np.float32))
    H, W = f.shape[2:]
    rois = np.array([[0, 10.3, 8.1, 120.7, 90.2], [0, 40, 40, 47, 43.5], [0, 4.1, 6.0, 171.0, 110.0],
                     [0, 64.0, 32.0, 100.0, 60.0]], np.float32)
    scale, P = 0.25, 7
    got = R.roi_align(f, torch.tensor(rois), P, scale).numpy()
    for k, (_, x1, y1, x2, y2) in enumerate(rois.astype(np.float64)):
        xs, ys = x1 * scale - 0.5, y1 * scale - 0.5
        rw, rh = (x2 - x1) * scale, (y2 - y1) * scale
        gw, gh = int(np.ceil(rw / P)), int(np.ceil(rh / P))
        sx = xs + (np.arange(P)[:, None] + (np.arange(gw)[None, :] + 0.5) / gw) * rw / P        # [P, gw]
        sy = ys + (np.arange(P)[:, None] + (np.arange(gh)[None, :] + 0.5) / gh) * rh / P        # [P, gh]
        assert sx.min() >= 0 and sx.max() <= W - 1 and sy.min() >= 0 and sy.max() <= H - 1
        gx = torch.tensor(sx.reshape(-1) / (W - 1) * 2 - 1, dtype=torch.float32)
        gy = torch.tensor(sy.reshape(-1) / (H - 1) * 2 - 1, dtype=torch.float32)
        grid = torch.stack(torch.meshgrid(gy, gx, indexing='ij')[::-1], -1)[None]                # [1, P*gh, P*gw, (x, y)]
        smp = F.grid_sample(f, grid, mode='bilinear', padding_mode='zeros', align_corners=True)[0]
        ref = smp.reshape(5, P, gh, P, gw).mean(dim=(2, 4)).numpy()
        assert np.abs(got[k] - ref).max() <= 2e-5, k


def test_spectral_residual_map_against_a_direct_dft():
    """the saliency map's DFT -> log-amplitude residual -> inverse DFT chain with an explicit 64-point DFT matrix
    (no FFT library): guards the forward / inverse sign and the missing 1/N of cv::dft(DFT_INVERSE)."""
    img = lowpass_image(np.random.RandomState(3), 80, 112, 5)
    ref_map = cv.spectral_residual_saliency(img)
    gray = cv.bgr2gray_u8(img)
    small = cv.resize_u8_linear(gray, 64, 64).astype(np.float64)
    n = np.arange(64)
    Wm = np.exp(-2j * np.pi * np.outer(n, n) / 64)
    Fq = Wm @ small @ Wm
    loga = np.log(np.abs(Fq))
    resid = np.exp(loga - ndi.uniform_filter(loga, 3, mode='mirror'))
    G = np.conj(Wm) @ (resid * np.exp(1j * np.angle(Fq))) @ np.conj(Wm)        # unscaled inverse
    k = np.exp(-0.5 * (np.arange(5) - 2) ** 2 / 64.0)
    m = ndi.correlate1d(ndi.correlate1d(np.abs(G), k / k.sum(), axis=1, mode='mirror'), k / k.sum(), axis=0, mode='mirror') ** 2
    m = m / m.max()
    mine = np.stack([cv.resize_profile_f32(r, 112) for r in m.astype(np.float32)])
    mine = np.stack([cv.resize_profile_f32(mine[:, c], 80) for c in range(112)], 1)
    assert np.abs(mine - ref_map).max() <= 1e-4
