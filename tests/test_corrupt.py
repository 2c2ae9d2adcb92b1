"""CPU: oadg_amd/pipelines/corrupt.py - the on-the-fly ``Corrupt`` transform of the robustness benchmark
(mmdet/datasets/pipelines/transforms.py:1277-1317 -> third-party ``imagecorruptions``, absent: restated).  Closed-form cases,
library cross-checks (Pillow, scipy) and the statistics of the noise models; severity tables as published (ImageNet-C)."""
import io
import os

import numpy as np
import pytest

import oadg_amd  # noqa: F401
from oadg_amd.pipelines import corrupt as C
from oadg_amd.pipelines.corrupt import Corrupt, corrupt
from oadg_amd.registry import PIPELINES, build_from_cfg


def _img(seed=0, h=96, w=160):
    rs = np.random.RandomState(seed)
    base = rs.randint(0, 256, (h // 8, w // 8, 3)).astype(np.float64)
    img = np.kron(base, np.ones((8, 8, 1)))                       # blocky, then smoothed a little
    img = (img + np.roll(img, 3, 0) + np.roll(img, 5, 1)) / 3
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.fixture
def frost_dir(tmp_path, monkeypatch):
    """stand-ins for the package's six frost photographs (data the repository does not hold): icy-looking noise"""
    from PIL import Image
    rs = np.random.RandomState(3)
    for i, name in enumerate(C.FROST_FILES):
        a = np.clip(rs.normal(170, 40, (60 + 7 * i, 90 + 5 * i, 1)) + rs.normal(0, 10, (60 + 7 * i, 90 + 5 * i, 3)), 0, 255)
        Image.fromarray(a.astype(np.uint8)).save(tmp_path / name)
    monkeypatch.setenv('OADG_FROST_DIR', str(tmp_path))
    return tmp_path


@pytest.mark.parametrize('name', C.IMPLEMENTED)
@pytest.mark.parametrize('severity', [1, 3, 5])
def test_every_implemented_corruption_returns_a_changed_uint8_image(name, severity, frost_dir):
    img = _img(1)
    np.random.seed(7)
    out = corrupt(img, name, severity)
    assert out.dtype == np.uint8 and out.shape == img.shape
    assert (out != img).any()
    np.random.seed(7)
    assert np.array_equal(out, corrupt(img, name, severity))          # numpy's global stream is the only randomness
    assert np.array_equal(corrupt(img, name, 0), img)


def test_contrast_brightness_saturate_closed_forms():
    img = _img(2)
    x = img / 255.
    for sev, c in zip(range(1, 6), [0.4, .3, .2, .1, .05]):
        m = x.mean(axis=(0, 1), keepdims=True)
        assert np.array_equal(corrupt(img, 'contrast', sev), np.uint8(np.clip((x - m) * c + m, 0, 1) * 255))
    grey = np.repeat(_img(3)[..., :1], 3, axis=2)                          # s = 0: HSV value = the grey level
    for sev, c in zip(range(1, 6), [.1, .2, .3, .4, .5]):
        exp = np.uint8(np.clip(np.clip(grey / 255. + c, 0, 1), 0, 1) * 255)
        assert np.abs(corrupt(grey, 'brightness', sev).astype(int) - exp.astype(int)).max() <= 1
        assert np.abs(corrupt(grey, 'saturate', sev).astype(int) - grey.astype(int)).max() <= \
            (1 if sev < 4 else 255)                                           # c1 > 0 adds saturation to greys at 4, 5
    hsv = C._rgb2hsv(x)
    assert np.abs(C._hsv2rgb(hsv) - x).max() <= 1e-12                      # the colour-space pair is an exact round trip
    assert np.abs(corrupt(img, 'saturate', 1).astype(int).std() - img.astype(int).std()) > 0


def test_pixelate_and_jpeg_are_pillow_round_trips():
    from PIL import Image
    img = _img(4, 90, 150)[:-1, :-3]                            # 87 x 141: sides that the scale factors do not divide
    h, w = img.shape[:2]
    for sev, c in zip(range(1, 6), [0.6, 0.5, 0.4, 0.3, 0.25]):
        im = Image.fromarray(img).resize((int(w * c), int(h * c)), Image.BOX).resize((w, h), Image.NEAREST)   # BOX down, NEAREST up
        assert len(np.unique(np.asarray(im).reshape(-1, 3), axis=0)) <= int(w * c) * int(h * c)       # blocks of constant colour
        assert np.array_equal(corrupt(img, 'pixelate', sev), np.asarray(im))
    for sev, q in zip(range(1, 6), [25, 18, 15, 10, 7]):
        buf = io.BytesIO()
        Image.fromarray(img).save(buf, 'JPEG', quality=q)
        assert np.array_equal(corrupt(img, 'jpeg_compression', sev), np.asarray(Image.open(buf)))
    errs = [np.abs(corrupt(img, 'jpeg_compression', s).astype(int) - img.astype(int)).mean() for s in (1, 5)]
    assert errs[1] > errs[0]


def test_blurs_against_scipy_and_monotone_in_severity():
    from scipy.ndimage import gaussian_filter
    img = _img(5)
    for sev, sig in zip(range(1, 6), [1, 2, 3, 4, 6]):
        exp = np.uint8(np.clip(gaussian_filter(img / 255., sigma=[sig, sig, 0], mode='nearest', truncate=4.0), 0, 1) * 255)
        assert np.array_equal(corrupt(img, 'gaussian_blur', sev), exp)

    def sharp(a):
        a = a.astype(float)
        return np.abs(np.diff(a, axis=0)).mean() + np.abs(np.diff(a, axis=1)).mean()
    for name in ('gaussian_blur', 'defocus_blur', 'zoom_blur'):
        s = [sharp(corrupt(img, name, k)) for k in (1, 3, 5)]
        assert sharp(img) > s[0] > s[1] > s[2], (name, s)
    k = C._disk(3, 0.1)
    assert k.shape == (17, 17) and abs(k.sum() - 1) < 1e-5 and k[8, 8] > 0 and k[0, 0] == 0
    flat = np.full((40, 60, 3), 77, np.uint8)                                 # a constant image is a fixed point of every blur
    for name in ('gaussian_blur', 'defocus_blur', 'zoom_blur'):
        assert np.abs(corrupt(flat, name, 3).astype(int) - 77).max() <= 1


def test_noise_models_statistics():
    flat = np.full((256, 256, 3), 128, np.uint8)
    x0 = 128 / 255.
    np.random.seed(0)
    for sev, c in zip(range(1, 6), [0.08, 0.12, 0.18, 0.26, 0.38]):
        d = corrupt(flat, 'gaussian_noise', sev) / 255. - x0
        if sev <= 3:                                                           # (clipping bites at the large scales)
            assert abs(d.std() - c) < 0.02 * c + 0.004 and abs(d.mean()) < 0.01
    for sev, c in zip(range(1, 6), [.15, .2, 0.35, 0.45, 0.6]):
        d = corrupt(flat, 'speckle_noise', sev) / 255. - x0
        if sev <= 3:
            assert abs(d.std() - c * x0) < 0.03 * c * x0 + 0.004
    for sev, c in zip(range(1, 6), [60, 25, 12, 5, 3]):
        v = corrupt(flat, 'shot_noise', sev) / 255.
        assert abs(v.mean() - x0) < 0.02 and abs(v.var() - x0 / c) < 0.15 * x0 / c + 1e-3 or sev > 3
    for sev, c in zip(range(1, 6), [.03, .06, .09, 0.17, 0.27]):
        out = corrupt(flat, 'impulse_noise', sev)
        frac = (out != 128).mean()
        assert abs(frac - c) < 0.01 and abs((out == 255).sum() - (out == 0).sum()) < 0.05 * (out != 128).sum() + 50
    np.random.seed(1)
    f = C._plasma_fractal(64, 2)
    assert f.shape == (64, 64) and f.min() == 0 and f.max() == 1
    img = _img(6, 64, 96)
    foggy = corrupt(img, 'fog', 3)
    assert foggy.astype(float).std() < img.astype(float).std()                  # fog lowers contrast


def test_corrupt_transform_registry_dict_api_and_unavailable_corruptions():
    t = build_from_cfg(dict(type='Corrupt', corruption='contrast', severity=2), PIPELINES)
    assert isinstance(t, Corrupt) and 'contrast' in repr(t)
    img = _img(8)
    res = t(dict(img=img.copy(), img_fields=['img']))
    assert np.array_equal(res['img'], corrupt(img, 'contrast', 2))
    os.environ.pop('OADG_FROST_DIR', None)
    for name in C.NEEDS_ASSETS:                           # frost without the package's photographs: refused by name
        with pytest.raises(NotImplementedError, match='OADG_FROST_DIR'):
            corrupt(img, name, 1)
    with pytest.raises(ValueError):
        corrupt(img, 'no_such_corruption', 1)
    with pytest.raises(TypeError):
        corrupt(img.astype(np.float32), 'contrast', 1)
    from oadg_amd import evaluation as E                  # every benchmark name is either implemented or refused by name
    assert set(E.CORRUPTION_SETS['all']) == set(C.IMPLEMENTED) and len(C.IMPLEMENTED) == 19


# ---------------------------------------------------------------------------------------------- round 6: the other six
def test_motion_blur_kernel_constant_image_and_impulse():
    for radius, sigma in [(10, 3), (15, 12), (20, 15)]:
        k = C._motion_kernel(radius, sigma)
        assert k.shape == (2 * radius + 1,) and abs(k.sum() - 1) < 1e-12 and (np.diff(k) < 0).all()
    flat = np.full((40, 56, 3), 93, np.uint8)
    np.random.seed(0)
    assert np.array_equal(corrupt(flat, 'motion_blur', 4), flat)              # weights sum to 1
    # an impulse smears along ONE line through it, towards the side the angle names, with the kernel's weights
    x = np.zeros((41, 41), np.float32)
    x[20, 20] = 1
    out = C._motion_blur(x, radius=5, sigma=3, angle=0.0)                      # angle 0: along +x (dx = -i shifts left ...)
    k = C._motion_kernel(5, 3)
    assert abs(out.sum() - 1) < 1e-5 and np.count_nonzero(out) == 11 and np.count_nonzero(out[20]) == 11
    assert np.allclose(np.sort(out[20][out[20] > 0])[::-1], k.astype(np.float32), atol=1e-6)
    out = C._motion_blur(x, radius=5, sigma=3, angle=90.0)
    assert np.count_nonzero(out[:, 20]) == 11 and np.count_nonzero(out) == 11
    # shift fills the vacated band from the adjacent column / row instead of wrapping
    a = np.arange(20, dtype=np.float32).reshape(4, 5)
    assert np.array_equal(C._shift(a, 2, 0)[:, :3], np.stack([a[:, 0]] * 3, 1)) and np.array_equal(C._shift(a, 2, 0)[:, 2:], a[:, :3])
    assert np.array_equal(C._shift(a, 0, -1)[:3], a[1:]) and np.array_equal(C._shift(a, 0, -1)[3], a[3])


def test_glass_shuffle_equals_the_sequential_python_loop_and_keeps_every_pixel():
    import ctypes
    from oadg_amd import _lib
    rs = np.random.RandomState(5)
    for (h, w, delta, iters) in [(13, 17, 1, 2), (16, 12, 3, 2), (9, 30, 4, 1)]:
        img = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
        d = rs.randint(-delta, delta, size=(iters, h - 2 * delta, w - 2 * delta, 2)).astype(np.int32)
        ref = img.copy()
        it = iter(d.reshape(-1, 2))
        for _ in range(iters):
            for y in range(h - delta, delta, -1):
                for x in range(w - delta, delta, -1):
                    dx, dy = next(it)
                    a, b = ref[y, x].copy(), ref[y + dy, x + dx].copy()
                    ref[y, x], ref[y + dy, x + dx] = b, a
        got = img.copy()
        assert _lib.lib().oadg_glass_shuffle_u8(got.ctypes.data, h, w, 3, delta, iters, d.ctypes.data) == 0
        assert np.array_equal(got, ref)
        assert np.array_equal(np.sort(got.reshape(-1, 3).view('u1,u1,u1'), 0), np.sort(img.reshape(-1, 3).view('u1,u1,u1'), 0))
    bad = np.full((1, 3, 3, 2), 7, np.int32)                               # an offset outside [-delta, delta)
    assert _lib.lib().oadg_glass_shuffle_u8(img.ctypes.data, 5, 5, 3, 1, 1, bad.ctypes.data) == -1
    flat = np.full((24, 24, 3), 77, np.uint8)
    np.random.seed(3)
    assert np.abs(corrupt(flat, 'glass_blur', 2).astype(int) - 77).max() <= 1         # nothing to shuffle, nothing to blur


def test_chamfer_transform_equals_a_brute_force_5x5_chamfer():
    from oadg_amd import _lib
    rs = np.random.RandomState(2)
    h, w = 23, 31
    src = (rs.rand(h, w) > 0.06).astype(np.uint8) * 255
    dist = np.empty((h, w), np.float32)
    assert _lib.lib().oadg_chamfer_l2_5x5(src.ctypes.data, h, w, dist.ctypes.data) == 0
    # the chamfer metric = shortest path over axial (1), diagonal (1.4) and knight (2.1969) steps INSIDE the image:
    # Bellman-Ford to convergence
    steps = [(0, 1, 65536), (1, 0, 65536), (1, 1, 91750), (1, -1, 91750), (1, 2, 143976), (2, 1, 143976), (1, -2, 143976),
             (2, -1, 143976)]
    steps = steps + [(-a, -b, c) for a, b, c in steps]
    d = np.where(src == 0, 0, 1 << 40).astype(np.int64)
    for _ in range(h + w):
        nd = d.copy()
        for dy, dx, c in steps:
            sh = np.full_like(d, 1 << 40)
            ys, yd = (slice(dy, None), slice(None, h - dy)) if dy >= 0 else (slice(None, h + dy), slice(-dy, None))
            xs, xd = (slice(dx, None), slice(None, w - dx)) if dx >= 0 else (slice(None, w + dx), slice(-dx, None))
            sh[yd, xd] = d[ys, xs] + c
            nd = np.minimum(nd, sh)
        if np.array_equal(nd, d):
            break
        d = nd
    # two raster passes reach the shortest path up to the usual chamfer slack (paths that zig-zag against the scan
    # order): never shorter, at most a few percent longer, and exact for most pixels
    ref = d / 65536.0
    assert (dist >= ref - 1e-4).all() and (dist <= ref * 1.05 + 1e-4).all() and (np.abs(dist - ref) < 1e-4).mean() > 0.9
    assert (dist[src == 0] == 0).all()
    # straight runs: k axial steps = k, k diagonal steps = 1.4 k
    one = np.full((9, 9), 255, np.uint8)
    one[4, 4] = 0
    out = np.empty((9, 9), np.float32)
    _lib.lib().oadg_chamfer_l2_5x5(one.ctypes.data, 9, 9, out.ctypes.data)
    assert np.allclose(out[4, 4:], np.arange(5)) and np.allclose(np.diag(out)[4:], 1.4 * np.arange(5), atol=1e-4)
    assert abs(out[5, 6] - 2.1969) < 1e-4


def test_canny_equalize_and_cubic_resize_restatements():
    # a vertical step edge: one column of edge pixels at the step, none elsewhere
    img = np.zeros((32, 40), np.uint8)
    img[:, 20:] = 200
    e = C._canny(img, 50, 150)
    cols = np.flatnonzero(e.any(0))
    assert set(np.unique(e)) == {0, 255} and len(cols) == 1 and cols[0] in (19, 20) and (e[:, cols[0]] == 255).all()
    # a weak edge (|gradient| between the thresholds) survives only where it touches a strong one
    img = np.zeros((32, 40), np.uint8)
    img[:16, 20:] = 200                                        # strong on the upper half
    img[16:, 20:] = 20                                         # 4 * 20 = 80: between 50 and 150
    e = C._canny(img, 50, 150)
    assert e[2:14].any() and e[18:30, 18:22].any()             # the weak half hangs on the strong half
    img[:16] = 0
    assert not C._canny(img, 50, 150)[20:30].any()             # alone it is dropped
    assert not C._canny(np.full((16, 16), 90, np.uint8), 50, 150).any()
    # equalizeHist: constant image unchanged; two levels -> 0 and 255
    assert np.array_equal(C._equalize_hist(np.full((4, 4), 9, np.uint8)), np.full((4, 4), 9, np.uint8))
    two = np.array([[3, 3, 8, 8]], np.uint8)
    assert np.array_equal(C._equalize_hist(two), np.array([[0, 0, 255, 255]], np.uint8))
    # cubic taps: partition of unity, interpolating at t = 0; identity resize; constant image
    t = np.linspace(0, 1, 11)
    wts = C._cubic_weights(t)
    assert np.allclose(wts.sum(-1), 1) and np.allclose(wts[0], [0, 1, 0, 0]) and np.allclose(wts[-1], [0, 0, 1, 0])
    assert abs(wts[5][0] + 0.09375) < 1e-12 and abs(wts[5][1] - 0.59375) < 1e-12       # a = -0.75 at t = 0.5
    a = _img(4, 24, 32)
    assert np.array_equal(C._resize_cubic(a, 24, 32), a)
    assert (C._resize_cubic(np.full((7, 9, 3), 50, np.uint8), 20, 31) == 50).all()


def test_snow_frost_elastic_spatter_properties(frost_dir):
    img = _img(9, 64, 96)
    np.random.seed(11)
    s = corrupt(img, 'snow', 3)
    assert s.astype(int).mean() > img.astype(int).mean()                       # snow whitens
    np.random.seed(11)
    dark = np.zeros((64, 96, 3), np.uint8)
    f = corrupt(dark, 'frost', 5)                                             # 0.6 * 0 + 0.75 * a crop of one photograph
    assert f.mean() > 60 and f.shape == dark.shape
    np.random.seed(12)
    flat = np.full((64, 96, 3), 120, np.uint8)
    assert np.abs(corrupt(flat, 'elastic_transform', 5).astype(int) - 120).max() <= 1     # a warp moves pixels, not levels
    np.random.seed(12)
    e = corrupt(img, 'elastic_transform', 3)
    assert 0 < np.abs(e.astype(int) - img.astype(int)).mean() < 40
    for sev, toward in [(2, (175, 238, 238)), (5, (63, 42, 20))]:             # water brightens, mud pulls towards brown
        np.random.seed(13)
        grey = np.full((96, 128, 3), 100, np.uint8)
        out = corrupt(grey, 'spatter', sev).astype(int)
        changed = (out != 100).any(-1)
        assert 0.005 < changed.mean() < 0.9
        d0 = np.abs(np.array(toward) - 100)
        assert (np.abs(out[changed] - np.array(toward)) <= d0 + 1).all()       # every changed pixel moved towards the colour
