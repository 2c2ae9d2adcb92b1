"""CPU: oadg_amd/pipelines/corrupt.py - the on-the-fly ``Corrupt`` transform of the robustness benchmark
(mmdet/datasets/pipelines/transforms.py:1277-1317 -> third-party ``imagecorruptions``, absent: restated).  Closed-form cases,
library cross-checks (Pillow, scipy) and the statistics of the noise models; severity tables as published (ImageNet-C)."""
import io

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


@pytest.mark.parametrize('name', C.IMPLEMENTED)
@pytest.mark.parametrize('severity', [1, 3, 5])
def test_every_implemented_corruption_returns_a_changed_uint8_image(name, severity):
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
    for name in C.NOT_IMPLEMENTED:
        with pytest.raises(NotImplementedError):
            corrupt(img, name, 1)
    with pytest.raises(ValueError):
        corrupt(img, 'no_such_corruption', 1)
    with pytest.raises(TypeError):
        corrupt(img.astype(np.float32), 'contrast', 1)
    from oadg_amd import evaluation as E                  # every benchmark name is either implemented or refused by name
    assert set(E.CORRUPTION_SETS['all']) == set(C.IMPLEMENTED) | set(C.NOT_IMPLEMENTED)
