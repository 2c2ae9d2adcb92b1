"""``Corrupt`` (mmdet/datasets/pipelines/transforms.py:1277-1317): on-the-fly image corruption for the robustness benchmark
(tools/analysis_tools/test_robustness.py:269-277, ``--load-dataset original``).

The reference delegates to the third-party package ``imagecorruptions`` (bethgelab, v1.1.2 = the ImageNet-C corruption
functions of Hendrycks & Dietterich adapted to arbitrary image sizes), which is neither vendored in the reference nor
installed here.  This file restates the PUBLISHED algorithms and severity tables of the corruptions that need nothing
beyond numpy / scipy / Pillow:

    benchmark: gaussian_noise  shot_noise  impulse_noise  defocus_blur  zoom_blur  fog  brightness  contrast  pixelate
               jpeg_compression                                                                      (10 of the 15)
    holdout:   speckle_noise  gaussian_blur  saturate                                                 (3 of the 4)

and stops with a message for the rest (glass_blur, motion_blur, snow, frost, elastic_transform, spatter: they need
OpenCV remaps, ImageMagick kernels or the package's frost image assets).  PARITY UNPINNED: no copy of the package exists to
compare with; tests/test_corrupt.py pins the closed-form cases (contrast, brightness on grey images, pixelate / jpeg against
Pillow, blur against scipy) and the statistics of the noise models.  Random draws come from numpy's global stream, as in
the package.  Evaluation-time host code (the reference corrupts numpy images inside its DataLoader workers too); it is
not part of the training hot path.
"""
from io import BytesIO

import numpy as np

from ..registry import PIPELINES

IMPLEMENTED = ('gaussian_noise', 'shot_noise', 'impulse_noise', 'defocus_blur', 'zoom_blur', 'fog', 'brightness',
               'contrast', 'pixelate', 'jpeg_compression', 'speckle_noise', 'gaussian_blur', 'saturate')
NOT_IMPLEMENTED = ('glass_blur', 'motion_blur', 'snow', 'frost', 'elastic_transform', 'spatter')


def _rgb2hsv(x):
    """skimage.color.rgb2hsv on float [H, W, 3] in [0, 1]"""
    v = x.max(-1)
    delta = np.ptp(x, -1)
    with np.errstate(invalid='ignore', divide='ignore'):
        s = np.where(delta == 0, 0.0, delta / v)
        s = np.where(v == 0, 0.0, s)
        r, g, b = x[..., 0], x[..., 1], x[..., 2]
        h = np.zeros_like(v)
        i = x[..., 0] == v
        h[i] = ((g - b) / delta)[i]
        i = x[..., 1] == v
        h[i] = (2.0 + (b - r) / delta)[i]
        i = x[..., 2] == v
        h[i] = (4.0 + (r - g) / delta)[i]
    h = (h / 6.0) % 1.0
    h[delta == 0] = 0.0
    out = np.stack([h, s, v], -1)
    out[np.isnan(out)] = 0
    return out


def _hsv2rgb(x):
    """skimage.color.hsv2rgb"""
    h, s, v = x[..., 0], x[..., 1], x[..., 2]
    hi = np.floor(h * 6)
    f = h * 6 - hi
    p = v * (1 - s)
    q = v * (1 - f * s)
    t = v * (1 - (1 - f) * s)
    hi = np.stack([hi, hi, hi], -1).astype(np.uint8) % 6
    return np.choose(hi, np.stack([np.stack((v, t, p), -1), np.stack((q, v, p), -1), np.stack((p, v, t), -1),
                                   np.stack((p, q, v), -1), np.stack((t, p, v), -1), np.stack((v, p, q), -1)]))


def gaussian_noise(x, severity=1):
    c = [0.08, 0.12, 0.18, 0.26, 0.38][severity - 1]
    x = np.array(x) / 255.
    return np.clip(x + np.random.normal(size=x.shape, scale=c), 0, 1) * 255


def shot_noise(x, severity=1):
    c = [60, 25, 12, 5, 3][severity - 1]
    x = np.array(x) / 255.
    return np.clip(np.random.poisson(x * c) / float(c), 0, 1) * 255


def impulse_noise(x, severity=1):
    """skimage.util.random_noise(mode='s&p', amount=c, salt_vs_pepper=0.5)"""
    c = [.03, .06, .09, 0.17, 0.27][severity - 1]
    out = np.array(x) / 255.
    flipped = np.random.choice([True, False], size=out.shape, p=[c, 1 - c])
    salted = np.random.choice([True, False], size=out.shape, p=[0.5, 0.5])
    out[flipped & salted] = 1
    out[flipped & ~salted] = 0
    return np.clip(out, 0, 1) * 255


def speckle_noise(x, severity=1):
    c = [.15, .2, 0.35, 0.45, 0.6][severity - 1]
    x = np.array(x) / 255.
    return np.clip(x + x * np.random.normal(size=x.shape, scale=c), 0, 1) * 255


def gaussian_blur(x, severity=1):
    """skimage.filters.gaussian(sigma=c, multichannel=True): scipy gaussian_filter, mode 'nearest', truncate 4"""
    from scipy.ndimage import gaussian_filter
    c = [1, 2, 3, 4, 6][severity - 1]
    x = gaussian_filter(np.array(x) / 255., sigma=[c, c, 0], mode='nearest', truncate=4.0)
    return np.clip(x, 0, 1) * 255


def _disk(radius, alias_blur=0.1, dtype=np.float32):
    from scipy.ndimage import correlate1d
    if radius <= 8:
        L, k = np.arange(-8, 8 + 1), 3
    else:
        L, k = np.arange(-radius, radius + 1), 5
    X, Y = np.meshgrid(L, L)
    aliased = np.array((X ** 2 + Y ** 2) <= radius ** 2, dtype=dtype)
    aliased /= np.sum(aliased)
    # cv2.GaussianBlur(ksize=(k, k), sigmaX=alias_blur): separable kernel exp(-(i - c)^2 / (2 sigma^2)) normalised, border
    # BORDER_REFLECT_101 (= scipy 'mirror')
    i = np.arange(k) - (k - 1) / 2.0
    g = np.exp(-(i * i) / (2.0 * alias_blur * alias_blur))
    g = (g / g.sum()).astype(np.float64)
    out = correlate1d(aliased.astype(np.float64), g, axis=0, mode='mirror')
    return correlate1d(out, g, axis=1, mode='mirror').astype(dtype)


def defocus_blur(x, severity=1):
    """cv2.filter2D of every channel with the anti-aliased disk (correlation, BORDER_REFLECT_101)"""
    from scipy.ndimage import correlate
    c = [(3, 0.1), (4, 0.5), (6, 0.5), (8, 0.5), (10, 0.5)][severity - 1]
    x = np.array(x) / 255.
    kernel = _disk(radius=c[0], alias_blur=c[1]).astype(np.float64)
    ch = [correlate(x[:, :, d], kernel, mode='mirror') for d in range(3)]
    return np.clip(np.stack(ch, -1), 0, 1) * 255


def _clipped_zoom(img, zoom_factor):
    """imagecorruptions.clipped_zoom for h x w images: centre crop of ceil(side / zoom), bilinear zoom back; the zoomed
    layer (a few pixels larger than the image) is trimmed from the TOP-LEFT, as zoom_blur of the package does
    (``zoom_layer[:h, :w]``), not around its centre"""
    from scipy.ndimage import zoom as scizoom
    h, w = img.shape[:2]
    ch, cw = int(np.ceil(h / float(zoom_factor))), int(np.ceil(w / float(zoom_factor)))
    top, left = (h - ch) // 2, (w - cw) // 2
    img = scizoom(img[top:top + ch, left:left + cw], (zoom_factor, zoom_factor, 1), order=1)
    return img[:h, :w]


def zoom_blur(x, severity=1):
    c = [np.arange(1, 1.11, 0.01), np.arange(1, 1.16, 0.01), np.arange(1, 1.21, 0.02), np.arange(1, 1.26, 0.02),
         np.arange(1, 1.33, 0.03)][severity - 1]
    x = (np.array(x) / 255.).astype(np.float32)
    out = np.zeros_like(x)
    for z in c:
        out += _clipped_zoom(x, z)
    x = (x + out) / (len(c) + 1)
    return np.clip(x, 0, 1) * 255


def _plasma_fractal(mapsize=256, wibbledecay=3):
    """diamond-square height map in [0, 1] (ImageNet-C plasma_fractal); mapsize a power of two"""
    maparray = np.empty((mapsize, mapsize), dtype=np.float64)
    maparray[0, 0] = 0
    stepsize, wibble = mapsize, 100.0

    def wibbledmean(array):
        return array / 4 + wibble * np.random.uniform(-wibble, wibble, array.shape)

    def fillsquares():
        cornerref = maparray[0:mapsize:stepsize, 0:mapsize:stepsize]
        squareaccum = cornerref + np.roll(cornerref, shift=-1, axis=0)
        squareaccum += np.roll(squareaccum, shift=-1, axis=1)
        maparray[stepsize // 2:mapsize:stepsize, stepsize // 2:mapsize:stepsize] = wibbledmean(squareaccum)

    def filldiamonds():
        ms = maparray.shape[0]
        drgrid = maparray[stepsize // 2:ms:stepsize, stepsize // 2:ms:stepsize]
        ulgrid = maparray[0:ms:stepsize, 0:ms:stepsize]
        ldrsum = drgrid + np.roll(drgrid, 1, axis=0)
        lulsum = ulgrid + np.roll(ulgrid, -1, axis=1)
        maparray[0:ms:stepsize, stepsize // 2:ms:stepsize] = wibbledmean(ldrsum + lulsum)
        tdrsum = drgrid + np.roll(drgrid, 1, axis=1)
        tulsum = ulgrid + np.roll(ulgrid, -1, axis=0)
        maparray[stepsize // 2:ms:stepsize, 0:ms:stepsize] = wibbledmean(tdrsum + tulsum)

    while stepsize >= 2:
        fillsquares()
        filldiamonds()
        stepsize //= 2
        wibble /= wibbledecay
    maparray -= maparray.min()
    return maparray / maparray.max()


def fog(x, severity=1):
    c = [(1.5, 2), (2., 2), (2.5, 1.7), (2.5, 1.5), (3., 1.4)][severity - 1]
    x = np.array(x) / 255.
    h, w = x.shape[:2]
    mapsize = int(2 ** np.ceil(np.log2(max(h, w))))
    max_val = x.max()
    x = x + c[0] * _plasma_fractal(mapsize=mapsize, wibbledecay=c[1])[:h, :w][..., np.newaxis]
    return np.clip(x * max_val / (max_val + c[0]), 0, 1) * 255


def brightness(x, severity=1):
    c = [.1, .2, .3, .4, .5][severity - 1]
    x = _rgb2hsv(np.array(x) / 255.)
    x[:, :, 2] = np.clip(x[:, :, 2] + c, 0, 1)
    return np.clip(_hsv2rgb(x), 0, 1) * 255


def saturate(x, severity=1):
    c = [(0.3, 0), (0.1, 0), (2, 0), (5, 0.1), (20, 0.2)][severity - 1]
    x = _rgb2hsv(np.array(x) / 255.)
    x[:, :, 1] = np.clip(x[:, :, 1] * c[0] + c[1], 0, 1)
    return np.clip(_hsv2rgb(x), 0, 1) * 255


def contrast(x, severity=1):
    c = [0.4, .3, .2, .1, .05][severity - 1]
    x = np.array(x) / 255.
    means = np.mean(x, axis=(0, 1), keepdims=True)
    return np.clip((x - means) * c + means, 0, 1) * 255


def jpeg_compression(x, severity=1):
    from PIL import Image
    c = [25, 18, 15, 10, 7][severity - 1]
    out = BytesIO()
    Image.fromarray(np.asarray(x, np.uint8)).save(out, 'JPEG', quality=c)
    return np.asarray(Image.open(out))


def pixelate(x, severity=1):
    from PIL import Image
    c = [0.6, 0.5, 0.4, 0.3, 0.25][severity - 1]
    h, w = np.asarray(x).shape[:2]
    im = Image.fromarray(np.asarray(x, np.uint8))
    im = im.resize((int(w * c), int(h * c)), Image.BOX)
    return np.asarray(im.resize((w, h), Image.NEAREST))      # the package: BOX down, NEAREST up


_FUNCS = dict(gaussian_noise=gaussian_noise, shot_noise=shot_noise, impulse_noise=impulse_noise, defocus_blur=defocus_blur,
              zoom_blur=zoom_blur, fog=fog, brightness=brightness, contrast=contrast, pixelate=pixelate,
              jpeg_compression=jpeg_compression, speckle_noise=speckle_noise, gaussian_blur=gaussian_blur, saturate=saturate)


def corrupt(image, corruption_name, severity=1):
    """imagecorruptions.corrupt: uint8 [H, W, 3] -> uint8 [H, W, 3]; severity 0 returns the image"""
    image = np.asarray(image)
    if image.ndim != 3 or image.shape[2] != 3 or image.dtype != np.uint8:
        raise TypeError('corrupt expects a uint8 HxWx3 image')
    if severity == 0:
        return image
    if not 1 <= severity <= 5:
        raise ValueError('severity must be 0 ... 5')
    if corruption_name in NOT_IMPLEMENTED:
        raise NotImplementedError(
            f"corruption '{corruption_name}' of the imagecorruptions package needs OpenCV / ImageMagick / image assets that "
            f"are not available here; implemented: {', '.join(IMPLEMENTED)} - for the others generate the -c tree once and "
            f"use --load-dataset corrupted")
    if corruption_name not in _FUNCS:
        raise ValueError(f'unknown corruption {corruption_name!r}')
    return np.uint8(_FUNCS[corruption_name](image, severity))


@PIPELINES.register_module()
class Corrupt:
    """transforms.py:1277-1317; accepts the results dict of the reference (numpy ``img``) and, for the device pipeline, a
    uint8 [N, H, W, 3] batch through :meth:`batch`."""

    def __init__(self, corruption, severity=1):
        self.corruption, self.severity = corruption, severity

    def __call__(self, results):
        if 'img_fields' in results:
            assert results['img_fields'] == ['img'], 'Only single img_fields is allowed'
        results['img'] = corrupt(results['img'].astype(np.uint8), corruption_name=self.corruption, severity=self.severity)
        return results

    def batch(self, imgs_u8):
        """a resident uint8 batch: corrupted on the host image by image (as the reference's workers do) and uploaded again"""
        import torch
        host = imgs_u8.cpu().numpy()
        out = np.stack([corrupt(im, self.corruption, self.severity) for im in host])
        return torch.from_numpy(out).to(imgs_u8.device)

    def __repr__(self):
        return f'{self.__class__.__name__}(corruption={self.corruption}, severity={self.severity})'
