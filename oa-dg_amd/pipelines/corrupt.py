"""``Corrupt`` (mmdet/datasets/pipelines/transforms.py:1277-1317): on-the-fly image corruption for the robustness benchmark
(tools/analysis_tools/test_robustness.py:269-277, ``--load-dataset original``).

The reference delegates to the third-party package ``imagecorruptions`` (bethgelab, v1.1.2 = the ImageNet-C corruption
functions of Hendrycks & Dietterich adapted to arbitrary image sizes), which is neither vendored in the reference nor
installed here.  This file restates the PUBLISHED algorithms and severity tables of all 19 names of
test_robustness.py:222-235 on numpy / scipy / Pillow:

    benchmark: gaussian_noise  shot_noise  impulse_noise  defocus_blur  glass_blur  motion_blur  zoom_blur  snow  frost
               fog  brightness  contrast  elastic_transform  pixelate  jpeg_compression
    holdout:   speckle_noise  gaussian_blur  spatter  saturate

The OpenCV leaves the package calls are restated here as well (OpenCV is not in the image): Canny (3 x 3 Sobel, L1
magnitude, the fixed-point direction test, hysteresis as connected components), distanceTransform(DIST_L2, 5) and the
sequential pixel shuffle of glass_blur as two host loops of the library (csrc/corrupt_host.hip), equalizeHist, box blur,
filter2D, bicubic resize (a = -0.75, in floating point: within 1 grey level of OpenCV's fixed-point path).  ``frost`` blends
one of the six photographs the package ships (frost1.png ... frost6.jpg); they are data, not code, and are not here: point
``OADG_FROST_DIR`` at the package's ``frost/`` directory, otherwise frost stops with a message.

PARITY UNPINNED: no copy of the package exists to compare with; tests/test_corrupt.py pins the closed-form cases
(contrast, brightness on grey images, pixelate / jpeg against Pillow, blur against scipy, motion blur of constant and
impulse images, the chamfer transform and the shuffle against plain-Python loops, Canny on synthetic edges) and the
statistics of the noise models.  Random draws come from numpy's global stream, as in the package - except glass_blur's
offsets, which the package draws from numba's private generator inside its compiled loop (no stream to replay).
Evaluation-time host code (the reference corrupts numpy images inside its DataLoader workers too); it is not part of the
training hot path.
"""
import math
import os
from io import BytesIO

import numpy as np

from ..registry import PIPELINES

IMPLEMENTED = ('gaussian_noise', 'shot_noise', 'impulse_noise', 'defocus_blur', 'glass_blur', 'motion_blur', 'zoom_blur',
               'snow', 'frost', 'fog', 'brightness', 'contrast', 'elastic_transform', 'pixelate', 'jpeg_compression',
               'speckle_noise', 'gaussian_blur', 'spatter', 'saturate')
NEEDS_ASSETS = ('frost',)          # runs when OADG_FROST_DIR holds the package's six frost photographs


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


def _clipped_zoom(img, zoom_factor, trim=True):
    """imagecorruptions.clipped_zoom for h x w images: centre crop of ceil(side / zoom), bilinear zoom back; the zoomed
    layer (a few pixels larger than the image) is trimmed from the TOP-LEFT, as zoom_blur of the package does
    (``zoom_layer[:h, :w]``), not around its centre.  snow trims only after its motion blur (trim=False)."""
    from scipy.ndimage import zoom as scizoom
    h, w = img.shape[:2]
    ch, cw = int(np.ceil(h / float(zoom_factor))), int(np.ceil(w / float(zoom_factor)))
    top, left = (h - ch) // 2, (w - cw) // 2
    img = scizoom(img[top:top + ch, left:left + cw], (zoom_factor, zoom_factor, 1), order=1)
    return img[:h, :w] if trim else img


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


# ---------------------------------------------------------------------------------------------- round 6: the other six
def _gaussian(x, sigma, mode='nearest', truncate=4.0):
    """skimage.filters.gaussian on [H, W] or [H, W, C] (no blur across channels)"""
    from scipy.ndimage import gaussian_filter
    sig = list(np.broadcast_to(np.asarray(sigma, np.float64), (2,))) + [0] * (x.ndim - 2)
    return gaussian_filter(x, sigma=sig, mode=mode, truncate=truncate)


def glass_blur(x, severity=1):
    """blur, local shuffle (every pixel from the bottom-right corner to the top-left swaps with a neighbour at a drawn
    offset in [-delta, delta), `iterations` sweeps - sequential, csrc/corrupt_host.hip), blur"""
    from .. import _lib
    sigma, delta, iters = [(0.7, 1, 2), (0.9, 2, 1), (1, 2, 3), (1.1, 3, 2), (1.5, 4, 2)][severity - 1]
    x = np.ascontiguousarray(np.uint8(_gaussian(np.array(x) / 255., sigma) * 255))
    h, w = x.shape[:2]
    if h > 2 * delta and w > 2 * delta:
        d = np.ascontiguousarray(np.random.randint(-delta, delta, size=(iters, h - 2 * delta, w - 2 * delta, 2)), np.int32)
        _lib.check(_lib.lib().oadg_glass_shuffle_u8(x.ctypes.data, h, w, x.shape[2], delta, iters, d.ctypes.data),
                   'oadg_glass_shuffle_u8')
    return np.clip(_gaussian(x / 255., sigma), 0, 1) * 255


def _shift(image, dx, dy):
    """the package's ``shift``: np.roll with the vacated columns / rows filled from the adjacent one"""
    if dx < 0:
        shifted = np.roll(image, shift=image.shape[1] + dx, axis=1)
        shifted[:, dx:] = shifted[:, dx - 1:dx]
    elif dx > 0:
        shifted = np.roll(image, shift=dx, axis=1)
        shifted[:, :dx] = shifted[:, dx:dx + 1]
    else:
        shifted = image
    if dy < 0:
        shifted = np.roll(shifted, shift=image.shape[0] + dy, axis=0)
        shifted[dy:, :] = shifted[dy - 1:dy, :]
    elif dy > 0:
        shifted = np.roll(shifted, shift=dy, axis=0)
        shifted[:dy, :] = shifted[dy:dy + 1, :]
    return shifted


def _motion_kernel(radius, sigma):
    """half of a Gaussian over 2 radius + 1 steps along the motion, normalised (the package's replacement for
    ImageMagick's motion_blur)"""
    k = np.exp(-np.arange(radius * 2 + 1) ** 2 / (2.0 * sigma ** 2)) / (np.sqrt(2 * np.pi) * sigma)
    return k / np.sum(k)


def _motion_blur(x, radius, sigma, angle):
    kernel = _motion_kernel(radius, sigma)
    width = kernel.shape[0]
    point = (width * np.sin(np.deg2rad(angle)), width * np.cos(np.deg2rad(angle)))
    hypot = math.hypot(point[0], point[1])
    blurred = np.zeros_like(x, dtype=np.float32)
    for i in range(width):
        dy = -math.ceil(((i * point[0]) / hypot) - 0.5)
        dx = -math.ceil(((i * point[1]) / hypot) - 0.5)
        if abs(dy) >= x.shape[0] or abs(dx) >= x.shape[1]:
            break                       # the simulated motion left the image
        blurred = blurred + kernel[i] * _shift(x, dx, dy)
    return blurred


def motion_blur(x, severity=1):
    c = [(10, 3), (15, 5), (15, 8), (15, 12), (20, 15)][severity - 1]
    x = np.array(x)
    angle = np.random.uniform(-45, 45)
    return np.clip(_motion_blur(x, radius=c[0], sigma=c[1], angle=angle), 0, 255)


def _gray(x):
    """cv2.cvtColor(x, COLOR_RGB2GRAY) on float images"""
    return (0.299 * x[..., 0] + 0.587 * x[..., 1] + 0.114 * x[..., 2]).astype(x.dtype)


def snow(x, severity=1):
    c = [(0.1, 0.3, 3, 0.5, 10, 4, 0.8), (0.2, 0.3, 2, 0.5, 12, 4, 0.7), (0.55, 0.3, 4, 0.9, 12, 8, 0.7),
         (0.55, 0.3, 4.5, 0.85, 12, 8, 0.65), (0.55, 0.3, 2.5, 0.85, 12, 12, 0.55)][severity - 1]
    x = np.array(x, dtype=np.float32) / 255.
    h, w = x.shape[:2]
    layer = np.random.normal(size=(h, w), loc=c[0], scale=c[1])
    layer = _clipped_zoom(layer[..., np.newaxis], c[2], trim=False)
    layer[layer < c[3]] = 0
    layer = np.clip(layer.squeeze(-1), 0, 1)
    layer = _motion_blur(layer, radius=c[4], sigma=c[5], angle=np.random.uniform(-135, -45))
    layer = (np.round(layer * 255).astype(np.uint8) / 255.)[..., np.newaxis][:h, :w, :]
    x = c[6] * x + (1 - c[6]) * np.maximum(x, _gray(x).reshape(h, w, 1) * 1.5 + 0.5)
    return np.clip(x + layer + np.rot90(layer, k=2), 0, 1) * 255


def _cubic_weights(t, a=-0.75):
    """the four taps of cv2's INTER_CUBIC at fractional offset t"""
    w0 = ((a * (t + 1) - 5 * a) * (t + 1) + 8 * a) * (t + 1) - 4 * a
    w1 = ((a + 2) * t - (a + 3)) * t * t + 1
    w2 = ((a + 2) * (1 - t) - (a + 3)) * (1 - t) * (1 - t) + 1
    return np.stack([w0, w1, w2, 1.0 - w0 - w1 - w2], -1)


def _resize_cubic(img, new_h, new_w):
    """cv2.resize(img, (new_w, new_h), INTER_CUBIC) of a uint8 [H, W, C] image, separable, replicated borders"""
    out = img.astype(np.float64)
    for axis, n_new in ((0, new_h), (1, new_w)):
        n = out.shape[axis]
        f = (np.arange(n_new) + 0.5) * (n / float(n_new)) - 0.5
        i0 = np.floor(f).astype(np.int64)
        wts = _cubic_weights(f - i0)
        acc = 0
        for k in range(4):
            idx = np.clip(i0 - 1 + k, 0, n - 1)
            shape = [1, 1, 1]
            shape[axis] = n_new
            acc = acc + np.take(out, idx, axis=axis) * wts[:, k].reshape(shape)
        out = acc
    return np.uint8(np.clip(np.rint(out), 0, 255))


FROST_FILES = ('frost1.png', 'frost2.png', 'frost3.png', 'frost4.jpg', 'frost5.jpg', 'frost6.jpg')


def frost(x, severity=1):
    from PIL import Image
    c = [(1, 0.4), (0.8, 0.6), (0.7, 0.7), (0.65, 0.7), (0.6, 0.75)][severity - 1]
    root = os.environ.get('OADG_FROST_DIR')
    if not root or not os.path.isdir(root):
        raise NotImplementedError(
            "corruption 'frost' blends one of the six photographs the imagecorruptions package ships "
            f"({', '.join(FROST_FILES)}); they are not part of this repository: set OADG_FROST_DIR to the package's "
            "frost/ directory, or generate the -c tree once and use --load-dataset corrupted")
    idx = np.random.randint(5)                            # (the package's draw: the sixth file is never used)
    fr = np.asarray(Image.open(os.path.join(root, FROST_FILES[idx])).convert('RGB'))[..., ::-1]     # cv2.imread: BGR
    xs = np.array(x).shape
    if fr.shape[0] >= xs[0] and fr.shape[1] >= xs[1]:
        scale = 1
    else:
        scale = np.maximum(xs[0] / fr.shape[0], xs[1] / fr.shape[1]) * 1.1
    fr = _resize_cubic(fr, int(np.ceil(fr.shape[0] * scale)), int(np.ceil(fr.shape[1] * scale)))
    y0, x0 = np.random.randint(0, fr.shape[0] - xs[0]), np.random.randint(0, fr.shape[1] - xs[1])
    fr = fr[y0:y0 + xs[0], x0:x0 + xs[1]][..., [2, 1, 0]]
    return np.clip(c[0] * np.array(x) + c[1] * fr, 0, 255)


def elastic_transform(x, severity=1):
    """smoothed uniform displacement fields, bilinear resampling with reflected borders (the package's arbitrary-size form:
    no affine part)"""
    from scipy.ndimage import map_coordinates
    img = np.array(x, dtype=np.float32) / 255.
    h, w = img.shape[:2]
    sigma = np.array((h, w)) * 0.01
    alpha = [250 * 0.05, 250 * 0.065, 250 * 0.085, 250 * 0.1, 250 * 0.12][severity - 1]
    max_d = h * 0.005
    dx = (_gaussian(np.random.uniform(-max_d, max_d, size=(h, w)), sigma, mode='reflect', truncate=3) * alpha).astype(np.float32)
    dy = (_gaussian(np.random.uniform(-max_d, max_d, size=(h, w)), sigma, mode='reflect', truncate=3) * alpha).astype(np.float32)
    xx, yy = np.meshgrid(np.arange(w), np.arange(h))
    coords = np.stack([yy + dy, xx + dx])
    out = np.stack([map_coordinates(img[..., d], coords, order=1, mode='reflect') for d in range(img.shape[2])], -1)
    return np.clip(out, 0, 1) * 255


def _canny(img, low, high):
    """cv2.Canny(img, low, high) of a uint8 [H, W] image: 3 x 3 Sobel (replicated border), |dx| + |dy|, non-maximum
    suppression with the fixed-point tan(22.5 deg) test, hysteresis = 8-connected components of the candidates that hold
    a pixel above `high`"""
    from scipy.ndimage import correlate1d, label
    a = img.astype(np.int32)
    sm = np.array([1, 2, 1]); df = np.array([-1, 0, 1])
    gx = correlate1d(correlate1d(a, df, axis=1, mode='nearest'), sm, axis=0, mode='nearest')
    gy = correlate1d(correlate1d(a, df, axis=0, mode='nearest'), sm, axis=1, mode='nearest')
    mag = np.abs(gx) + np.abs(gy)
    m = np.pad(mag, 1)
    h, w = mag.shape
    c = m[1:-1, 1:-1]
    ax, ay = np.abs(gx).astype(np.int64), np.abs(gy).astype(np.int64) << 15
    tg22 = ax * 13573                                       # round(tan(22.5 deg) * 2^15)
    tg67 = tg22 + (ax << 16)
    horiz = ay < tg22
    vert = ~horiz & (ay > tg67)
    diag = ~horiz & ~vert
    s = np.where((gx ^ gy) < 0, -1, 1)
    yy, xx = np.mgrid[1:h + 1, 1:w + 1]
    keep = np.zeros((h, w), bool)
    keep |= horiz & (c > m[1:-1, :-2]) & (c >= m[1:-1, 2:])
    keep |= vert & (c > m[:-2, 1:-1]) & (c >= m[2:, 1:-1])
    keep |= diag & (c > m[yy - 1, xx - s]) & (c > m[yy + 1, xx + s])
    cand = keep & (mag > low)
    strong = cand & (mag > high)
    lab, n = label(cand, structure=np.ones((3, 3), int))
    good = np.zeros(n + 1, bool)
    good[np.unique(lab[strong])] = True
    good[0] = False
    return np.where(good[lab], 255, 0).astype(np.uint8)


def _equalize_hist(img):
    """cv2.equalizeHist"""
    hist = np.bincount(img.ravel(), minlength=256)
    i = int(np.flatnonzero(hist)[0])
    total = img.size
    if hist[i] == total:
        return np.full_like(img, i)
    scale = np.float32(255.0) / np.float32(total - hist[i])
    cum = np.cumsum(hist[i + 1:]).astype(np.float32) * scale
    lut = np.zeros(256, np.uint8)
    lut[i + 1:] = np.clip(np.rint(cum), 0, 255).astype(np.uint8)
    return lut[img]


def spatter(x, severity=1):
    from scipy.ndimage import correlate, uniform_filter
    from .. import _lib
    c = [(0.65, 0.3, 4, 0.69, 0.6, 0), (0.65, 0.3, 3, 0.68, 0.6, 0), (0.65, 0.3, 2, 0.68, 0.5, 0),
         (0.65, 0.3, 1, 0.65, 1.5, 1), (0.67, 0.4, 1, 0.65, 1.5, 1)][severity - 1]
    x = np.array(x, dtype=np.float32) / 255.
    h, w = x.shape[:2]
    liquid = _gaussian(np.random.normal(size=(h, w), loc=c[0], scale=c[1]), sigma=c[2])
    liquid[liquid < c[3]] = 0
    if c[5] == 0:                                           # water: droplets shaded along their rims
        liquid = (liquid * 255).astype(np.uint8)
        edges = np.ascontiguousarray(255 - _canny(liquid, 50, 150))
        dist = np.empty((h, w), np.float32)
        _lib.check(_lib.lib().oadg_chamfer_l2_5x5(edges.ctypes.data, h, w, dist.ctypes.data), 'oadg_chamfer_l2_5x5')
        dist = np.minimum(dist, 20)                          # threshold(20, THRESH_TRUNC)
        dist = uniform_filter(dist, size=3, mode='mirror').astype(np.uint8)           # blur 3 x 3, BORDER_REFLECT_101
        dist = _equalize_hist(dist)
        ker = np.array([[-2, -1, 0], [-1, 1, 1], [0, 1, 2]])
        dist = np.clip(correlate(dist.astype(np.int32), ker, mode='mirror'), 0, 255)   # filter2D to CV_8U
        dist = np.rint(uniform_filter(dist.astype(np.float64), size=3, mode='mirror')).astype(np.float32)
        m = liquid.astype(np.float32) * dist
        peak = m.max()
        m = (m / peak if peak > 0 else m) * c[4]
        color = np.array([175 / 255., 238 / 255., 238 / 255.], np.float32)     # pale turquoise
        return np.clip(x + m[..., np.newaxis] * color, 0, 1) * 255
    m = np.where(liquid > c[3], 1, 0)                       # mud
    m = _gaussian(m.astype(np.float32), sigma=c[4])
    m[m < 0.8] = 0
    color = np.array([63 / 255., 42 / 255., 20 / 255.], np.float32) * m[..., np.newaxis]
    x = x * (1 - m[..., np.newaxis])
    return np.clip(x + color, 0, 1) * 255


_FUNCS = dict(gaussian_noise=gaussian_noise, shot_noise=shot_noise, impulse_noise=impulse_noise, defocus_blur=defocus_blur,
              zoom_blur=zoom_blur, fog=fog, brightness=brightness, contrast=contrast, pixelate=pixelate,
              jpeg_compression=jpeg_compression, speckle_noise=speckle_noise, gaussian_blur=gaussian_blur, saturate=saturate,
              glass_blur=glass_blur, motion_blur=motion_blur, snow=snow, frost=frost, elastic_transform=elastic_transform,
              spatter=spatter)


def corrupt(image, corruption_name, severity=1):
    """imagecorruptions.corrupt: uint8 [H, W, 3] -> uint8 [H, W, 3]; severity 0 returns the image"""
    image = np.asarray(image)
    if image.ndim != 3 or image.shape[2] != 3 or image.dtype != np.uint8:
        raise TypeError('corrupt expects a uint8 HxWx3 image')
    if severity == 0:
        return image
    if not 1 <= severity <= 5:
        raise ValueError('severity must be 0 ... 5')
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
