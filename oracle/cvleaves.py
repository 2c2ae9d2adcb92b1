"""CPU restatement of the OpenCV functions OA-Mix calls.  Test infrastructure only (see oracle/__init__).

PARITY UNPINNED: opencv-python / opencv-contrib-python are unpinned in the reference (README.md:74-75), are not
vendored and are not installed here.  Restated from OpenCV 4.x's documented CPU semantics:

  cv2.warpAffine        imgproc/imgwarp.cpp  WarpAffineInvoker + remapBilinear<uchar>: 10-bit fixed-point
                        coordinates, 5-bit sub-pixel phase, 15-bit integer bilinear weights, BORDER_CONSTANT 0
  cv2.getRotationMatrix2D
  cv2.GaussianBlur      ksize=(0,0): k = round(sigma*8+1)|1 for CV_32F, float kernel normalised in float,
                        BORDER_REFLECT_101
  cv2.resize            INTER_LINEAR, half-pixel centres, edge clamp
  cv2.saliency.StaticSaliencySpectralResidual  (opencv-contrib saliency/src/staticSaliencySpectralResidual.cpp)

call sites: mmdet/datasets/pipelines/augmix.py:91-181, oa_mix.py:90-91,108-110,273-276.

One deliberate definition: OA-Mix only ever blurs/resizes axis-aligned box indicators.  Such a mask is an outer
product, and this oracle DEFINES its blurred / resized image as the float32 product of the two 1-D blurred /
resized profiles, fl(My[y] * Mx[x]) (a two-pass float32 filter differs from it by <= 2 ulp).  The HIP kernels
evaluate exactly this product, so no H x W x 3 float mask is ever materialised on the device.
"""
import math

import numpy as np

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS
AB_BITS = 10
AB_SCALE = 1 << AB_BITS
REMAP_COEF_BITS = 15


# ------------------------------------------------------------------------------------------- warpAffine
def invert_affine(M):
    """cv::warpAffine without WARP_INVERSE_MAP inverts the (double) matrix like this."""
    M = np.array(M, dtype=np.float64).reshape(2, 3).copy()
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[1, 1] * D, M[0, 0] * D
    M[0, 0] = A11
    M[0, 1] *= -D
    M[1, 0] *= -D
    M[1, 1] = A22
    b1 = -M[0, 0] * M[0, 2] - M[0, 1] * M[1, 2]
    b2 = -M[1, 0] * M[0, 2] - M[1, 1] * M[1, 2]
    M[0, 2], M[1, 2] = b1, b2
    return M


def _sat_int(x):
    """saturate_cast<int>(double) = round half to even, clamped to int32."""
    return np.clip(np.rint(x), -2147483648.0, 2147483647.0).astype(np.int64)


def warp_coords(Minv, width, height, x0=0, y0=0, w=None, h=None):
    """Fixed-point source coordinates (in 1/32 px) for the dst rectangle [x0,x0+w) x [y0,y0+h)."""
    w = width - x0 if w is None else w
    h = height - y0 if h is None else h
    xs = np.arange(x0, x0 + w, dtype=np.float64)
    ys = np.arange(y0, y0 + h, dtype=np.float64)
    adelta = _sat_int(Minv[0, 0] * xs * AB_SCALE)
    bdelta = _sat_int(Minv[1, 0] * xs * AB_SCALE)
    rd = AB_SCALE // INTER_TAB_SIZE // 2
    X0 = _sat_int((Minv[0, 1] * ys + Minv[0, 2]) * AB_SCALE) + rd
    Y0 = _sat_int((Minv[1, 1] * ys + Minv[1, 2]) * AB_SCALE) + rd
    X = (X0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (Y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    return X, Y


def remap_bilinear_u8(src, X, Y):
    """remapBilinear for CV_8U with BORDER_CONSTANT(0): X, Y in 1/32 px."""
    H, W = src.shape[:2]
    s = src.reshape(H, W, -1).astype(np.int64)
    sx = np.clip(X >> INTER_BITS, -32768, 32767)
    sy = np.clip(Y >> INTER_BITS, -32768, 32767)
    fx = X & (INTER_TAB_SIZE - 1)
    fy = Y & (INTER_TAB_SIZE - 1)
    w00 = (32 - fx) * (32 - fy) * 32
    w01 = fx * (32 - fy) * 32
    w10 = (32 - fx) * fy * 32
    w11 = fx * fy * 32

    def fetch(yy, xx):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        v = s[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)]
        return v * ok[..., None]

    acc = (fetch(sy, sx) * w00[..., None] + fetch(sy, sx + 1) * w01[..., None] +
           fetch(sy + 1, sx) * w10[..., None] + fetch(sy + 1, sx + 1) * w11[..., None])
    out = (acc + (1 << (REMAP_COEF_BITS - 1))) >> REMAP_COEF_BITS
    return np.clip(out, 0, 255).astype(np.uint8).reshape(X.shape + src.shape[2:])


def warp_affine(src, M, dsize=(0, 0), flags=1, borderMode=0, borderValue=0):
    """cv2.warpAffine(src uint8 [H,W(,C)], M 2x3) with INTER_LINEAR / BORDER_CONSTANT 0."""
    src = np.asarray(src)
    assert src.dtype == np.uint8, 'OA-Mix only warps uint8 images and masks'
    H, W = src.shape[:2]
    dw, dh = (W, H) if (dsize is None or tuple(dsize) == (0, 0)) else (int(dsize[0]), int(dsize[1]))
    assert (dw, dh) == (W, H), 'OA-Mix always warps to the source size'
    Minv = invert_affine(np.asarray(M, dtype=np.float64))
    X, Y = warp_coords(Minv, dw, dh)
    return remap_bilinear_u8(src, X, Y)


def get_rotation_matrix_2d(center, angle, scale):
    cx, cy = float(np.float32(center[0])), float(np.float32(center[1]))   # Point2f
    a = angle * math.pi / 180.0
    alpha, beta = math.cos(a) * scale, math.sin(a) * scale
    return np.array([[alpha, beta, (1 - alpha) * cx - beta * cy],
                     [-beta, alpha, beta * cx + (1 - alpha) * cy]], dtype=np.float64)


# ------------------------------------------------------------------------------------------- blur / resize
_LOG2E = 1.4426950408889634
_LN2_HI = 6.93147180369123816490e-01
_LN2_LO = 1.90821492927058770002e-10
_INV_FACT = [1.0]
for _n in range(1, 14):
    _INV_FACT.append(_INV_FACT[-1] / _n)


def exp_det(x):
    """exp(x) for x <= 0 from IEEE +,-,* only (no libm): identical bits on the host and in the HIP kernel
    (same constants, same operation order, no fused multiply-add).  |error| < 1 ulp of double."""
    x = np.asarray(x, np.float64)
    k = np.rint(x * _LOG2E)
    r = (x - k * _LN2_HI) - k * _LN2_LO
    p = np.full_like(r, _INV_FACT[13])
    for n in range(12, -1, -1):
        p = p * r + _INV_FACT[n]
    return np.ldexp(p, k.astype(np.int64))


def gaussian_kernel_f32(sigma):
    """getGaussianKernel(ksize = round(sigma*8+1)|1, sigma, CV_32F)."""
    n = int(np.rint(sigma * 4 * 2 + 1)) | 1
    scale2x = -0.5 / (sigma * sigma)
    x = np.arange(n, dtype=np.float64) - (n - 1) * 0.5
    cf = exp_det(scale2x * x * x).astype(np.float32)      # std::exp in OpenCV; see exp_det
    s = 0.0
    for c in cf:                                           # sequential double sum, as getGaussianKernel
        s += float(c)
    inv = 1.0 / s
    return (cf.astype(np.float64) * inv).astype(np.float32)


def _reflect101(i, n):
    if n == 1:
        return np.zeros_like(i)
    p = 2 * (n - 1)
    i = np.mod(i, p)
    return np.where(i >= n, p - i, i)


def blur_profile_f32(a, sigma):
    """1-D Gaussian blur of a float32 profile, symmetric accumulation, BORDER_REFLECT_101."""
    a = np.asarray(a, np.float32)
    if sigma <= 0:
        return a.copy()
    k = gaussian_kernel_f32(sigma)
    r = (len(k) - 1) // 2
    n = len(a)
    idx = np.arange(n)
    acc = (k[r] * a).astype(np.float32)
    for j in range(1, r + 1):
        pair = (a[_reflect101(idx + j, n)] + a[_reflect101(idx - j, n)]).astype(np.float32)
        acc = (acc + (k[r + j] * pair).astype(np.float32)).astype(np.float32)
    return acc


def resize_profile_f32(p, dst_n):
    """1-D cv2.resize INTER_LINEAR of a float32 profile (half-pixel centres, edge clamp)."""
    p = np.asarray(p, np.float32)
    src_n = len(p)
    scale = 1.0 / (float(dst_n) / src_n)
    d = np.arange(dst_n, dtype=np.float64)
    f = (d + 0.5) * scale - 0.5
    s = np.floor(f).astype(np.int64)
    f = (f - s).astype(np.float32)
    lo = s < 0
    f[lo] = 0
    s[lo] = 0
    hi = s >= src_n - 1
    f[hi] = 0
    s[hi] = src_n - 1
    s1 = np.minimum(s + 1, src_n - 1)
    a0 = (np.float32(1.0) - f).astype(np.float32)
    return ((p[s] * a0).astype(np.float32) + (p[s1] * f).astype(np.float32)).astype(np.float32)


class SepMask(np.ndarray):
    """float32 [h,w,c] outer-product mask that remembers its two 1-D factors."""
    sep = None


def _as_sep(py, px, c):
    m = (py[:, None] * px[None, :]).astype(np.float32)
    out = np.repeat(m[:, :, None], c, axis=2).view(SepMask) if c else m.view(SepMask)
    out.sep = (py, px)
    return out


def _indicator_factors(mask):
    m2 = mask[..., 0] if mask.ndim == 3 else mask
    py = m2.max(axis=1)
    px = m2.max(axis=0)
    if np.array_equal(m2, py[:, None] * px[None, :]) and set(np.unique(m2).tolist()) <= {0.0, 1.0}:
        return py.astype(np.float32), px.astype(np.float32)
    return None


def gaussian_blur(src, ksize, sigmaX, sigmaY=0, borderType=4):
    src = np.asarray(src)
    assert src.dtype == np.float32 and tuple(ksize) == (0, 0), 'OA-Mix blurs float32 masks with ksize=(0,0)'
    if sigmaY <= 0:
        sigmaY = sigmaX
    fac = getattr(src, 'sep', None) or _indicator_factors(src)
    if fac is None:
        raise NotImplementedError('oracle GaussianBlur is defined for box-indicator masks only')
    py, px = fac
    return _as_sep(blur_profile_f32(py, sigmaY), blur_profile_f32(px, sigmaX), src.shape[2] if src.ndim == 3 else 0)


def resize(src, dsize, fx=0, fy=0, interpolation=1):
    src = np.asarray(src) if not isinstance(src, SepMask) else src
    assert src.dtype == np.float32, 'OA-Mix resizes float32 masks'
    W, H = int(dsize[0]), int(dsize[1])
    fac = getattr(src, 'sep', None) or _indicator_factors(src)
    if fac is None:
        raise NotImplementedError('oracle resize is defined for separable masks only')
    py, px = fac
    return np.asarray(_as_sep(resize_profile_f32(py, H), resize_profile_f32(px, W),
                              src.shape[2] if src.ndim == 3 else 0))


def box_mask_profiles(box, H, W, spatial_ratio, sigma_ratio):
    """The two 1-D profiles (My [H], Mx [W]) of OAMix._get_mask(box, ..., spatial_ratio, sigma_ratio)
    (oa_mix.py:74-93) under the separable definition above."""
    b = np.asarray(box)
    x1, y1, x2, y2 = np.array(b // spatial_ratio, dtype=np.int32)
    hq, wq = H // spatial_ratio, W // spatial_ratio
    ay = np.zeros(hq, np.float32)
    ax = np.zeros(wq, np.float32)
    ay[slice(y1, y2)] = 1.0
    ax[slice(x1, x2)] = 1.0
    sx = (x2 - x1) * sigma_ratio / 3 * 2
    sy = (y2 - y1) * sigma_ratio / 3 * 2
    if not (sx <= 0 or sy <= 0):
        ay, ax = blur_profile_f32(ay, sy), blur_profile_f32(ax, sx)
    return resize_profile_f32(ay, H), resize_profile_f32(ax, W)


# ------------------------------------------------------------------------------------------- saliency
def bgr2gray_u8(img):
    b, g, r = (img[..., i].astype(np.int64) for i in range(3))
    return ((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14).astype(np.uint8)


def resize_u8_linear(src, dst_w, dst_h):
    """bilinear resize of a uint8 plane, evaluated in float64 then rounded to nearest (INTER_LINEAR_EXACT
    is exact fixed point; ties are not reproduced bit-for-bit - the score is only compared with a threshold)."""
    H, W = src.shape

    def axis(n_src, n_dst):
        scale = n_src / n_dst
        f = (np.arange(n_dst) + 0.5) * scale - 0.5
        s = np.floor(f).astype(np.int64)
        f = f - s
        lo = s < 0
        f[lo], s[lo] = 0, 0
        hi = s >= n_src - 1
        f[hi], s[hi] = 0, n_src - 1
        return s, np.minimum(s + 1, n_src - 1), f
    y0, y1, fy = axis(H, dst_h)
    x0, x1, fx = axis(W, dst_w)
    s = src.astype(np.float64)
    top = s[y0][:, x0] * (1 - fx) + s[y0][:, x1] * fx
    bot = s[y1][:, x0] * (1 - fx) + s[y1][:, x1] * fx
    return np.clip(np.floor(top * (1 - fy)[:, None] + bot * fy[:, None] + 0.5), 0, 255).astype(np.uint8)


def _box3(a):
    """cv::blur 3x3, BORDER_REFLECT_101, float64."""
    p = np.pad(a, 1, mode='reflect')
    out = np.zeros_like(a)
    for dy in range(3):
        for dx in range(3):
            out += p[dy:dy + a.shape[0], dx:dx + a.shape[1]]
    return out / 9.0


def _gauss5_sigma8(a):
    k = np.array([math.exp(-0.5 * (i - 2) ** 2 / 64.0) for i in range(5)])
    k = k / k.sum()
    p = np.pad(a, 2, mode='reflect')
    tmp = sum(k[i] * p[:, i:i + a.shape[1]] for i in range(5))
    return sum(k[i] * tmp[i:i + a.shape[0], :] for i in range(5))


def resize_f32_linear(src, dst_w, dst_h):
    H, W = src.shape
    out_rows = np.stack([resize_profile_f32(src[r], dst_w) for r in range(H)])          # horizontal
    return np.stack([resize_profile_f32(out_rows[:, c], dst_h) for c in range(dst_w)], axis=1)


def spectral_residual_saliency(img):
    """StaticSaliencySpectralResidual.computeSaliency(img uint8 [h,w,3]) -> float32 [h,w] in [0,1]."""
    h, w = img.shape[:2]
    gray = bgr2gray_u8(img) if img.ndim == 3 else img
    small = resize_u8_linear(gray, 64, 64).astype(np.float64)
    F = np.fft.fft2(small)
    mag, ang = np.abs(F), np.angle(F)
    with np.errstate(divide='ignore'):
        loga = np.log(mag)
    resid = np.exp(loga - _box3(loga))
    G = np.fft.ifft2(resid * np.exp(1j * ang)) * (64 * 64)   # cv::dft(DFT_INVERSE) does not scale
    m = _gauss5_sigma8(np.abs(G))
    m = m * m
    m = (m / m.max()).astype(np.float32)
    return resize_f32_linear(m, w, h)


def saliency_score(img_crop):
    """oa_mix.py:107-111: mean of uint8(saliency_map * 255)."""
    sal = spectral_residual_saliency(img_crop)
    return np.mean((sal * 255).astype('uint8'))


def resize_u8_cv2(src, dsize):
    """cv2.resize(src uint8 [H,W(,C)], (Wn, Hn), interpolation=INTER_LINEAR): OpenCV's 8-bit fixed-point path
    (resize.cpp: HResizeLinear / VResizeLinear<uchar,int,short>, INTER_RESIZE_COEF_BITS = 11), restated; the HIP
    kernel csrc/imgxform.hip follows the same text.  PARITY UNPINNED (OpenCV not installed, no reference vectors)."""
    src = np.asarray(src)
    assert src.dtype == np.uint8
    Wn, Hn = int(dsize[0]), int(dsize[1])
    H, W = src.shape[:2]
    s3 = src.reshape(H, W, -1).astype(np.int64)
    dx = np.arange(Wn)
    fx = ((dx + 0.5) * (np.float64(W) / Wn) - 0.5).astype(np.float32)
    sx = np.floor(fx).astype(np.int64)
    fx = (fx - sx.astype(np.float32)).astype(np.float32)
    lo, hi = sx < 0, sx >= W - 1
    fx = np.where(lo | hi, np.float32(0), fx)
    sx = np.where(lo, 0, np.where(hi, W - 1, sx))
    x1 = np.minimum(sx + 1, W - 1)
    a0 = np.rint((np.float32(1) - fx) * np.float32(2048)).astype(np.int64)
    a1 = np.rint(fx * np.float32(2048)).astype(np.int64)
    dy = np.arange(Hn)
    fy = ((dy + 0.5) * (np.float64(H) / Hn) - 0.5).astype(np.float32)
    sy = np.floor(fy).astype(np.int64)
    fy = (fy - sy.astype(np.float32)).astype(np.float32)
    y0, y1 = np.clip(sy, 0, H - 1), np.clip(sy + 1, 0, H - 1)
    b0 = np.rint((np.float32(1) - fy) * np.float32(2048)).astype(np.int64)
    b1 = np.rint(fy * np.float32(2048)).astype(np.int64)
    hrow = s3[:, sx, :] * a0[None, :, None] + s3[:, x1, :] * a1[None, :, None]          # [H, Wn, C]
    h0, h1 = hrow[y0], hrow[y1]
    v = (((b0[:, None, None] * (h0 >> 4)) >> 16) + ((b1[:, None, None] * (h1 >> 4)) >> 16) + 2) >> 2
    out = np.clip(v, 0, 255).astype(np.uint8)
    return out.reshape((Hn, Wn) + src.shape[2:])


def imrescale_size(w, h, scale):
    """mmcv.rescale_size for a (long, short) tuple scale: (new_w, new_h) with int(x * f + 0.5), and f."""
    max_long, max_short = max(scale), min(scale)
    f = min(max_long / max(h, w), max_short / min(h, w))
    return int(w * float(f) + 0.5), int(h * float(f) + 0.5), f
