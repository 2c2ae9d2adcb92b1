// OA-Mix on the device: box-mask profiles, spectral-residual saliency, histogram LUT colour ops, fixed-point
// affine warps blended through analytic blurred masks, region-wise multi-level composition, object-aware
// mixing fused with normalisation.  gfx950 only.
//
// Replaces, behind the C ABI in include/oadg_hip.h (the Python class `OAMix` of oa-dg_amd/pipelines/oa_mix.py
// replays the reference's RNG stream on the host and enqueues these kernels):
//   mmdet/datasets/pipelines/oa_mix.py:74-93     _get_mask               -> oadg_oamix_box_profiles
//   mmdet/datasets/pipelines/oa_mix.py:95-120    get_fg_regions          -> oadg_oamix_saliency (+ profiles)
//   mmdet/datasets/pipelines/augmix.py:64-78,103 autocontrast/equalize/posterize/solarize (Pillow)
//                                                                       -> oadg_oamix_hist/_luts/_compose
//   mmdet/datasets/pipelines/augmix.py:83-188    rotate/shear/translate (cv2.warpAffine)
//   mmdet/datasets/pipelines/bbox_augmentation.py:31-88    bbox-only ops -> oadg_oamix_bbox_step
//   mmdet/datasets/pipelines/bbox_augmentation.py:240-272  bg-only ops   -> oadg_oamix_compose (kind BG_WARP)
//   mmdet/datasets/pipelines/oa_mix.py:221-236   multi-level chain       -> oadg_oamix_compose
//   mmdet/datasets/pipelines/oa_mix.py:281-309   object_aware_mixing     -> oadg_oamix_final
//   mmdet/datasets/pipelines/transforms.py:618-629,699-701 Pad/Normalize -> oadg_oamix_final/_normalize
//
// Every byte-producing kernel performs the oracle's arithmetic operation for operation (float32 / float64
// IEEE ops without contraction, OpenCV's 10+5-bit fixed-point coordinates and 15-bit integer bilinear
// weights), so outputs are compared bit-for-bit.  All kernels are HBM-streaming or box-local; masks are
// never materialised as H x W x 3 float images (25 MB each in the reference): a blurred box mask is the
// outer product My[y] * Mx[x] of two 1-D profiles.
#include "common.h"
#include <stdlib.h>
#include "oadg_hip.h"

namespace {

// ------------------------------------------------------------------------------------------------ helpers
__device__ __forceinline__ double exp_det(double x) {
    // oracle/cvleaves.py exp_det: same constants, same order, +,-,* only
    const double LOG2E = 1.4426950408889634;
    const double LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const double k = rint(x * LOG2E);
    const double r = (x - k * LN2_HI) - k * LN2_LO;
    double f[14];
    f[0] = 1.0;
#pragma unroll
    for (int n = 1; n < 14; ++n) f[n] = f[n - 1] / (double)n;
    double p = f[13];
#pragma unroll
    for (int n = 12; n >= 0; --n) p = p * r + f[n];
    return ldexp(p, (int)k);
}

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    const int p = 2 * (n - 1);
    i %= p;
    if (i < 0) i += p;
    return i >= n ? p - i : i;
}

__device__ __forceinline__ long sat_int(double v) {
    // cv::saturate_cast<int>(cvRound(v)): round to nearest even, clamp to the int range - exactly what v_cvt_i32_f64 does
    // in ONE instruction (the rint / fmin / fmax / double -> int64 sequence it replaces was ~25: four of them per pixel were
    // a third of the per-box blend kernel's instructions)
    return (long)__double2int_rn(v);
}

struct Warp {
    double m[6];
};

// OpenCV WarpAffineInvoker coordinates in 1/32 px for destination pixel (x, y)
__device__ __forceinline__ void warp_xy(const Warp& w, int x, int y, long& X, long& Y) {
    const long adelta = sat_int(w.m[0] * (double)x * 1024.0);
    const long bdelta = sat_int(w.m[3] * (double)x * 1024.0);
    const long X0 = sat_int((w.m[1] * (double)y + w.m[2]) * 1024.0) + 16;
    const long Y0 = sat_int((w.m[4] * (double)y + w.m[5]) * 1024.0) + 16;
    X = (X0 + adelta) >> 5;
    Y = (Y0 + bdelta) >> 5;
}

struct Tap {
    int o00, o01, o10, o11;   // pixel offsets (y*W + x), -1 = outside (reads as 0)
    int w00, w01, w10, w11;   // 15-bit integer weights
};

__device__ __forceinline__ Tap make_tap(long X, long Y, int H, int W) {
    long sx = X >> 5, sy = Y >> 5;
    sx = sx < -32768 ? -32768 : (sx > 32767 ? 32767 : sx);
    sy = sy < -32768 ? -32768 : (sy > 32767 ? 32767 : sy);
    const int fx = (int)(X & 31), fy = (int)(Y & 31);
    Tap t;
    t.w00 = (32 - fx) * (32 - fy) * 32;
    t.w01 = fx * (32 - fy) * 32;
    t.w10 = (32 - fx) * fy * 32;
    t.w11 = fx * fy * 32;
    const bool x0 = sx >= 0 && sx < W, x1 = sx + 1 >= 0 && sx + 1 < W;
    const bool y0 = sy >= 0 && sy < H, y1 = sy + 1 >= 0 && sy + 1 < H;
    t.o00 = (y0 && x0) ? (int)(sy * W + sx) : -1;
    t.o01 = (y0 && x1) ? (int)(sy * W + sx + 1) : -1;
    t.o10 = (y1 && x0) ? (int)((sy + 1) * W + sx) : -1;
    t.o11 = (y1 && x1) ? (int)((sy + 1) * W + sx + 1) : -1;
    return t;
}

__device__ __forceinline__ int tap_fetch(const Tap& t, const uint8_t* img, int stride, int c) {
    const int v00 = t.o00 >= 0 ? img[(size_t)t.o00 * stride + c] : 0;
    const int v01 = t.o01 >= 0 ? img[(size_t)t.o01 * stride + c] : 0;
    const int v10 = t.o10 >= 0 ? img[(size_t)t.o10 * stride + c] : 0;
    const int v11 = t.o11 >= 0 ? img[(size_t)t.o11 * stride + c] : 0;
    int r = (v00 * t.w00 + v01 * t.w01 + v10 * t.w10 + v11 * t.w11 + (1 << 14)) >> 15;
    return r < 0 ? 0 : (r > 255 ? 255 : r);
}

// ------------------------------------------------------------------------------------------------ profiles
constexpr int PROF_MAXQ = 2048;   // quarter-resolution length
constexpr int PROF_MAXK = 4096;   // Gaussian taps

// grid (n_boxes, 2): axis 0 -> My (length H), axis 1 -> Mx (length W)
__global__ __launch_bounds__(256) void box_profiles_kernel(const int* __restrict__ qbox,
                                                          const double* __restrict__ sigma, int H, int W,
                                                          int ratio, float* __restrict__ My,
                                                          float* __restrict__ Mx) {
    __shared__ float a[PROF_MAXQ];
    __shared__ float bl[PROF_MAXQ];
    __shared__ float kf[PROF_MAXK];
    __shared__ double s_inv;
    const int b = blockIdx.x, axis = blockIdx.y, tid = threadIdx.x;
    const int L = axis == 0 ? H : W;
    const int Lq = L / ratio;
    const int lo = axis == 0 ? qbox[4 * b + 1] : qbox[4 * b + 0];
    const int hi = axis == 0 ? qbox[4 * b + 3] : qbox[4 * b + 2];
    const double sg = axis == 0 ? sigma[2 * b + 1] : sigma[2 * b + 0];
    for (int i = tid; i < Lq; i += 256) a[i] = (i >= lo && i < hi) ? 1.0f : 0.0f;
    __syncthreads();
    if (sg > 0.0) {
        const int n = ((int)rint(sg * 4.0 * 2.0 + 1.0)) | 1;
        const int r = (n - 1) / 2;
        const double scale2x = -0.5 / (sg * sg);
        for (int i = tid; i < n; i += 256) {
            const double x = (double)i - (double)(n - 1) * 0.5;
            kf[i] = (float)exp_det(scale2x * x * x);
        }
        __syncthreads();
        if (tid == 0) {
            double s = 0.0;
            for (int i = 0; i < n; ++i) s += (double)kf[i];
            s_inv = 1.0 / s;
        }
        __syncthreads();
        const double inv = s_inv;
        for (int i = tid; i < n; i += 256) kf[i] = (float)((double)kf[i] * inv);
        __syncthreads();
        for (int i = tid; i < Lq; i += 256) {
            float acc = kf[r] * a[i];
            for (int j = 1; j <= r; ++j) {
                const float pair = a[reflect101(i + j, Lq)] + a[reflect101(i - j, Lq)];
                acc = acc + kf[r + j] * pair;
            }
            bl[i] = acc;
        }
    } else {
        for (int i = tid; i < Lq; i += 256) bl[i] = a[i];
    }
    __syncthreads();
    float* out = axis == 0 ? My + (size_t)b * H : Mx + (size_t)b * W;
    const double scale = 1.0 / ((double)L / (double)Lq);
    for (int d = tid; d < L; d += 256) {
        double f = ((double)d + 0.5) * scale - 0.5;
        long s = (long)floor(f);
        float ff = (float)(f - (double)s);
        if (s < 0) { ff = 0.f; s = 0; }
        if (s >= Lq - 1) { ff = 0.f; s = Lq - 1; }
        const long s1 = s + 1 < Lq ? s + 1 : Lq - 1;
        const float a0 = 1.0f - ff;
        out[d] = bl[s] * a0 + bl[s1] * ff;
    }
}

__global__ void fg_union_kernel(const float* __restrict__ My, const float* __restrict__ Mx, int n, int H,
                                int W, float* __restrict__ uf, uint8_t* __restrict__ u8) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= (long)H * W) return;
    const int y = (int)(p / W), x = (int)(p - (long)y * W);
    float m = 0.f;
    for (int i = 0; i < n; ++i) {
        const float v = My[(size_t)i * H + y] * Mx[(size_t)i * W + x];
        m = i == 0 ? v : fmaxf(m, v);
    }
    uf[p] = m;
    u8[p] = (uint8_t)(m * 255.0f);
}

// The same union for MANY boxes (a 4096-box image: the loop above is 8.6 G mask evaluations, 4.3 ms): every mask is zero
// outside its support rect (`rects` [n][4] = x0, y0, w, h; w <= 0: empty), and all mask values are >= 0, so the union is
// the maximum of 0 and the masks over their rects - one workgroup row per box, integer atomicMax on the float bits (the
// order of non-negative floats is the order of their bit patterns; a maximum does not depend on the visiting order:
// bit-identical to fg_union_kernel).  `uf` must be zero on entry.
__global__ __launch_bounds__(256) void fg_union_scatter_kernel(const float* __restrict__ My, const float* __restrict__ Mx,
                                                               const int* __restrict__ rects, int H, int W,
                                                               unsigned* __restrict__ uf_bits) {
    const int i = blockIdx.x;
    const int x0 = rects[4 * i], y0 = rects[4 * i + 1], rw = rects[4 * i + 2], rh = rects[4 * i + 3];
    if (rw <= 0 || rh <= 0) return;
    const float* my = My + (size_t)i * H;
    const float* mx = Mx + (size_t)i * W;
    const int area = rw * rh;
    for (int q = blockIdx.y * 256 + threadIdx.x; q < area; q += gridDim.y * 256) {
        const int yy = q / rw, xx = q - yy * rw;
        const int x = x0 + xx, y = y0 + yy;
        if (x < 0 || y < 0 || x >= W || y >= H) continue;
        const float v = my[y] * mx[x];
        if (v > 0.f) atomicMax(uf_bits + (size_t)y * W + x, __float_as_uint(v));
    }
}
__global__ void fg_union_u8_kernel(const float* __restrict__ uf, long npix, uint8_t* __restrict__ u8) {
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < npix) u8[p] = (uint8_t)(uf[p] * 255.0f);
}

// ------------------------------------------------------------------------------------------------ histogram / LUTs
__global__ __launch_bounds__(256) void hist_kernel(const uint8_t* __restrict__ img, long npix,
                                                   int* __restrict__ hist) {
    __shared__ int h[768];
    for (int i = threadIdx.x; i < 768; i += 256) h[i] = 0;
    __syncthreads();
    const long ngroups = npix / 4;   // 4 pixels = 12 bytes = 3 aligned dwords
    const unsigned* w = reinterpret_cast<const unsigned*>(img);
    for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < ngroups; g += (long)gridDim.x * 256) {
        const unsigned a = w[3 * g], b = w[3 * g + 1], c = w[3 * g + 2];
        atomicAdd(&h[0 * 256 + (a & 255)], 1);         atomicAdd(&h[1 * 256 + ((a >> 8) & 255)], 1);
        atomicAdd(&h[2 * 256 + ((a >> 16) & 255)], 1); atomicAdd(&h[0 * 256 + (a >> 24)], 1);
        atomicAdd(&h[1 * 256 + (b & 255)], 1);         atomicAdd(&h[2 * 256 + ((b >> 8) & 255)], 1);
        atomicAdd(&h[0 * 256 + ((b >> 16) & 255)], 1); atomicAdd(&h[1 * 256 + (b >> 24)], 1);
        atomicAdd(&h[2 * 256 + (c & 255)], 1);         atomicAdd(&h[0 * 256 + ((c >> 8) & 255)], 1);
        atomicAdd(&h[1 * 256 + ((c >> 16) & 255)], 1); atomicAdd(&h[2 * 256 + (c >> 24)], 1);
    }
    if (blockIdx.x == 0 && threadIdx.x < 3) {   // tail pixels
        for (long p = ngroups * 4; p < npix; ++p) atomicAdd(&h[threadIdx.x * 256 + img[p * 3 + threadIdx.x]], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 768; i += 256)
        if (h[i]) atomicAdd(&hist[i], h[i]);
}

// one block, thread = intensity; luts[0] = ImageOps.autocontrast, luts[1] = ImageOps.equalize
__global__ __launch_bounds__(256) void luts_kernel(const int* __restrict__ hist, uint8_t* __restrict__ luts) {
    __shared__ int h[256];
    __shared__ long pre[256], inc[256];
    __shared__ int s_lo, s_hi, s_nz, s_last;
    const int ix = threadIdx.x;
    for (int c = 0; c < 3; ++c) {
        __syncthreads();
        h[ix] = hist[c * 256 + ix];
        __syncthreads();
        // exclusive prefix sums, first / last non-empty bin and their count: block-parallel (integers: any order)
        if (ix == 0) { s_lo = 256; s_hi = -1; s_nz = 0; }
        inc[ix] = h[ix];
        __syncthreads();
        if (h[ix]) { atomicMin(&s_lo, ix); atomicMax(&s_hi, ix); atomicAdd(&s_nz, 1); }
        for (int o = 1; o < 256; o <<= 1) {
            const long v = ix >= o ? inc[ix - o] : 0;
            __syncthreads();
            inc[ix] += v;
            __syncthreads();
        }
        pre[ix] = inc[ix] - h[ix];
        __syncthreads();
        if (ix == 0) {
            s_last = s_hi >= 0 ? h[s_hi] : 0;     // the last non-empty bin's count
            if (s_lo > 255) s_lo = 255;           // empty image: Python's loop leaves lo = 255, hi = 0
            if (s_hi < 0) s_hi = 0;
        }
        __syncthreads();
        int ac = ix;
        if (s_hi > s_lo) {
            const double scale = 255.0 / (double)(s_hi - s_lo);
            const double offset = -(double)s_lo * scale;
            const double v = (double)ix * scale + offset;
            int q = (int)v;   // Python int(): truncation toward zero
            ac = q < 0 ? 0 : (q > 255 ? 255 : q);
        }
        luts[(0 * 3 + c) * 256 + ix] = (uint8_t)ac;
        int eq = ix;
        if (s_nz > 1) {
            const long total = pre[255] + h[255];
            const long step = (total - s_last) / 255;
            if (step) {
                const long q = (step / 2 + pre[ix]) / step;
                eq = q > 255 ? 255 : (int)q;
            }
        }
        luts[(1 * 3 + c) * 256 + ix] = (uint8_t)eq;
    }
}

// ------------------------------------------------------------------------------------------------ bbox-only step
// phase 1: blended rect -> scratch ; phase 2: scratch -> image.  rect = support of the blurred mask.
__global__ void bbox_blend_kernel(const uint8_t* __restrict__ img, int H, int W, Warp wp, int rx0, int ry0,
                                  int rw, int rh, const float* __restrict__ My, const float* __restrict__ Mx,
                                  uint8_t* __restrict__ scratch) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rw * rh) return;
    const int yy = i / rw, xx = i - yy * rw;
    const int x = rx0 + xx, y = ry0 + yy;
    long X, Y;
    warp_xy(wp, x, y, X, Y);
    const Tap t = make_tap(X, Y, H, W);
    const float b = My[y] * Mx[x];
    const float m = 1.0f - b;
    const float om = 1.0f - m;
    const size_t p = ((size_t)y * W + x) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float wv = (float)tap_fetch(t, img, 3, c);
        const float v = (float)img[p + c] * m + wv * om;
        scratch[(size_t)i * 3 + c] = (uint8_t)v;
    }
}

__global__ void rect_copy_kernel(uint8_t* __restrict__ img, int W, int rx0, int ry0, int rw, int rh,
                                 const uint8_t* __restrict__ scratch) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rw * rh) return;
    const int yy = i / rw, xx = i - yy * rw;
    const size_t p = ((size_t)(ry0 + yy) * W + rx0 + xx) * 3;
    img[p] = scratch[(size_t)i * 3];
    img[p + 1] = scratch[(size_t)i * 3 + 1];
    img[p + 2] = scratch[(size_t)i * 3 + 2];
}

// The same two phases for MANY boxes per launch: `steps` (device) are sorted into dependency levels by the host
// (pipelines/oa_mix.py: a box joins a later level than every earlier box whose written rect meets its read footprint
// or vice versa), so all boxes of one level may be blended concurrently and the reference's sequential result is kept
// bit for bit.  Workgroup b of a level's launch owns 256 consecutive pixels of one box's rect (binary search of its
// global tile number in the prefix `tile_prefix`).
// (round 5: the workgroup's 256 threads look at 256 prefix entries at ONCE - the prefix is ascending, so the step is the
//  number of entries <= tile, minus one: one memory round trip and a ballot instead of log2(count) DEPENDENT loads at the
//  head of a kernel whose whole life is a chain of five round trips.  Every thread of the 256-thread block must call it.)
__device__ __forceinline__ int find_step(const int* __restrict__ tile_prefix, int first, int count, int tile) {
    __shared__ int below_sh[4];
    const int wave = threadIdx.x >> 6;
    int below = 0;
    for (int base = 0; base < count; base += 256) {
        const int j = base + (int)threadIdx.x;
        const bool le = j < count && tile_prefix[first + j] <= tile;
        const unsigned long long m = __ballot(le);
        if ((threadIdx.x & 63) == 0) below_sh[wave] = __popcll(m);
        __syncthreads();
        const int c = below_sh[0] + below_sh[1] + below_sh[2] + below_sh[3];
        __syncthreads();
        below += c;
        if (c < 256) break;                        // (uniform) the prefix passed `tile` inside this chunk
    }
    return first + below - 1;                      // last step whose prefix <= tile (tile_prefix[first] <= tile always)
}

// 8 source bytes = the two horizontally adjacent taps (3 + 3 bytes) of one source row in ONE unaligned 8-byte load
__device__ __forceinline__ void tap_row(const uint8_t* img, long pix, int v0[3], int v1[3]) {
    uint2 r;
    __builtin_memcpy(&r, img + pix * 3, 8);
    v0[0] = r.x & 255; v0[1] = (r.x >> 8) & 255; v0[2] = (r.x >> 16) & 255;
    v1[0] = r.x >> 24; v1[1] = r.y & 255; v1[2] = (r.y >> 8) & 255;
}

// warped pixel (OpenCV remapBilinear, BORDER_CONSTANT 0) at 1/32-px coordinates (X, Y): the interior case reads its
// four taps with two 8-byte loads, the border cases go through tap_fetch; identical integer arithmetic
__device__ __forceinline__ void warp_pixel(const uint8_t* img, int H, int W, long X, long Y, int out[3]) {
    const long sx = X >> 5, sy = Y >> 5;
    if (sx >= 0 && sx + 2 < W && sy >= 0 && sy + 1 < H) {
        const int fx = (int)(X & 31), fy = (int)(Y & 31);
        const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
        int a0[3], a1[3], b0[3], b1[3];
        tap_row(img, sy * W + sx, a0, a1);
        tap_row(img, (sy + 1) * W + sx, b0, b1);
#pragma unroll
        for (int c = 0; c < 3; ++c) out[c] = (a0[c] * w00 + a1[c] * w01 + b0[c] * w10 + b1[c] * w11 + (1 << 14)) >> 15;
    } else {
        const Tap t = make_tap(X, Y, H, W);
#pragma unroll
        for (int c = 0; c < 3; ++c) out[c] = tap_fetch(t, img, 3, c);
    }
}

// 4 consecutive pixels of a step's rect (row-major inside the rect, from pixel i0): the blended bytes packed into pk[3]
__device__ __forceinline__ void blend4(const uint8_t* img, int H, int W, const oadg_bbox_step& st, const float* my,
                                       const float* mx, int i0, unsigned pk[3]) {
    const int rw = st.rect[2], rh = st.rect[3];
    Warp wp;
#pragma unroll
    for (int k = 0; k < 6; ++k) wp.m[k] = st.minv[k];
    pk[0] = pk[1] = pk[2] = 0u;
    int yy = i0 / rw, xx = i0 - yy * rw;           // (one division per thread; the next pixels step along the rect's rows)
    int my_row = -1;
    float my_val = 0.f;
    long rowX = 0, rowY = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = i0 + u;
        if (u > 0) {
            ++xx;
            while (xx >= rw) { xx -= rw; ++yy; }
        }
        if (i < rw * rh) {
            const int x = st.rect[0] + xx, y = st.rect[1] + yy;
            if (y != my_row) {                     // (the four pixels mostly share a row: its terms once - warp_xy's X0 / Y0)
                my_row = y; my_val = my[y];
                rowX = sat_int((wp.m[1] * (double)y + wp.m[2]) * 1024.0) + 16;
                rowY = sat_int((wp.m[4] * (double)y + wp.m[5]) * 1024.0) + 16;
            }
            const long X = (rowX + sat_int(wp.m[0] * (double)x * 1024.0)) >> 5;
            const long Y = (rowY + sat_int(wp.m[3] * (double)x * 1024.0)) >> 5;
            int wv[3];
            warp_pixel(img, H, W, X, Y, wv);
            const float b = my_val * mx[x];
            const float m = 1.0f - b;
            const float om = 1.0f - m;
            const size_t p = ((size_t)y * W + x) * 3;
            // the pixel's three bytes in one unaligned 4-byte load (the image's very last pixel: byte by byte)
            unsigned px;
            if (p + 4 <= (size_t)H * W * 3) __builtin_memcpy(&px, img + p, 4);
            else px = (unsigned)img[p] | ((unsigned)img[p + 1] << 8) | ((unsigned)img[p + 2] << 16);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float v = (float)((px >> (8 * c)) & 255u) * m + (float)wv[c] * om;
                const unsigned byte = (unsigned)(uint8_t)v;
                const int bi = u * 3 + c;
                pk[bi >> 2] |= byte << ((bi & 3) * 8);
            }
        }
    }
}

// one thread = 4 consecutive pixels of a rect (row-major inside the rect): 12 result bytes as three aligned dwords
__device__ __forceinline__ void blend_tile(const uint8_t* __restrict__ img, int H, int W,
                                           const oadg_bbox_step* __restrict__ steps, const int* __restrict__ tile_prefix,
                                           int first, int count, int tile, const float* __restrict__ My,
                                           const float* __restrict__ Mx, uint8_t* __restrict__ scratch) {
    const int s = find_step(tile_prefix, first, count, tile);
    const oadg_bbox_step st = steps[s];
    const int rw = st.rect[2], rh = st.rect[3];
    const int i0 = ((tile - tile_prefix[s]) * 256 + threadIdx.x) * 4;
    if (i0 >= rw * rh) return;
    unsigned pk[3];
    blend4(img, H, W, st, My + (size_t)st.row * H, Mx + (size_t)st.row * W, i0, pk);
    uint8_t* ob = scratch + st.scratch_off + (size_t)i0 * 3;                                   // scratch_off % 4 == 0
    const int left = rw * rh - i0;
    if (left >= 4) {
        unsigned* out = reinterpret_cast<unsigned*>(ob);
        out[0] = pk[0]; out[1] = pk[1]; out[2] = pk[2];
    } else {
        // tail group of a rect whose area is not a multiple of 4: only the bytes of its own pixels - the next rect's
        // scratch slice starts at the next 4-byte boundary and belongs to another workgroup of this launch
        for (int bi = 0; bi < left * 3; ++bi) ob[bi] = (uint8_t)(pk[bi >> 2] >> ((bi & 3) * 8));
    }
}

__global__ __launch_bounds__(256) void bbox_blend_multi_kernel(const uint8_t* __restrict__ img, int H, int W,
                                                               const oadg_bbox_step* __restrict__ steps,
                                                               const int* __restrict__ tile_prefix, int first,
                                                               int count, int tile_base,
                                                               const float* __restrict__ My,
                                                               const float* __restrict__ Mx,
                                                               uint8_t* __restrict__ scratch) {
    blend_tile(img, H, W, steps, tile_prefix, first, count, tile_base + blockIdx.x, My, Mx, scratch);
}

// one thread = the 4 consecutive pixels its blend thread wrote: 12 scratch bytes as three aligned dwords (the tail group of a
// rect byte by byte), 12 image bytes.  (Rounds 2-5: one pixel per thread, four workgroups per tile - three single-byte loads
// and stores per thread; with the chains of a whole batch x three mixture chains in one launch (round 6) the copy pass had
// become the longer half of a level: 23.6 us against the blend's 20.)
__device__ __forceinline__ void copy_tile(uint8_t* __restrict__ img, int W, const oadg_bbox_step* __restrict__ steps,
                                          const int* __restrict__ tile_prefix, int first, int count, int tile,
                                          const uint8_t* __restrict__ scratch) {
    const int s = find_step(tile_prefix, first, count, tile);
    const oadg_bbox_step st = steps[s];
    const int rw = st.rect[2], area = rw * st.rect[3];
    const int i0 = ((tile - tile_prefix[s]) * 256 + threadIdx.x) * 4;
    if (i0 >= area) return;
    const uint8_t* in = scratch + st.scratch_off + (size_t)i0 * 3;                              // 4-byte aligned
    unsigned pk[3] = {0u, 0u, 0u};
    const int left = area - i0;
    if (left >= 4) {
        const unsigned* q = reinterpret_cast<const unsigned*>(in);
        pk[0] = q[0]; pk[1] = q[1]; pk[2] = q[2];
    } else {
        for (int bi = 0; bi < left * 3; ++bi) pk[bi >> 2] |= (unsigned)in[bi] << ((bi & 3) * 8);
    }
    const int yy0 = i0 / rw, xx0 = i0 - yy0 * rw;
    if (left >= 4 && xx0 + 3 < rw) {                 // the four pixels lie side by side in one image row: 12 contiguous bytes
        uint8_t* o = img + ((size_t)(st.rect[1] + yy0) * W + st.rect[0] + xx0) * 3;
        if ((reinterpret_cast<uintptr_t>(o) & 3u) == 0) {
            unsigned* od = reinterpret_cast<unsigned*>(o);
            od[0] = pk[0]; od[1] = pk[1]; od[2] = pk[2];
        } else {
#pragma unroll
            for (int bi = 0; bi < 12; ++bi) o[bi] = (uint8_t)(pk[bi >> 2] >> ((bi & 3) * 8));
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = i0 + u;
        if (i < area) {
            const int yy = i / rw, xx = i - yy * rw;
            uint8_t* o = img + ((size_t)(st.rect[1] + yy) * W + st.rect[0] + xx) * 3;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const int bi = u * 3 + ch;
                o[ch] = (uint8_t)(pk[bi >> 2] >> ((bi & 3) * 8));
            }
        }
    }
}

// (one workgroup per 1024-pixel tile of the blend kernel)
__global__ __launch_bounds__(256) void rect_copy_multi_kernel(uint8_t* __restrict__ img, int W,
                                                              const oadg_bbox_step* __restrict__ steps,
                                                              const int* __restrict__ tile_prefix, int first,
                                                              int count, int tile_base,
                                                              const uint8_t* __restrict__ scratch) {
    copy_tile(img, W, steps, tile_prefix, first, count, tile_base + blockIdx.x, scratch);
}

// ---- the chains of SEVERAL images in lockstep (round 4) -----------------------------------------------------------------
// The images of a batch are augmented independently (own work image, own mask profiles, own scratch), and a
// bboxes_only_* chain needs its launch pair per dependency level only because of the dependencies INSIDE its image.  The
// pipeline therefore advances the chains of all images of the batch together: level l of every chain in ONE launch pair -
// launches per batch = 2 x the deepest chain instead of 2 x the sum over the images.  Up to CHAIN_MAX chains per launch; their
// descriptors travel by value in the launch arguments and are selected with compile-time indices only (a dynamically
// indexed argument block would be copied to scratch memory per thread).
// chains per launch: the images of a batch x the chains of an image that are ready together (three mixture chains, up to
// three region ops each; round 6) - 24 descriptors of 72 bytes travel in the kernel arguments
constexpr int CHAIN_MAX = 24;
struct ChainImg {
    uint8_t* img;
    const oadg_bbox_step* steps;
    const int* tile_prefix;
    const float* My;
    const float* Mx;
    uint8_t* scratch;
    int H, W, first, count, tile_base, block0;        // block0: first workgroup of this image in the launch (blend tiles)
};
struct ChainLevel {
    ChainImg im[CHAIN_MAX];
    int n;
};
__device__ __forceinline__ ChainImg pick_image(const ChainLevel& a, int tile) {
    ChainImg s = a.im[0];
#pragma unroll
    for (int k = 1; k < CHAIN_MAX; ++k)
        if (k < a.n && tile >= a.im[k].block0) s = a.im[k];
    return s;
}
__global__ __launch_bounds__(256) void bbox_blend_imgs_kernel(ChainLevel a) {
    const ChainImg s = pick_image(a, blockIdx.x);
    blend_tile(s.img, s.H, s.W, s.steps, s.tile_prefix, s.first, s.count, s.tile_base + (blockIdx.x - s.block0), s.My, s.Mx,
               s.scratch);
}
__global__ __launch_bounds__(256) void rect_copy_imgs_kernel(ChainLevel a) {
    const int tile = blockIdx.x;
    const ChainImg s = pick_image(a, tile);
    copy_tile(s.img, s.W, s.steps, s.tile_prefix, s.first, s.count, s.tile_base + (tile - s.block0), s.scratch);
}

// (Round 6, three ways around the launch pair per level, all byte-exact, all measured on BASELINE config 5 and dropped:
//  1. every chain on a GROUP of workgroups that walk its levels and meet at a counter in memory between them (write-through
//     stores + one agent-scope acquire per meeting, the MI355X guide's hand-over recipe): 48 us per level against the pair's
//     20 - 80.5 ms per step against 47.1 (64 arrivals per meeting, byte-wide write-through stores), default step 30.2 / 27.6;
//  2. a chain's runs of narrow levels inside ONE 1024-thread workgroup (barrier-only, tables double-buffered in LDS): a
//     level of a 4096-box image turned out to be ~40 rects of ~140 x 140 pixels = ~200 tiles (the blur support of a
//     48-pixel box is three sigmas of 0.3 x its quarter-resolution size on every side) - 5 of ~110 levels qualified;
//  3. one launch per level with workgroup = rect, blended in registers and written in place behind a workgroup barrier
//     (round 4 had tried it with the rect's tiles walked one after the other: 57 us): 37.7 us per level - a rect is 5 - 20
//     tiles of work for ONE workgroup where the pair spreads them over as many workgroups; 50.9 ms per step against 47.2.
//  The pair IS the parallel form: ~1,600 tile workgroups per level (8 images) = one resident wave of the chip, 12 + 8.4 us.
//  profiles/r06_oamix_chain_experiments.txt.)

// ------------------------------------------------------------------------------------------------ compose
struct ComposeArgs {
    oadg_region_op op[3];     // [0],[1]: random boxes, [2]: outside
    int rect[2][4];           // x1,y1,x2,y2 (exclusive) of the random boxes
    int n_rects;
};

// ``sp``: the pixel's own three source bytes (the caller has them in registers: src[p .. p + 2])
__device__ __forceinline__ void apply_region_op(const oadg_region_op& op, const uint8_t* src, const uint8_t sp[3], int H, int W,
                                                int x, int y, const uint8_t* lut_s, const float* uf,
                                                const uint8_t* u8, uint8_t out[3]) {
    const size_t p = ((size_t)y * W + x) * 3;
    switch (op.kind) {
        case OADG_OP_LUT_AUTOCONTRAST:
        case OADG_OP_LUT_EQUALIZE: {
            const uint8_t* l = lut_s + (op.kind == OADG_OP_LUT_EQUALIZE ? 768 : 0);
            out[0] = l[sp[0]]; out[1] = l[256 + sp[1]]; out[2] = l[512 + sp[2]];
            break;
        }
        case OADG_OP_POSTERIZE:
            for (int c = 0; c < 3; ++c) out[c] = sp[c] & (uint8_t)op.param;
            break;
        case OADG_OP_SOLARIZE:
            for (int c = 0; c < 3; ++c) { const int v = sp[c]; out[c] = (uint8_t)(v < op.param ? v : 255 - v); }
            break;
        case OADG_OP_IMAGE: {
            const uint8_t* im = (const uint8_t*)op.image;
            out[0] = im[p]; out[1] = im[p + 1]; out[2] = im[p + 2];
            break;
        }
        case OADG_OP_BG_WARP: {
            Warp wp;
            for (int i = 0; i < 6; ++i) wp.m[i] = op.minv[i];
            long X, Y;
            warp_xy(wp, x, y, X, Y);
            const Tap t = make_tap(X, Y, H, W);
            const double wm = (double)tap_fetch(t, u8, 1, 0) / 255.0;
            const double keep = fmax((double)uf[(size_t)y * W + x], wm);
            const double ok = 1.0 - keep;
            int wv[3];
            warp_pixel(src, H, W, X, Y, wv);         // (interior pixels: two 8-byte loads instead of twelve single bytes)
            for (int c = 0; c < 3; ++c) {
                const double v = keep * (double)sp[c] + ok * (double)wv[c];
                out[c] = (uint8_t)v;
            }
            break;
        }
        case OADG_OP_ENH_BRIGHTNESS:
        case OADG_OP_ENH_COLOR:
        case OADG_OP_ENH_CONTRAST:
        case OADG_OP_ENH_SHARPNESS: {
            // PIL.ImageEnhance: Image.blend(degenerate, image, factor) with float32 arithmetic; the degenerate
            // image is black / the pixel's luma / the global mean luma / the 3x3 SMOOTH filter output
            const float alpha = (float)op.minv[0];
            int deg[3];
            if (op.kind == OADG_OP_ENH_BRIGHTNESS) {
                deg[0] = deg[1] = deg[2] = 0;
            } else if (op.kind == OADG_OP_ENH_COLOR) {
                deg[0] = deg[1] = deg[2] = (sp[0] * 19595 + sp[1] * 38470 + sp[2] * 7471 + 0x8000) >> 16;
            } else if (op.kind == OADG_OP_ENH_CONTRAST) {
                const long long sum = *reinterpret_cast<const long long*>(op.image);
                deg[0] = deg[1] = deg[2] = (int)((double)sum / (double)((long)H * W) + 0.5);
            } else {
                if (x == 0 || y == 0 || x == W - 1 || y == H - 1) {
                    deg[0] = sp[0]; deg[1] = sp[1]; deg[2] = sp[2];
                } else {
                    const float k1 = 1.0f / 13.0f, k5 = 5.0f / 13.0f;
                    for (int c = 0; c < 3; ++c) {
                        float ss = 0.5f;
                        for (int dy = 1; dy >= -1; --dy) {           // rows y+1, y, y-1 (Filter.c order)
                            const uint8_t* r = src + ((size_t)(y + dy) * W + x) * 3 + c;
                            const float kc = dy == 0 ? k5 : k1;
                            ss = ss + (((float)r[-3] * k1 + (float)r[0] * kc) + (float)r[3] * k1);
                        }
                        deg[c] = ss <= 0.f ? 0 : (ss >= 255.f ? 255 : (int)ss);
                    }
                }
            }
            for (int c = 0; c < 3; ++c) {
                const int a0 = deg[c], b0 = sp[c];
                const float t = (float)a0 + alpha * (float)(b0 - a0);
                if (alpha >= 0.f && alpha <= 1.f) out[c] = (uint8_t)t;
                else out[c] = t <= 0.f ? 0 : (t >= 255.f ? 255 : (uint8_t)t);
            }
            break;
        }
        case OADG_OP_WARP_NEG: {
            Warp wp;
            for (int i = 0; i < 6; ++i) wp.m[i] = op.minv[i];
            long X, Y;
            warp_xy(wp, x, y, X, Y);
            int wv[3];
            warp_pixel(src, H, W, X, Y, wv);
            for (int c = 0; c < 3; ++c) out[c] = (uint8_t)(0 - wv[c]);
            break;
        }
        default:
            out[0] = sp[0]; out[1] = sp[1]; out[2] = sp[2];
    }
}

__global__ __launch_bounds__(256) void compose_kernel(const uint8_t* __restrict__ src,
                                                      uint8_t* __restrict__ dst, int H, int W,
                                                      ComposeArgs a, const uint8_t* __restrict__ luts,
                                                      const float* __restrict__ uf,
                                                      const uint8_t* __restrict__ u8,
                                                      float* __restrict__ acc, float acc_w, int acc_mode) {
    __shared__ uint8_t lut_s[2 * 768];
    if (luts)
        for (int i = threadIdx.x; i < 2 * 768; i += 256) lut_s[i] = luts[i];
    __syncthreads();
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < (long)H * W; p += (long)gridDim.x * 256) {
        const int y = (int)(p / W), x = (int)(p - (long)y * W);
        int r = 2;
        for (int k = 0; k < a.n_rects; ++k)
            if (x >= a.rect[k][0] && x < a.rect[k][2] && y >= a.rect[k][1] && y < a.rect[k][3]) r = k;
        uint8_t o[3];
        const uint8_t sp[3] = {src[p * 3], src[p * 3 + 1], src[p * 3 + 2]};
        apply_region_op(a.op[r], src, sp, H, W, x, y, lut_s, uf, u8, o);
        dst[p * 3] = o[0]; dst[p * 3 + 1] = o[1]; dst[p * 3 + 2] = o[2];
        if (acc_mode) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float t = acc_w * (float)o[c];
                acc[p * 3 + c] = acc_mode == 1 ? t : acc[p * 3 + c] + t;
            }
        }
    }
}

// the same pass with FOUR pixels per thread (W % 4 == 0, 16-byte aligned buffers): source, result and accumulator travel as
// three 4-byte / 16-byte accesses per thread instead of twelve single-byte ones (round 6: the one-pixel form moved a
// 6.3 MB image at 0.25 TB/s - 52 us per compose step, 25 steps per training step).  Per pixel the same arithmetic: byte-identical.
__global__ __launch_bounds__(256) void compose4_kernel(const uint8_t* __restrict__ src,
                                                       uint8_t* __restrict__ dst, int H, int W,
                                                       ComposeArgs a, const uint8_t* __restrict__ luts,
                                                       const float* __restrict__ uf,
                                                       const uint8_t* __restrict__ u8,
                                                       float* __restrict__ acc, float acc_w, int acc_mode) {
    __shared__ uint8_t lut_s[2 * 768];
    if (luts)
        for (int i = threadIdx.x; i < 2 * 768; i += 256) lut_s[i] = luts[i];
    __syncthreads();
    const long nq = (long)H * W / 4;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < nq; q += (long)gridDim.x * 256) {
        const long p0 = q * 4;
        const int y = (int)(p0 / W), x0 = (int)(p0 - (long)y * W);
        const unsigned* s32 = reinterpret_cast<const unsigned*>(src + p0 * 3);
        const unsigned sw[3] = {s32[0], s32[1], s32[2]};
        unsigned ow[3] = {0u, 0u, 0u};
        float of[12];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int x = x0 + u;
            int r = 2;
            for (int k = 0; k < a.n_rects; ++k)
                if (x >= a.rect[k][0] && x < a.rect[k][2] && y >= a.rect[k][1] && y < a.rect[k][3]) r = k;
            uint8_t sp[3], o[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) { const int bi = u * 3 + c; sp[c] = (uint8_t)(sw[bi >> 2] >> ((bi & 3) * 8)); }
            apply_region_op(a.op[r], src, sp, H, W, x, y, lut_s, uf, u8, o);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int bi = u * 3 + c;
                ow[bi >> 2] |= (unsigned)o[c] << ((bi & 3) * 8);
                of[bi] = acc_w * (float)o[c];
            }
        }
        unsigned* d32 = reinterpret_cast<unsigned*>(dst + p0 * 3);
        d32[0] = ow[0]; d32[1] = ow[1]; d32[2] = ow[2];
        if (acc_mode) {
            float4* a4 = reinterpret_cast<float4*>(acc + p0 * 3);
#pragma unroll
            for (int v = 0; v < 3; ++v) {
                float4 t = make_float4(of[4 * v], of[4 * v + 1], of[4 * v + 2], of[4 * v + 3]);
                if (acc_mode != 1) {
                    const float4 old = a4[v];
                    t.x = old.x + t.x; t.y = old.y + t.y; t.z = old.z + t.z; t.w = old.w + t.w;
                }
                a4[v] = t;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ final mix
struct NormArgs {
    float mean[3], stdinv[3];
    int to_rgb;
};

template <typename T>
__device__ __forceinline__ void store_norm(T* out, size_t idx, float v);
template <>
__device__ __forceinline__ void store_norm<float>(float* out, size_t idx, float v) { out[idx] = v; }
template <>
__device__ __forceinline__ void store_norm<unsigned short>(unsigned short* out, size_t idx, float v) {
    out[idx] = f32_to_bf16(v);
}

template <typename T>
__device__ __forceinline__ void write_norm(T* out, int Wp, int x, int y, const uint8_t px[3], const NormArgs& n) {
    // mmcv.imnormalize: BGR->RGB swap, then (x - mean) * (1/std) in float32; NHWC, padded right/bottom with 0
    const size_t base = ((size_t)y * Wp + x) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int sc = n.to_rgb ? 2 - c : c;
        store_norm<T>(out, base + c, ((float)px[sc] - n.mean[c]) * n.stdinv[c]);
    }
}

// Tile bins of the mixing targets (images with many of them: every target at every pixel is n_tg x H x W mask
// evaluations - 1.2 ms for the ~4100 targets of a 4096-box image).  A target whose mask is zero at a pixel leaves the
// pixel's running sums untouched (mask_sum + 0, max(mask_max, 0), weight 0: every update adds +0), so a pixel only has to
// visit, IN TARGET ORDER, the targets whose rect (support rect of the fg mask / the random box) reaches its 32 x 32 tile.
// One workgroup per tile compacts the target indices in order (ballot + prefix over the waves): `lists` [tiles][cap]
// uint16, `counts` [tiles] (a count above cap = list truncated: the pixels of that tile visit every target).
constexpr int MIX_TS = 32;
__global__ __launch_bounds__(256) void mix_bins_kernel(const oadg_mix_target* __restrict__ tg, int n_tg,
                                                       const int* __restrict__ fg_rects, int tiles_x, int cap,
                                                       int* __restrict__ counts, unsigned short* __restrict__ lists) {
    __shared__ int wcnt[4];
    const int tile = blockIdx.x, ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int X0 = tx * MIX_TS, Y0 = ty * MIX_TS, X1 = X0 + MIX_TS, Y1 = Y0 + MIX_TS;      // [X0, X1) x [Y0, Y1)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned short* out = lists + (size_t)tile * cap;
    int total = 0;
    for (int base = 0; base < n_tg; base += 256) {
        const int t = base + threadIdx.x;
        bool hit = false;
        if (t < n_tg) {
            const oadg_mix_target g = tg[t];
            int x0, y0, x1, y1;
            if (g.fg_index >= 0) {
                const int* r = fg_rects + 4 * g.fg_index;
                x0 = r[0]; y0 = r[1]; x1 = r[0] + r[2]; y1 = r[1] + r[3];
            } else {
                x0 = g.rect[0]; y0 = g.rect[1]; x1 = g.rect[2]; y1 = g.rect[3];
            }
            hit = x0 < X1 && x1 > X0 && y0 < Y1 && y1 > Y0;
        }
        const unsigned long long m = __ballot(hit);
        __syncthreads();
        if (lane == 0) wcnt[wave] = __popcll(m);
        __syncthreads();
        int before = 0, nh = 0;
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) {
            if (w2 < wave) before += wcnt[w2];
            nh += wcnt[w2];
        }
        if (hit) {
            const int pos = total + before + __popcll(m & ((1ull << lane) - 1ull));
            if (pos < cap) out[pos] = (unsigned short)t;
        }
        total += nh;
    }
    if (threadIdx.x == 0) counts[tile] = total;
}

template <typename T, bool BINNED>
__global__ __launch_bounds__(256) void final_mix_kernel(const uint8_t* __restrict__ img,
                                                        const float* __restrict__ acc, int H, int W,
                                                        const oadg_mix_target* __restrict__ tg, int n_tg,
                                                        const float* __restrict__ My,
                                                        const float* __restrict__ Mx, double m_beta,
                                                        NormArgs nrm, uint8_t* __restrict__ out_u8,
                                                        T* __restrict__ out_norm, int Hp, int Wp,
                                                        const int* __restrict__ counts,
                                                        const unsigned short* __restrict__ lists, int tiles_x, int cap) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= (long)Hp * Wp) return;
    const int y = (int)(p / Wp), x = (int)(p - (long)y * Wp);
    if (y >= H || x >= W) {   // padding
        if (out_norm) for (int c = 0; c < 3; ++c) store_norm<T>(out_norm, (size_t)p * 3 + c, 0.f);
        return;
    }
    const size_t q = ((size_t)y * W + x) * 3;
    float im[3], ag[3], orig[3] = {0.f, 0.f, 0.f}, aug[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3; ++c) { im[c] = (float)img[q + c]; ag[c] = acc[q + c]; }
    float mask_sum = 0.f, mask_max = 0.f;
    int n_visit = n_tg;
    const unsigned short* list = nullptr;
    if (BINNED) {
        const int tile = (y / MIX_TS) * tiles_x + x / MIX_TS;
        const int cnt = counts[tile];
        if (cnt <= cap) { n_visit = cnt; list = lists + (size_t)tile * cap; }
    }
    for (int j = 0; j < n_visit; ++j) {
        const int t = (BINNED && list) ? (int)list[j] : j;
        const oadg_mix_target g = tg[t];
        float mask;
        if (g.fg_index >= 0) mask = My[(size_t)g.fg_index * H + y] * Mx[(size_t)g.fg_index * W + x];
        else mask = (x >= g.rect[0] && x < g.rect[2] && y >= g.rect[1] && y < g.rect[3]) ? 1.0f : 0.0f;
        mask_sum = mask_sum + mask;
        mask_max = j == 0 ? mask : fmaxf(mask_max, mask);         // (masks are >= 0 and mask_max starts at 0: the same)
        const float overlap = mask_sum - mask_max;
        const float wgt = mask - overlap * 0.5f;
        const float a = 1.0f - g.m_oa;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            orig[c] = orig[c] + (a * im[c]) * wgt;
            aug[c] = aug[c] + (g.m_oa * ag[c]) * wgt;
        }
        mask_sum = mask_max;
    }
    const float rest = 1.0f - mask_sum;
    const float mf = (float)m_beta;
    uint8_t px[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float o = orig[c] + aug[c];
        o = (float)((double)o + ((1.0 - m_beta) * (double)img[q + c]) * (double)rest);
        o = o + (mf * ag[c]) * rest;
        o = fminf(fmaxf(o, 0.f), 255.f);
        px[c] = (uint8_t)o;
    }
    if (out_u8) { out_u8[q] = px[0]; out_u8[q + 1] = px[1]; out_u8[q + 2] = px[2]; }
    if (out_norm) write_norm<T>(out_norm, Wp, x, y, px, nrm);
}

template <typename T>
__global__ __launch_bounds__(256) void normalize_kernel(const uint8_t* __restrict__ img, int H, int W,
                                                        NormArgs nrm, T* __restrict__ out, int Hp, int Wp) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= (long)Hp * Wp) return;
    const int y = (int)(p / Wp), x = (int)(p - (long)y * Wp);
    if (y >= H || x >= W) {
        for (int c = 0; c < 3; ++c) store_norm<T>(out, (size_t)p * 3 + c, 0.f);
        return;
    }
    const size_t q = ((size_t)y * W + x) * 3;
    const uint8_t px[3] = {img[q], img[q + 1], img[q + 2]};
    write_norm<T>(out, Wp, x, y, px, nrm);
}

// ------------------------------------------------------------------------------------------------ saliency
// StaticSaliencySpectralResidual on one box crop per block: gray -> 64x64 -> FFT2 -> log-amplitude minus its
// 3x3 mean -> inverse FFT -> |.| -> 5x5 Gaussian (sigma 8) -> square -> /max -> bilinear to the crop size ->
// mean of uint8(map * 255).  All of it in LDS (64x64 complex fp64 = 64 KiB).
constexpr int SN = 64;

constexpr int SLD = SN + 1;     // row stride of the 64 x 64 complex image in LDS (doubles): odd, so a lane-per-row access
                                // pattern (address = lane * 65 + i) spreads over all banks like lane-per-column does

// 64 independent 64-point radix-2 DIT FFTs (rows: cols = false, or columns) by all 256 threads: four threads per line,
// 8 butterflies each per stage, twiddles exp(-+ 2 pi i k / 64) from a table (tw_re / tw_im, k < 32, forward sign).
__device__ void fft64_lines(double* re, double* im, const double* tw_re, const double* tw_im, bool cols, bool inverse) {
    const int line = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int base = cols ? line : line * SLD, stride = cols ? SLD : 1;
    for (int u = 0; u < 16; ++u) {                       // bit reversal: the owner of the smaller index swaps the pair
        const int i = part * 16 + u;
        int j = 0;
        for (int bb = 0; bb < 6; ++bb) j |= ((i >> bb) & 1) << (5 - bb);
        if (j > i) {
            double t = re[base + i * stride]; re[base + i * stride] = re[base + j * stride]; re[base + j * stride] = t;
            t = im[base + i * stride]; im[base + i * stride] = im[base + j * stride]; im[base + j * stride] = t;
        }
    }
    __syncthreads();
    for (int len = 2; len <= SN; len <<= 1) {
        const int half = len >> 1, tstep = SN / len;
        for (int u = 0; u < 8; ++u) {
            const int bfly = part * 8 + u;
            const int k = bfly & (half - 1), s0 = (bfly / half) * len;
            const double wr = tw_re[k * tstep], wi = inverse ? -tw_im[k * tstep] : tw_im[k * tstep];
            const int i0 = base + (s0 + k) * stride, i1 = base + (s0 + k + half) * stride;
            const double xr = re[i1] * wr - im[i1] * wi, xi = re[i1] * wi + im[i1] * wr;
            re[i1] = re[i0] - xr; im[i1] = im[i0] - xi;
            re[i0] = re[i0] + xr; im[i0] = im[i0] + xi;
        }
        __syncthreads();
    }
}

__device__ __forceinline__ void lin_axis(int d, int n_src, int n_dst, int& s0, int& s1, double& f) {
    const double scale = (double)n_src / (double)n_dst;
    double ff = ((double)d + 0.5) * scale - 0.5;
    long s = (long)floor(ff);
    ff = ff - (double)s;
    if (s < 0) { ff = 0.0; s = 0; }
    if (s >= n_src - 1) { ff = 0.0; s = n_src - 1; }
    s0 = (int)s; s1 = s + 1 < n_src ? (int)s + 1 : n_src - 1; f = ff;
}

// (box_img / img_stride: the boxes of SEVERAL images of one [N,H,W,3] batch in one launch - box b lies in image box_img[b])
__global__ __launch_bounds__(256) void saliency_kernel(const uint8_t* __restrict__ img0, int H, int W,
                                                       const int* __restrict__ boxes, int min_side,
                                                       float* __restrict__ sal_maps, const int* __restrict__ box_img,
                                                       long long img_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* re = reinterpret_cast<double*>(smem);            // [64][65]
    double* im = re + SN * SLD;                               // [64][65]
    __shared__ double red[16];
    __shared__ double s_max;
    __shared__ double tw_re[32], tw_im[32];
    const int b = blockIdx.x, tid = threadIdx.x;
    const uint8_t* __restrict__ img = img0 + (box_img ? (size_t)box_img[b] * (size_t)img_stride : (size_t)0);
    const int x1 = boxes[4 * b], y1 = boxes[4 * b + 1], x2 = boxes[4 * b + 2], y2 = boxes[4 * b + 3];
    const int w = x2 - x1, h = y2 - y1;
    if (w < min_side || h < min_side) return;      // score -1 (saliency_mean_kernel)
    if (tid < 32) {
        double sn, cs;
        sincos(-2.0 * 3.14159265358979323846 * (double)tid / 64.0, &sn, &cs);
        tw_re[tid] = cs; tw_im[tid] = sn;
    }
    // gray (cv::cvtColor BGR2GRAY fixed point) + bilinear to 64x64, rounded to uint8
    for (int i = tid; i < SN * SN; i += 256) {
        const int dy = i / SN, dx = i - dy * SN;
        int ya, yb, xa, xb; double fy, fx;
        lin_axis(dy, h, SN, ya, yb, fy);
        lin_axis(dx, w, SN, xa, xb, fx);
        auto gray = [&](int yy, int xx) -> double {
            const uint8_t* p = img + ((size_t)(y1 + yy) * W + (x1 + xx)) * 3;
            return (double)((p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + (1 << 13)) >> 14);
        };
        const double top = gray(ya, xa) * (1.0 - fx) + gray(ya, xb) * fx;
        const double bot = gray(yb, xa) * (1.0 - fx) + gray(yb, xb) * fx;
        double v = floor(top * (1.0 - fy) + bot * fy + 0.5);
        v = fmin(fmax(v, 0.0), 255.0);
        re[dy * SLD + dx] = v; im[dy * SLD + dx] = 0.0;
    }
    __syncthreads();
    fft64_lines(re, im, tw_re, tw_im, false, false);
    fft64_lines(re, im, tw_re, tw_im, true, false);
    // log amplitude -> re; the spectrum's unit phasors (cos, sin of its angle = re / |.|, im / |.|) stay in registers: thread
    // tid owns the points tid + 256 k in every pass.  (Rounds 1-5 stored atan2(im, re) and took sincos of it again two passes
    // later: three double-precision transcendentals per point for a division - the kernel was bound by them, 100 us per crop.)
    double uc[16], us[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int i = tid + k * 256;
        const int p = (i >> 6) * SLD + (i & 63);
        const double a = re[p], b_ = im[p];
        const double mag = sqrt(a * a + b_ * b_);
        uc[k] = a / mag;                        // (mag == 0: NaN here, and log(0) = -inf makes the whole map NaN anyway - as before)
        us[k] = b_ / mag;
        re[p] = log(mag);
    }
    __syncthreads();
    // residual = exp(L - boxblur3(L)) ; needs L intact while reading neighbours -> stage result in registers
    double resid[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int i = tid + k * 256;
        const int y = i / SN, x = i - y * SN;
        double s = 0.0;
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) s += re[reflect101(y + dy, SN) * SLD + reflect101(x + dx, SN)];
        resid[k] = exp(re[y * SLD + x] - s / 9.0);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int i = tid + k * 256;
        const int p = (i >> 6) * SLD + (i & 63);
        re[p] = resid[k] * uc[k];
        im[p] = resid[k] * us[k];
    }
    __syncthreads();
    fft64_lines(re, im, tw_re, tw_im, false, true);
    fft64_lines(re, im, tw_re, tw_im, true, true);
    for (int i = tid; i < SN * SN; i += 256) {
        const int p = (i >> 6) * SLD + (i & 63);
        re[p] = sqrt(re[p] * re[p] + im[p] * im[p]);
    }
    __syncthreads();
    // 5x5 Gaussian sigma 8 (separable, reflect101): horizontal into im, vertical back into re
    double g[5];
    {
        double s = 0.0;
        for (int i = 0; i < 5; ++i) { g[i] = exp(-0.5 * (double)((i - 2) * (i - 2)) / 64.0); s += g[i]; }
        for (int i = 0; i < 5; ++i) g[i] /= s;
    }
    for (int i = tid; i < SN * SN; i += 256) {
        const int y = i / SN, x = i - y * SN;
        double s = 0.0;
        for (int k = 0; k < 5; ++k) s += g[k] * re[y * SLD + reflect101(x + k - 2, SN)];
        im[y * SLD + x] = s;
    }
    __syncthreads();
    double lmax = 0.0;
    for (int i = tid; i < SN * SN; i += 256) {
        const int y = i / SN, x = i - y * SN;
        double s = 0.0;
        for (int k = 0; k < 5; ++k) s += g[k] * im[reflect101(y + k - 2, SN) * SLD + x];
        s = s * s;
        re[y * SLD + x] = s;
        lmax = fmax(lmax, s);
    }
    for (int o = 32; o > 0; o >>= 1) lmax = fmax(lmax, __shfl_xor(lmax, o, 64));
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = lmax;
    __syncthreads();
    if (tid == 0) s_max = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    __syncthreads();
    // the 64x64 saliency map leaves the workgroup here: the resize to w x h and the mean run on many workgroups per box
    float* sal_out = sal_maps + (size_t)b * SN * SN;
    for (int i = tid; i < SN * SN; i += 256) sal_out[i] = (float)(re[(i >> 6) * SLD + (i & 63)] / s_max);
}

// bilinear (float32, horizontal then vertical) of the 64x64 map to w x h; total of uint8(v * 255) per box.  The
// addends are integers, so the block partials and the atomic total are exact in any order.
__global__ __launch_bounds__(256) void saliency_sum_kernel(const float* __restrict__ sal_maps,
                                                           const int* __restrict__ boxes, int min_side,
                                                           unsigned long long* __restrict__ totals) {
    __shared__ float sal[SN * SN];
    __shared__ unsigned long long red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int w = boxes[4 * b + 2] - boxes[4 * b], h = boxes[4 * b + 3] - boxes[4 * b + 1];
    if (w < min_side || h < min_side) return;
    const long npx = (long)w * h;
    const long per = (npx + gridDim.y - 1) / gridDim.y;
    const long p0 = per * blockIdx.y, p1 = p0 + per < npx ? p0 + per : npx;
    if (p0 >= npx) return;
    for (int i = tid; i < SN * SN; i += 256) sal[i] = sal_maps[(size_t)b * SN * SN + i];
    __syncthreads();
    unsigned long long acc = 0;
    for (long i = p0 + tid; i < p1; i += 256) {
        const int dy = (int)(i / w), dx = (int)(i - (long)dy * w);
        int ya, yb, xa, xb; double fyd, fxd;
        lin_axis(dy, SN, h, ya, yb, fyd);
        lin_axis(dx, SN, w, xa, xb, fxd);
        const float fx = (float)fxd, fy = (float)fyd;
        const float ax0 = 1.0f - fx, ay0 = 1.0f - fy;
        const float r0 = sal[ya * SN + xa] * ax0 + sal[ya * SN + xb] * fx;
        const float r1 = sal[yb * SN + xa] * ax0 + sal[yb * SN + xb] * fx;
        const float v = r0 * ay0 + r1 * fy;
        acc += (unsigned long long)(uint8_t)(v * 255.0f);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) atomicAdd(&totals[b], red[0] + red[1] + red[2] + red[3]);
}

__global__ void saliency_mean_kernel(const unsigned long long* __restrict__ totals, const int* __restrict__ boxes,
                                     int n, int min_side, double* __restrict__ scores) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    const int w = boxes[4 * b + 2] - boxes[4 * b], h = boxes[4 * b + 3] - boxes[4 * b + 1];
    scores[b] = (w < min_side || h < min_side) ? -1.0 : (double)totals[b] / (double)((long)w * h);
}

__global__ __launch_bounds__(256) void gray_sum_kernel(const uint8_t* __restrict__ img, long npix,
                                                       unsigned long long* __restrict__ out) {
    // sum of PIL's "L" conversion over the image (ImageStat mean for ImageEnhance.Contrast)
    unsigned long long s = 0;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256)
        s += (unsigned long long)((img[p * 3] * 19595 + img[p * 3 + 1] * 38470 + img[p * 3 + 2] * 7471 + 0x8000) >> 16);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    __shared__ unsigned long long red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

int grid1d(long n, int block) { return (int)((n + block - 1) / block); }

}  // namespace

extern "C" {

int oadg_oamix_box_profiles(const int* qbox, const double* sigma, int n, int H, int W, int ratio, float* My,
                            float* Mx, void* stream) {
    if (n == 0) return OADG_OK;
    if (!qbox || !sigma || !My || !Mx || n < 0 || H < 1 || W < 1 || ratio < 1) return OADG_EARG;
    if (H / ratio > PROF_MAXQ || W / ratio > PROF_MAXQ) return OADG_EARG;
    hipLaunchKernelGGL(box_profiles_kernel, dim3(n, 2), dim3(256), 0, (hipStream_t)stream, qbox, sigma, H, W,
                       ratio, My, Mx);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

int oadg_oamix_fg_union(const float* My, const float* Mx, int n, int H, int W, float* union_f,
                        uint8_t* union_u8, void* stream) {
    if (!union_f || !union_u8 || n < 0 || (n > 0 && (!My || !Mx))) return OADG_EARG;
    hipLaunchKernelGGL(fg_union_kernel, dim3(grid1d((long)H * W, 256)), dim3(256), 0, (hipStream_t)stream, My,
                       Mx, n, H, W, union_f, union_u8);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

// as oadg_oamix_fg_union, given the masks' support rects on the device (int32 [n][4]: x0, y0, w, h; w <= 0 = empty mask):
// work proportional to the rect areas instead of n x H x W.  Byte-identical outputs.
int oadg_oamix_fg_union_rects(const float* My, const float* Mx, const int* rects_dev, int n, int H, int W, float* union_f,
                              uint8_t* union_u8, void* stream) {
    if (!union_f || !union_u8 || n < 0 || (n > 0 && (!My || !Mx || !rects_dev))) return OADG_EARG;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(union_f, 0, (size_t)H * W * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    if (n > 0) {
        hipLaunchKernelGGL(fg_union_scatter_kernel, dim3(n, n >= 1024 ? 2 : 16), dim3(256), 0, st, My, Mx, rects_dev, H, W,
                           (unsigned*)union_f);
        OADG_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(fg_union_u8_kernel, dim3(grid1d((long)H * W, 256)), dim3(256), 0, st, (const float*)union_f,
                       (long)H * W, union_u8);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

size_t oadg_oamix_saliency_workspace_bytes(int n) {
    return n > 0 ? (size_t)n * (SN * SN * sizeof(float) + sizeof(unsigned long long)) : 0;
}

static int saliency_launch(const uint8_t* img, int H, int W, const int* boxes, int n, int min_side, double* scores,
                           void* workspace, size_t workspace_bytes, const int* box_img, long long img_stride, void* stream);

int oadg_oamix_saliency(const uint8_t* img, int H, int W, const int* boxes, int n, int min_side,
                        double* scores, void* workspace, size_t workspace_bytes, void* stream) {
    return saliency_launch(img, H, W, boxes, n, min_side, scores, workspace, workspace_bytes, nullptr, 0, stream);
}

// the boxes of several images of ONE [N,H,W,3] uint8 batch (image i at imgs + i * img_stride bytes) in one launch triple:
// box b belongs to image box_img[b] (device int32 [n]).  Same scores as one oadg_oamix_saliency call per image.
int oadg_oamix_saliency_batch(const uint8_t* imgs, long long img_stride, const int* box_img, int H, int W, const int* boxes,
                              int n, int min_side, double* scores, void* workspace, size_t workspace_bytes, void* stream) {
    if (n > 0 && (!box_img || img_stride < (long long)H * W * 3)) return OADG_EARG;
    return saliency_launch(imgs, H, W, boxes, n, min_side, scores, workspace, workspace_bytes, box_img, img_stride, stream);
}

static int saliency_launch(const uint8_t* img, int H, int W, const int* boxes, int n, int min_side, double* scores,
                           void* workspace, size_t workspace_bytes, const int* box_img, long long img_stride, void* stream) {
    if (n == 0) return OADG_OK;
    if (!img || !boxes || !scores || !workspace || n < 0) return OADG_EARG;
    if (workspace_bytes < oadg_oamix_saliency_workspace_bytes(n)) return OADG_ESIZE;
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* totals = (unsigned long long*)workspace;          // [n], then the [n][64*64] float maps
    float* maps = (float*)(totals + n);
    const size_t lds = (size_t)SN * SLD * (8 + 8);
    static bool attr_set = false;   // > 64 KiB of dynamic LDS must be opted into once (idempotent)
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)saliency_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    hipError_t e = hipMemsetAsync(totals, 0, (size_t)n * sizeof(unsigned long long), st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(saliency_kernel, dim3(n), dim3(256), lds, st, img, H, W, boxes, min_side, maps, box_img, img_stride);
    OADG_LAUNCH_CHECK();
    // crops are up to the whole image: ~64k pixels per workgroup, at most 32 workgroups per box
    int per_box = n >= 512 ? 1 : (n >= 64 ? 4 : 32);
    hipLaunchKernelGGL(saliency_sum_kernel, dim3(n, per_box), dim3(256), 0, st, (const float*)maps, boxes, min_side,
                       totals);
    OADG_LAUNCH_CHECK();
    hipLaunchKernelGGL(saliency_mean_kernel, dim3((n + 255) / 256), dim3(256), 0, st,
                       (const unsigned long long*)totals, boxes, n, min_side, scores);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

int oadg_oamix_hist(const uint8_t* img, long npix, int* hist, void* stream) {
    if (!img || !hist || npix < 0) return OADG_EARG;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(hist, 0, 768 * sizeof(int), st);
    if (e != hipSuccess) return (int)e;
    int g = grid1d(npix / 4 + 1, 256 * 16);
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(hist_kernel, dim3(g), dim3(256), 0, st, img, npix, hist);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

int oadg_oamix_gray_sum(const uint8_t* img, long npix, long long* sum, void* stream) {
    if (!img || !sum || npix < 0) return OADG_EARG;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(sum, 0, sizeof(long long), st);
    if (e != hipSuccess) return (int)e;
    int g = grid1d(npix, 256 * 8);
    if (g > 1024) g = 1024;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(gray_sum_kernel, dim3(g), dim3(256), 0, st, img, npix, (unsigned long long*)sum);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

int oadg_oamix_luts(const int* hist, uint8_t* luts, void* stream) {
    if (!hist || !luts) return OADG_EARG;
    hipLaunchKernelGGL(luts_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, hist, luts);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

int oadg_oamix_bbox_step(uint8_t* img, int H, int W, const double* minv_host, int rx0, int ry0, int rw, int rh,
                         const float* My_row, const float* Mx_row, uint8_t* scratch, void* stream) {
    if (!img || !minv_host || !My_row || !Mx_row || !scratch) return OADG_EARG;
    if (rw <= 0 || rh <= 0) return OADG_OK;
    if (rx0 < 0 || ry0 < 0 || rx0 + rw > W || ry0 + rh > H) return OADG_EARG;
    Warp wp;
    for (int i = 0; i < 6; ++i) wp.m[i] = minv_host[i];
    hipStream_t st = (hipStream_t)stream;
    const int g = grid1d((long)rw * rh, 256);
    hipLaunchKernelGGL(bbox_blend_kernel, dim3(g), dim3(256), 0, st, (const uint8_t*)img, H, W, wp, rx0, ry0, rw,
                       rh, My_row, Mx_row, scratch);
    OADG_LAUNCH_CHECK();
    hipLaunchKernelGGL(rect_copy_kernel, dim3(g), dim3(256), 0, st, img, W, rx0, ry0, rw, rh,
                       (const uint8_t*)scratch);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

int oadg_oamix_bbox_chain(uint8_t* img, int H, int W, const oadg_bbox_step* steps_dev, const int* tile_prefix_dev,
                          const int* level_first_host, int n_levels, const int* tile_prefix_host, const float* My,
                          const float* Mx, uint8_t* scratch, void* stream) {
    if (!img || !steps_dev || !tile_prefix_dev || !level_first_host || !tile_prefix_host || !My || !Mx || !scratch ||
        n_levels < 0)
        return OADG_EARG;
    hipStream_t st = (hipStream_t)stream;
    for (int l = 0; l < n_levels; ++l) {
        const int first = level_first_host[l], count = level_first_host[l + 1] - first;
        if (count <= 0) continue;
        const int tile_base = tile_prefix_host[first];
        const int tiles = tile_prefix_host[first + count] - tile_base;
        if (tiles <= 0) continue;
        hipLaunchKernelGGL(bbox_blend_multi_kernel, dim3(tiles), dim3(256), 0, st, (const uint8_t*)img, H, W,
                           steps_dev, tile_prefix_dev, first, count, tile_base, My, Mx, scratch);
        OADG_LAUNCH_CHECK();
        hipLaunchKernelGGL(rect_copy_multi_kernel, dim3(tiles), dim3(256), 0, st, img, W, steps_dev,
                           tile_prefix_dev, first, count, tile_base, (const uint8_t*)scratch);
        OADG_LAUNCH_CHECK();
    }
    return OADG_OK;
}

// n chains (one per image, each with the operands of oadg_oamix_bbox_chain) advanced level by level together: level l of
// all chains that have one = one launch pair.  Per image the same kernels' arithmetic in the same order as
// oadg_oamix_bbox_chain: byte-identical images.  The images, scratch buffers and tables must be pairwise distinct.
int oadg_oamix_bbox_chain_multi(const oadg_bbox_chain* chains_host, int n, void* stream) {
    if (!chains_host || n < 1) return OADG_EARG;
    int deepest = 0;
    for (int i = 0; i < n; ++i) {
        const oadg_bbox_chain& c = chains_host[i];
        if (!c.img || !c.steps_dev || !c.tile_prefix_dev || !c.level_first_host || !c.tile_prefix_host || !c.My || !c.Mx ||
            !c.scratch || c.n_levels < 0)
            return OADG_EARG;
        for (int j = 0; j < i; ++j)
            if (chains_host[j].img == c.img || chains_host[j].scratch == c.scratch) return OADG_EARG;
        if (c.n_levels > deepest) deepest = c.n_levels;
    }
    hipStream_t st = (hipStream_t)stream;
    for (int l = 0; l < deepest; ++l) {
        for (int i0 = 0; i0 < n; i0 += CHAIN_MAX) {
            ChainLevel a;
            a.n = 0;
            int blocks = 0;
            for (int i = i0; i < n && i < i0 + CHAIN_MAX; ++i) {
                const oadg_bbox_chain& c = chains_host[i];
                if (l >= c.n_levels) continue;
                const int first = c.level_first_host[l], count = c.level_first_host[l + 1] - first;
                if (count <= 0) continue;
                const int tile_base = c.tile_prefix_host[first];
                const int tiles = c.tile_prefix_host[first + count] - tile_base;
                if (tiles <= 0) continue;
                ChainImg& m = a.im[a.n++];
                m.img = c.img; m.steps = c.steps_dev; m.tile_prefix = c.tile_prefix_dev; m.My = c.My; m.Mx = c.Mx;
                m.scratch = c.scratch; m.H = c.H; m.W = c.W; m.first = first; m.count = count; m.tile_base = tile_base;
                m.block0 = blocks;
                blocks += tiles;
            }
            if (a.n == 0) continue;
            for (int k = a.n; k < CHAIN_MAX; ++k) a.im[k] = a.im[0];
            hipLaunchKernelGGL(bbox_blend_imgs_kernel, dim3(blocks), dim3(256), 0, st, a);
            OADG_LAUNCH_CHECK();
            hipLaunchKernelGGL(rect_copy_imgs_kernel, dim3(blocks), dim3(256), 0, st, a);
            OADG_LAUNCH_CHECK();
        }
    }
    return OADG_OK;
}

int oadg_oamix_compose(const uint8_t* src, uint8_t* dst, int H, int W, const oadg_region_op* ops_host,
                       const int* rects_host, int n_rects, const uint8_t* luts, const float* union_f,
                       const uint8_t* union_u8, float* acc, float acc_w, int acc_mode, void* stream) {
    if (!src || !dst || !ops_host || n_rects < 0 || n_rects > 2 || (n_rects && !rects_host)) return OADG_EARG;
    if (acc_mode && !acc) return OADG_EARG;
    ComposeArgs a;
    a.n_rects = n_rects;
    for (int k = 0; k < 3; ++k) {
        a.op[k] = ops_host[k];
        const int kd = a.op[k].kind;
        if ((kd == OADG_OP_LUT_AUTOCONTRAST || kd == OADG_OP_LUT_EQUALIZE) && !luts) return OADG_EARG;
        if ((kd == OADG_OP_IMAGE || kd == OADG_OP_ENH_CONTRAST) && !a.op[k].image) return OADG_EARG;
        if (kd == OADG_OP_BG_WARP && (!union_f || !union_u8)) return OADG_EARG;
    }
    for (int k = 0; k < 2; ++k)
        for (int j = 0; j < 4; ++j) a.rect[k][j] = (k < n_rects) ? rects_host[4 * k + j] : 0;
    const bool quad = (W & 3) == 0 && ((((uintptr_t)src) | ((uintptr_t)dst)) & 3) == 0 && (((uintptr_t)acc) & 15) == 0;
    int g = grid1d(quad ? (long)H * W / 4 : (long)H * W, 256);
    if (g > 8192) g = 8192;
    if (quad)
        hipLaunchKernelGGL(compose4_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, src, dst, H, W, a, luts,
                           union_f, union_u8, acc, acc_w, acc_mode);
    else
        hipLaunchKernelGGL(compose_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, src, dst, H, W, a, luts,
                           union_f, union_u8, acc, acc_w, acc_mode);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

static int fill_norm(NormArgs& n, const float* mean_host, const float* stdinv_host, int to_rgb) {
    if (!mean_host || !stdinv_host) return OADG_EARG;
    for (int c = 0; c < 3; ++c) { n.mean[c] = mean_host[c]; n.stdinv[c] = stdinv_host[c]; }
    n.to_rgb = to_rgb;
    return OADG_OK;
}

int oadg_oamix_final(const uint8_t* img, const float* acc, int H, int W, const oadg_mix_target* targets,
                     int n_targets, const float* My, const float* Mx, double m_beta, const float* mean_host,
                     const float* stdinv_host, int to_rgb, uint8_t* out_u8, void* out_norm, int out_dtype,
                     int Hp, int Wp, void* stream) {
    if (!img || !acc || n_targets < 0 || (n_targets && !targets) || Hp < H || Wp < W) return OADG_EARG;
    if (!out_u8 && !out_norm) return OADG_EARG;
    NormArgs nrm = {};
    if (out_norm && fill_norm(nrm, mean_host, stdinv_host, to_rgb)) return OADG_EARG;
    const int g = grid1d((long)Hp * Wp, 256);
    hipStream_t st = (hipStream_t)stream;
    if (out_dtype == 1)
        hipLaunchKernelGGL((final_mix_kernel<unsigned short, false>), dim3(g), dim3(256), 0, st, img, acc, H, W, targets,
                           n_targets, My, Mx, m_beta, nrm, out_u8, (unsigned short*)out_norm, Hp, Wp, nullptr, nullptr, 0, 0);
    else
        hipLaunchKernelGGL((final_mix_kernel<float, false>), dim3(g), dim3(256), 0, st, img, acc, H, W, targets,
                           n_targets, My, Mx, m_beta, nrm, out_u8, (float*)out_norm, Hp, Wp, nullptr, nullptr, 0, 0);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

// oadg_oamix_final for MANY targets: the targets are first binned into 32 x 32 pixel tiles (in target order) and a pixel
// only visits its tile's list.  fg_rects_dev: the support rects of the fg masks (int32 [n_fg][4]: x0, y0, w, h; as for
// oadg_oamix_fg_union_rects); workspace: oadg_oamix_final_tiles_workspace_bytes(H, W, n_targets) bytes.  Byte-identical to
// oadg_oamix_final (a target with a zero mask at a pixel adds +0 to every running sum of that pixel).
static int mix_cap(int n_targets) { return n_targets < 1024 ? (n_targets > 0 ? n_targets : 1) : 1024; }

size_t oadg_oamix_final_tiles_workspace_bytes(int H, int W, int n_targets) {
    if (H < 1 || W < 1 || n_targets < 0) return 0;
    const size_t tiles = (size_t)((W + MIX_TS - 1) / MIX_TS) * ((H + MIX_TS - 1) / MIX_TS);
    return tiles * sizeof(int) + tiles * (size_t)mix_cap(n_targets) * sizeof(unsigned short) + 16;
}

int oadg_oamix_final_tiles(const uint8_t* img, const float* acc, int H, int W, const oadg_mix_target* targets,
                           int n_targets, const int* fg_rects_dev, const float* My, const float* Mx, double m_beta,
                           const float* mean_host, const float* stdinv_host, int to_rgb, uint8_t* out_u8, void* out_norm,
                           int out_dtype, int Hp, int Wp, void* workspace, size_t workspace_bytes, void* stream) {
    if (!img || !acc || n_targets < 1 || n_targets > 65535 || !targets || !fg_rects_dev || Hp < H || Wp < W)
        return OADG_EARG;
    if (!out_u8 && !out_norm) return OADG_EARG;
    if (!workspace || workspace_bytes < oadg_oamix_final_tiles_workspace_bytes(H, W, n_targets)) return OADG_ESIZE;
    NormArgs nrm = {};
    if (out_norm && fill_norm(nrm, mean_host, stdinv_host, to_rgb)) return OADG_EARG;
    const int tiles_x = (W + MIX_TS - 1) / MIX_TS, tiles = tiles_x * ((H + MIX_TS - 1) / MIX_TS), cap = mix_cap(n_targets);
    int* counts = (int*)workspace;
    unsigned short* lists = (unsigned short*)(counts + tiles);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(mix_bins_kernel, dim3(tiles), dim3(256), 0, st, targets, n_targets, fg_rects_dev, tiles_x, cap, counts,
                       lists);
    OADG_LAUNCH_CHECK();
    const int g = grid1d((long)Hp * Wp, 256);
    if (out_dtype == 1)
        hipLaunchKernelGGL((final_mix_kernel<unsigned short, true>), dim3(g), dim3(256), 0, st, img, acc, H, W, targets,
                           n_targets, My, Mx, m_beta, nrm, out_u8, (unsigned short*)out_norm, Hp, Wp, (const int*)counts,
                           (const unsigned short*)lists, tiles_x, cap);
    else
        hipLaunchKernelGGL((final_mix_kernel<float, true>), dim3(g), dim3(256), 0, st, img, acc, H, W, targets,
                           n_targets, My, Mx, m_beta, nrm, out_u8, (float*)out_norm, Hp, Wp, (const int*)counts,
                           (const unsigned short*)lists, tiles_x, cap);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

int oadg_oamix_normalize(const uint8_t* img, int H, int W, const float* mean_host, const float* stdinv_host,
                         int to_rgb, void* out, int out_dtype, int Hp, int Wp, void* stream) {
    if (!img || !out || Hp < H || Wp < W) return OADG_EARG;
    NormArgs nrm;
    if (fill_norm(nrm, mean_host, stdinv_host, to_rgb)) return OADG_EARG;
    const int g = grid1d((long)Hp * Wp, 256);
    hipStream_t st = (hipStream_t)stream;
    if (out_dtype == 1)
        hipLaunchKernelGGL((normalize_kernel<unsigned short>), dim3(g), dim3(256), 0, st, img, H, W, nrm,
                           (unsigned short*)out, Hp, Wp);
    else
        hipLaunchKernelGGL((normalize_kernel<float>), dim3(g), dim3(256), 0, st, img, H, W, nrm, (float*)out, Hp,
                           Wp);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

}  // extern "C"
