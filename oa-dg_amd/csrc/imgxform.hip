// Geometric pipeline steps in front of OA-Mix, on uint8 HWC images resident in HBM:
//   Resize      mmdet/datasets/pipelines/transforms.py:210-239 -> mmcv.imrescale / imresize -> cv2.resize(INTER_LINEAR)
//   RandomFlip  transforms.py:423-470 -> mmcv.imflip
// cv2.resize for 8-bit images is restated from OpenCV's published algorithm (resize.cpp, HResizeLinear /
// VResizeLinear<uchar,int,short>): source coordinate fx = (dx + 0.5) * (src/dst) - 0.5 in float, edge clamp, 11-bit
// fixed-point coefficients (cvRound(w * 2048) as short), horizontal pass in int, vertical pass
//   dst = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.
// OpenCV is not installed here and the reference carries no vectors for it: PARITY UNPINNED (oracle/cvleaves.py
// resize_u8_cv2 is the same restatement on the CPU).  HBM-bound: reads ~4 source bytes per output byte through L2.
#include "common.h"
#include "../../include/oadg_hip.h"

namespace {

__device__ __forceinline__ int cv_round(float v) { return (int)rintf(v); }     // cvRound: round half to even

// source index and the two 11-bit coefficients of destination coordinate d
__device__ __forceinline__ void lin_coef(int d, double scale, int ssize, int* s0, int* s1, int* a0, int* a1) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
    *s0 = s;
    *s1 = s + 1 < ssize ? s + 1 : s;
    int c0 = cv_round((1.f - f) * 2048.f), c1 = cv_round(f * 2048.f);
    c0 = c0 > 32767 ? 32767 : c0;
    c1 = c1 > 32767 ? 32767 : c1;
    *a0 = c0;
    *a1 = c1;
}

__global__ __launch_bounds__(256) void resize_u8_kernel(const unsigned char* __restrict__ src, int H, int W, int C,
                                                        unsigned char* __restrict__ dst, int Hn, int Wn, double sx,
                                                        double sy) {
    const long total = (long)Hn * Wn;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int dx = (int)(i % Wn), dy = (int)(i / Wn);
        int x0, x1, ax0, ax1, y0, y1, by0, by1;
        lin_coef(dx, sx, W, &x0, &x1, &ax0, &ax1);
        {   // rows: no coefficient reset at the border, the two source rows are clamped individually (resize.cpp)
            float f = (float)((dy + 0.5) * sy - 0.5);
            const int sr = (int)floorf(f);
            f -= sr;
            y0 = sr < 0 ? 0 : (sr > H - 1 ? H - 1 : sr);
            y1 = sr + 1 < 0 ? 0 : (sr + 1 > H - 1 ? H - 1 : sr + 1);
            by0 = cv_round((1.f - f) * 2048.f);
            by1 = cv_round(f * 2048.f);
        }
        const unsigned char* r0 = src + (long)y0 * W * C;
        const unsigned char* r1 = src + (long)y1 * W * C;
        for (int c = 0; c < C; ++c) {
            const int h0 = r0[x0 * C + c] * ax0 + r0[x1 * C + c] * ax1;      // horizontal pass (int, scale 2048)
            const int h1 = r1[x0 * C + c] * ax0 + r1[x1 * C + c] * ax1;
            const int v = (((by0 * (h0 >> 4)) >> 16) + ((by1 * (h1 >> 4)) >> 16) + 2) >> 2;
            dst[i * C + c] = (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
}

__global__ __launch_bounds__(256) void flip_u8_kernel(const unsigned char* __restrict__ src, int H, int W, int C,
                                                      unsigned char* __restrict__ dst, int flip_x, int flip_y) {
    const long total = (long)H * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int x = (int)(i % W), y = (int)(i / W);
        const long s = (long)(flip_y ? H - 1 - y : y) * W + (flip_x ? W - 1 - x : x);
        for (int c = 0; c < C; ++c) dst[i * C + c] = src[s * C + c];
    }
}

}  // namespace

extern "C" int oadg_resize_bilinear_u8(const uint8_t* src, int H, int W, int C, uint8_t* dst, int Hn, int Wn,
                                       void* stream) {
    if (!src || !dst || H < 1 || W < 1 || Hn < 1 || Wn < 1 || C < 1 || C > 4) return OADG_EARG;
    const long total = (long)Hn * Wn;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(resize_u8_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, H, W, C, dst, Hn, Wn,
                       (double)W / Wn, (double)H / Hn);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

// direction: 1 horizontal, 2 vertical, 3 diagonal (mmcv.imflip)
extern "C" int oadg_flip_u8(const uint8_t* src, int H, int W, int C, uint8_t* dst, int direction, void* stream) {
    if (!src || !dst || src == dst || H < 1 || W < 1 || C < 1 || C > 4 || direction < 1 || direction > 3) return OADG_EARG;
    const long total = (long)H * W;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(flip_u8_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, H, W, C, dst,
                       direction & 1, (direction >> 1) & 1);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}
