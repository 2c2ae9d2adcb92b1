// Two sequential per-pixel loops of the robustness benchmark's corruptions, on the HOST (no device work).
//
// Replaces, for `Corrupt` (mmdet/datasets/pipelines/transforms.py:1277-1317 -> imagecorruptions.corrupt, third party,
// v1.1.2; tools/analysis_tools/test_robustness.py:222-235 lists the 19 names):
//   glass_blur  - the local pixel shuffle between its two Gaussian blurs: every pixel, bottom-right to top-left, swaps
//                 with a neighbour at a drawn offset; a swap sees the result of every earlier one, so the loop cannot
//                 be vectorised (the package compiles it with numba)
//   spatter     - cv2.distanceTransform(DIST_L2, 5): the two-pass 5 x 5 chamfer transform in 16-bit fixed point
//                 (weights 1 / 1.4 / 2.1969 for axial / diagonal / knight steps)
// Both are O(pixels) loops a Python interpreter needs seconds for at 1024 x 2048.  The draws of the shuffle stay on the
// numpy side (the caller passes them in), the arithmetic here is integer and exact.
#include <limits.h>
#include <stdlib.h>
#include <string.h>
#include "common.h"
#include "oadg_hip.h"

extern "C" {

int oadg_glass_shuffle_u8(uint8_t* img, int H, int W, int C, int delta, int iters, const int32_t* dxdy) {
    if (!img || !dxdy || H < 1 || W < 1 || C < 1 || C > 4 || delta < 1 || iters < 0) return OADG_EARG;
    const long nh = (long)H - 2 * delta, nw = (long)W - 2 * delta;
    if (nh <= 0 || nw <= 0) return OADG_OK;               // (an image smaller than the window: the loops are empty)
    const int32_t* d = dxdy;
    for (int it = 0; it < iters; ++it)
        for (int h = H - delta; h > delta; --h)
            for (int w = W - delta; w > delta; --w, d += 2) {
                const int dx = d[0], dy = d[1];
                if (dx < -delta || dx >= delta || dy < -delta || dy >= delta) return OADG_EARG;
                uint8_t* p = img + ((size_t)h * W + w) * C;
                uint8_t* q = img + ((size_t)(h + dy) * W + (w + dx)) * C;
                for (int c = 0; c < C; ++c) { const uint8_t t = p[c]; p[c] = q[c]; q[c] = t; }
            }
    return OADG_OK;
}

int oadg_chamfer_l2_5x5(const uint8_t* src, int H, int W, float* dist) {
    if (!src || !dist || H < 1 || W < 1) return OADG_EARG;
    constexpr int B = 2;                                   // border of the working image
    constexpr unsigned HV = 65536u, DIAG = 91750u, LONG = 143976u;   // round(1, 1.4, 2.1969 * 2^16)
    constexpr unsigned INIT = (unsigned)(INT_MAX >> 2);
    const long step = (long)W + 2 * B;
    unsigned* tmp = (unsigned*)malloc(sizeof(unsigned) * (size_t)step * (H + 2 * B));
    if (!tmp) return OADG_EIO;
    for (long i = 0; i < step * (H + 2 * B); ++i) tmp[i] = INIT;
    auto mn = [](unsigned a, unsigned b) { return a < b ? a : b; };
    for (int y = 0; y < H; ++y) {                          // forward: rows above and the pixel to the left
        unsigned* t = tmp + (y + B) * step + B;
        const uint8_t* s = src + (size_t)y * W;
        for (int x = 0; x < W; ++x) {
            if (!s[x]) { t[x] = 0; continue; }
            unsigned v = t[x - step * 2 - 1] + LONG;
            v = mn(v, t[x - step * 2 + 1] + LONG);
            v = mn(v, t[x - step - 2] + LONG);
            v = mn(v, t[x - step - 1] + DIAG);
            v = mn(v, t[x - step] + HV);
            v = mn(v, t[x - step + 1] + DIAG);
            v = mn(v, t[x - step + 2] + LONG);
            v = mn(v, t[x - 1] + HV);
            t[x] = v;
        }
    }
    for (int y = H - 1; y >= 0; --y) {                     // backward: rows below and the pixel to the right
        unsigned* t = tmp + (y + B) * step + B;
        float* o = dist + (size_t)y * W;
        for (int x = W - 1; x >= 0; --x) {
            unsigned v = t[x];
            if (v > HV) {
                v = mn(v, t[x + step * 2 + 1] + LONG);
                v = mn(v, t[x + step * 2 - 1] + LONG);
                v = mn(v, t[x + step + 2] + LONG);
                v = mn(v, t[x + step + 1] + DIAG);
                v = mn(v, t[x + step] + HV);
                v = mn(v, t[x + step - 1] + DIAG);
                v = mn(v, t[x + step - 2] + LONG);
                v = mn(v, t[x + 1] + HV);
                t[x] = v;
            }
            o[x] = (float)v * (1.0f / 65536.0f);
        }
    }
    free(tmp);
    return OADG_OK;
}

}  // extern "C"
