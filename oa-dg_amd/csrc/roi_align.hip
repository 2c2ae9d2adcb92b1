// RoIAlign (aligned, average pooling, adaptive sampling grid) over an FPN pyramid, forward and backward,
// for gfx950.  One launch covers every pyramid level: the level of each RoI is computed in-kernel.
//
// Replaces, behind the C ABI in include/oadg_hip.h:
//   mmcv.ops.RoIAlign fwd/bwd (mmcv-full, not vendored; semantics SURVEY.md A.3) as constructed at
//     mmdet/models/roi_heads/roi_extractors/base_roi_extractor.py:54-59 and called at
//     mmdet/models/roi_heads/roi_extractors/single_level_roi_extractor.py:110,134
//   SingleRoIExtractor.map_roi_levels  single_level_roi_extractor.py:36-55
//   the per-level gather/scatter loop   single_level_roi_extractor.py:112-146
//
// Layout: features are NHWC in HBM ([N, H_l, W_l, C], fp32 or bf16), output is [K, PH, PW, C] (the
// channels_last image of the reference's [K, C, PH, PW]).  A lane owns 4 consecutive channels, so every
// corner fetch of a wave is one fully coalesced 1 KiB (fp32) / 512 B (bf16) row segment; the four waves
// of a block split the PH*PW bins of one RoI.  Sample coordinates and bilinear weights are wave-uniform
// (they depend on blockIdx only) and live in SGPRs.  Accumulation is fp32 in the reference's order:
// iy-major, ix-minor, val = w1 v1 + w2 v2 + w3 v3 + w4 v4, out = sum / count.
//
// Algorithmic HBM bytes per RoI: PH*PW*C*sizeof(T) written + the unique input footprint
// (ceil(w_l)+1)(ceil(h_l)+1)*C*sizeof(T) read; corner re-reads (~4x) are served by L1/L2.
#include "common.h"

#define OADG_MAX_LEVELS 8

namespace {

struct Pyramid {
    const void* feat[OADG_MAX_LEVELS];
    float* dfeat[OADG_MAX_LEVELS];
    int H[OADG_MAX_LEVELS];
    int W[OADG_MAX_LEVELS];
    float scale[OADG_MAX_LEVELS];
    int levels;
    int N;
    int C;
    float finest_scale;
};

// single_level_roi_extractor.py:50-54
__device__ __forceinline__ int roi_level(const float* roi, const Pyramid& p) {
    if (p.levels == 1) return 0;
    const float s = sqrtf((roi[3] - roi[1]) * (roi[4] - roi[2]));
    float l = floorf(log2f(s / p.finest_scale + 1e-6f));
    l = fminf(fmaxf(l, 0.f), (float)(p.levels - 1));
    return (int)l;
}

struct Sample {
    int o1, o2, o3, o4;  // element offsets of the four corners / C
    float w1, w2, w3, w4;
    bool in;
};

__device__ __forceinline__ Sample make_sample(float y, float x, int H, int W) {
    Sample s;
    s.in = !(y < -1.0f || y > (float)H || x < -1.0f || x > (float)W);
    if (y <= 0.f) y = 0.f;
    if (x <= 0.f) x = 0.f;
    int yl = (int)y, xl = (int)x, yh, xh;
    if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else { yh = yl + 1; }
    if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else { xh = xl + 1; }
    const float ly = y - yl, lx = x - xl, hy = 1.0f - ly, hx = 1.0f - lx;
    s.w1 = hy * hx; s.w2 = hy * lx; s.w3 = ly * hx; s.w4 = ly * lx;
    s.o1 = yl * W + xl; s.o2 = yl * W + xh; s.o3 = yh * W + xl; s.o4 = yh * W + xh;
    return s;
}

__device__ __forceinline__ f32x4 load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 load4(const unsigned short* p) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    f32x4 r;
    r[0] = bf16_to_f32((unsigned short)v[0]); r[1] = bf16_to_f32((unsigned short)v[1]);
    r[2] = bf16_to_f32((unsigned short)v[2]); r[3] = bf16_to_f32((unsigned short)v[3]);
    return r;
}
__device__ __forceinline__ void store4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void store4(unsigned short* p, f32x4 v) {
    bf16x4 r;
    r[0] = (short)f32_to_bf16(v[0]); r[1] = (short)f32_to_bf16(v[1]);
    r[2] = (short)f32_to_bf16(v[2]); r[3] = (short)f32_to_bf16(v[3]);
    *reinterpret_cast<bf16x4*>(p) = r;
}

struct RoiGeom {
    float start_h, start_w, bin_h, bin_w;
    int grid_h, grid_w, lvl, batch;
    float count;
};

__device__ __forceinline__ RoiGeom roi_geom(const float* roi, const Pyramid& p, int PH, int PW,
                                            int sampling_ratio, int aligned) {
    RoiGeom g;
    g.lvl = roi_level(roi, p);
    g.batch = (int)roi[0];
    const float sc = p.scale[g.lvl];
    const float off = aligned ? 0.5f : 0.0f;
    g.start_w = roi[1] * sc - off;
    g.start_h = roi[2] * sc - off;
    const float end_w = roi[3] * sc - off, end_h = roi[4] * sc - off;
    float rw = end_w - g.start_w, rh = end_h - g.start_h;
    if (!aligned) { rw = fmaxf(rw, 1.f); rh = fmaxf(rh, 1.f); }
    g.bin_h = rh / (float)PH;
    g.bin_w = rw / (float)PW;
    g.grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)PH);
    g.grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)PW);
    g.count = (float)max(g.grid_h * g.grid_w, 1);
    return g;
}

template <typename T>
__global__ __launch_bounds__(256) void roi_align_fwd_kernel(Pyramid p, const float* __restrict__ rois,
                                                            int K, int PH, int PW, int sampling_ratio,
                                                            int aligned, T* __restrict__ out) {
    const int k = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* roi = rois + (size_t)k * 5;
    const RoiGeom g = roi_geom(roi, p, PH, PW, sampling_ratio, aligned);
    const int H = p.H[g.lvl], W = p.W[g.lvl], C = p.C;
    const bool bad_batch = g.batch < 0 || g.batch >= p.N;
    const T* base = reinterpret_cast<const T*>(p.feat[g.lvl]) + (size_t)(bad_batch ? 0 : g.batch) * H * W * C;
    for (int bin = wave; bin < PH * PW; bin += 4) {
        const int ph = bin / PW, pw = bin - ph * PW;
        for (int c0 = lane * 4; c0 < C; c0 += 256) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (!bad_batch) {
                for (int iy = 0; iy < g.grid_h; ++iy) {
                    const float y = g.start_h + ph * g.bin_h + (iy + 0.5f) * g.bin_h / (float)g.grid_h;
                    for (int ix = 0; ix < g.grid_w; ++ix) {
                        const float x = g.start_w + pw * g.bin_w + (ix + 0.5f) * g.bin_w / (float)g.grid_w;
                        const Sample s = make_sample(y, x, H, W);
                        if (!s.in) continue;
                        const f32x4 v1 = load4(base + (size_t)s.o1 * C + c0);
                        const f32x4 v2 = load4(base + (size_t)s.o2 * C + c0);
                        const f32x4 v3 = load4(base + (size_t)s.o3 * C + c0);
                        const f32x4 v4 = load4(base + (size_t)s.o4 * C + c0);
                        acc += s.w1 * v1 + s.w2 * v2 + s.w3 * v3 + s.w4 * v4;
                    }
                }
            }
            store4(out + ((size_t)k * PH * PW + bin) * C + c0, acc / g.count);
        }
    }
}

__device__ __forceinline__ float load1(const float* p) { return *p; }
__device__ __forceinline__ float load1(const unsigned short* p) { return bf16_to_f32(*p); }

// Backward: scatter g/count to the four corners of every sample (atomic fp32 adds into the level's gradient
// map).  Lane l owns channels l, l+64, l+128, l+192 of a 256-channel group, so each wave-wide atomic covers
// 256 contiguous bytes (two cache lines) of one pixel instead of 64 words strided over 1 KiB.
template <typename T>
__global__ __launch_bounds__(256) void roi_align_bwd_kernel(Pyramid p, const float* __restrict__ rois,
                                                            int K, int PH, int PW, int sampling_ratio,
                                                            int aligned, const T* __restrict__ gout) {
    const int k = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* roi = rois + (size_t)k * 5;
    const RoiGeom g = roi_geom(roi, p, PH, PW, sampling_ratio, aligned);
    const int H = p.H[g.lvl], W = p.W[g.lvl], C = p.C;
    if (g.batch < 0 || g.batch >= p.N) return;
    float* base = p.dfeat[g.lvl] + (size_t)g.batch * H * W * C;
    for (int bin = wave; bin < PH * PW; bin += 4) {
        const int ph = bin / PW, pw = bin - ph * PW;
        for (int cb = 0; cb < C; cb += 256) {
            float gv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = cb + lane + 64 * j;
                gv[j] = c < C ? load1(gout + ((size_t)k * PH * PW + bin) * C + c) / g.count : 0.f;
            }
            for (int iy = 0; iy < g.grid_h; ++iy) {
                const float y = g.start_h + ph * g.bin_h + (iy + 0.5f) * g.bin_h / (float)g.grid_h;
                for (int ix = 0; ix < g.grid_w; ++ix) {
                    const float x = g.start_w + pw * g.bin_w + (ix + 0.5f) * g.bin_w / (float)g.grid_w;
                    const Sample s = make_sample(y, x, H, W);
                    if (!s.in) continue;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int c = cb + lane + 64 * j;
                        if (c >= C) continue;
                        unsafeAtomicAdd(base + (size_t)s.o1 * C + c, gv[j] * s.w1);
                        unsafeAtomicAdd(base + (size_t)s.o2 * C + c, gv[j] * s.w2);
                        unsafeAtomicAdd(base + (size_t)s.o3 * C + c, gv[j] * s.w3);
                        unsafeAtomicAdd(base + (size_t)s.o4 * C + c, gv[j] * s.w4);
                    }
                }
            }
        }
    }
}

int fill_pyramid(Pyramid& p, const void* const* feats, float* const* dfeats, const int* heights,
                 const int* widths, const float* scales, int levels, int N, int C, float finest_scale) {
    if (levels < 1 || levels > OADG_MAX_LEVELS || N < 1 || C < 4 || (C & 3)) return OADG_EARG;
    p.levels = levels; p.N = N; p.C = C; p.finest_scale = finest_scale;
    for (int l = 0; l < levels; ++l) {
        if ((feats && !feats[l]) || (dfeats && !dfeats[l]) || heights[l] < 1 || widths[l] < 1) return OADG_EARG;
        p.feat[l] = feats ? feats[l] : nullptr;
        p.dfeat[l] = dfeats ? dfeats[l] : nullptr;
        p.H[l] = heights[l]; p.W[l] = widths[l]; p.scale[l] = scales[l];
    }
    return OADG_OK;
}

}  // namespace

extern "C" {

// dtype: 0 = fp32, 1 = bf16 (features and output share it)
int oadg_roi_align_fwd(const void* const* feats, const int* heights, const int* widths,
                       const float* scales, int levels, int N, int C, int dtype, float finest_scale,
                       const float* rois, int K, int PH, int PW, int sampling_ratio, int aligned,
                       void* out, void* stream) {
    if (!feats || !heights || !widths || !scales || !rois || !out) return OADG_EARG;
    if (K < 0 || PH < 1 || PW < 1 || (dtype != 0 && dtype != 1)) return OADG_EARG;
    Pyramid p;
    const int rc = fill_pyramid(p, feats, nullptr, heights, widths, scales, levels, N, C, finest_scale);
    if (rc) return rc;
    if (K == 0) return OADG_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0)
        hipLaunchKernelGGL((roi_align_fwd_kernel<float>), dim3(K), dim3(256), 0, st, p, rois, K, PH, PW,
                           sampling_ratio, aligned, (float*)out);
    else
        hipLaunchKernelGGL((roi_align_fwd_kernel<unsigned short>), dim3(K), dim3(256), 0, st, p, rois, K,
                           PH, PW, sampling_ratio, aligned, (unsigned short*)out);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

// Accumulates into the fp32 gradient maps dfeats[l] ([N, H_l, W_l, C], caller zero-initialises).
int oadg_roi_align_bwd(float* const* dfeats, const int* heights, const int* widths, const float* scales,
                       int levels, int N, int C, int dtype, float finest_scale, const float* rois, int K,
                       int PH, int PW, int sampling_ratio, int aligned, const void* grad_out,
                       void* stream) {
    if (!dfeats || !heights || !widths || !scales || !rois || !grad_out) return OADG_EARG;
    if (K < 0 || PH < 1 || PW < 1 || (dtype != 0 && dtype != 1)) return OADG_EARG;
    Pyramid p;
    const int rc = fill_pyramid(p, nullptr, dfeats, heights, widths, scales, levels, N, C, finest_scale);
    if (rc) return rc;
    if (K == 0) return OADG_OK;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0)
        hipLaunchKernelGGL((roi_align_bwd_kernel<float>), dim3(K), dim3(256), 0, st, p, rois, K, PH, PW,
                           sampling_ratio, aligned, (const float*)grad_out);
    else
        hipLaunchKernelGGL((roi_align_bwd_kernel<unsigned short>), dim3(K), dim3(256), 0, st, p, rois, K,
                           PH, PW, sampling_ratio, aligned, (const unsigned short*)grad_out);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

}  // extern "C"
