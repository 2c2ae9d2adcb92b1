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
#include <stdlib.h>

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
    const int* order;      // optional: workgroup b handles RoI order[b] (spatially sorted RoIs share cache lines)
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

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 load2(const float* p) { return *reinterpret_cast<const f32x2*>(p); }
__device__ __forceinline__ f32x2 load2(const unsigned short* p) {
    const unsigned u = *reinterpret_cast<const unsigned*>(p);
    return f32x2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
}
__device__ __forceinline__ void store2(float* p, f32x2 v) { *reinterpret_cast<f32x2*>(p) = v; }
__device__ __forceinline__ void store2(unsigned short* p, f32x2 v) {
    *reinterpret_cast<unsigned*>(p) = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
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
__device__ void roi_fwd_samples(const Pyramid& p, const RoiGeom& g, int k, int PH, int PW, T* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int H = p.H[g.lvl], W = p.W[g.lvl], C = p.C;
    const bool bad_batch = g.batch < 0 || g.batch >= p.N;
    const T* base = reinterpret_cast<const T*>(p.feat[g.lvl]) + (size_t)(bad_batch ? 0 : g.batch) * H * W * C;
    for (int bin = wave; bin < PH * PW; bin += 4) {
        const int ph = bin / PW, pw = bin - ph * PW;
        // (channel chunks of 256 on blockIdx.y: a 2048-channel C5 map - R101-DC5 - gives every RoI eight workgroups instead
        //  of an eight-trip loop around ~200 dependent corner loads; C <= 256: gridDim.y = 1, the loop runs once)
        for (int c0 = blockIdx.y * 256 + lane * 4; c0 < min(C, (int)(blockIdx.y + 1) * 256); c0 += 256) {
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

template <typename T>
__global__ __launch_bounds__(256) void roi_align_fwd_kernel(Pyramid p, const float* __restrict__ rois,
                                                            int K, int PH, int PW, int sampling_ratio,
                                                            int aligned, T* __restrict__ out) {
    const int k = p.order ? p.order[blockIdx.x] : (int)blockIdx.x;
    const RoiGeom g = roi_geom(rois + (size_t)k * 5, p, PH, PW, sampling_ratio, aligned);
    roi_fwd_samples<T>(p, g, k, PH, PW, out);
}

__device__ __forceinline__ float load1(const float* p) { return *p; }
__device__ __forceinline__ float load1(const unsigned short* p) { return bf16_to_f32(*p); }

// Backward.  The bilinear weight of a sample factorises, w(y, x) = wy(y) * wx(x), the samples of a bin form a
// product grid and the validity test / border clamps of a sample are per axis, so the total weight a bin puts on
// pixel (py, px) is WY[ph][py] * WX[pw][px] with two small 1-D tables (sum over iy, sum over ix).  The gradient
// of a pixel of the RoI's footprint is then a GATHER over the few bins whose spans contain it:
//     d feat[py][px][c] += (1/count) * sum_{ph, pw} WY[ph][py] WX[pw][px] g[ph][pw][c]
// - ONE fp32 atomic per footprint pixel and channel instead of 4 * grid_h * grid_w per bin (about 3x fewer for
// typical RoIs), still as dense wave-wide atomics: lane l owns channels l, l+64, l+128, l+192 of a 256-channel
// chunk, so each wave instruction covers 256 contiguous bytes of one pixel.  The RoI's gradient slab
// [PH*PW][256] is staged in LDS once per channel chunk.  RoIs whose bins span more than RB_SPAN pixels (or
// PH/PW > 8) take the per-sample scatter path.
constexpr int RB_SPAN = 96;
constexpr int RB_MAXP = 8;

template <typename T>
__device__ void roi_bwd_scatter(const Pyramid& p, const RoiGeom& g, int k, int PH, int PW,
                                const T* __restrict__ gout) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int H = p.H[g.lvl], W = p.W[g.lvl], C = p.C;
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

// 1-D weight table of one bin along one axis (the per-axis half of make_sample): returns false on overflow
__device__ bool axis_table(float start, float bin, int grid, int b, int size, float* w, int* off, int* cnt) {
    for (int j = 0; j < RB_SPAN; ++j) w[j] = 0.f;
    int o = -1, n = 0;
    bool ok = true;
    for (int i = 0; i < grid; ++i) {
        float v = start + b * bin + (i + 0.5f) * bin / (float)grid;
        if (v < -1.0f || v > (float)size) continue;
        if (v <= 0.f) v = 0.f;
        int lo = (int)v, hi;
        if (lo >= size - 1) { hi = lo = size - 1; v = (float)lo; } else { hi = lo + 1; }
        const float l = v - lo, h = 1.0f - l;
        if (o < 0) o = lo;
        if (hi - o >= RB_SPAN) { ok = false; break; }
        w[lo - o] += h;
        w[hi - o] += l;
        n = hi - o + 1;
    }
    *off = o < 0 ? 0 : o;
    *cnt = n;
    return ok;
}

// Forward by footprint rows (round 5).  The per-sample form above reads four corner rows per sample - 4 g^2 rows of C
// channels per bin, 49 bins - and is bound by the L1 / texture path (~3.8 GB of rows per launch at BASELINE config 2).  The
// bilinear weights factorise (see the backward pass): out[ph][pw] = (1 / count) sum_r WY[ph][r] * T_r[pw] with
// T_r[pw] = sum_q WX[pw][q] f[r][q].  A wave owns a bin row ph (waves 0-3: ph = w and w + 4): for every footprint row r of
// that bin row it reads each pixel of the RoI's footprint width ONCE (lane = 4 channels: 512-byte wave rows), adds it into
// the seven column accumulators with the pixel's seven column weights (one 32-byte LDS broadcast), and adds WY * T into
// the bin row's seven outputs: 7 (b + 2) (7 b + 1) rows per RoI instead of 196 g^2 (b = bin size in pixels, g = ceil(b)):
// 2 - 2.6x fewer.  fp32 sums in another order than the reference's (iy-major samples): within the 1e-4 bar.  RoIs whose
// bins span more than FR_SPAN pixels, or PH / PW > 8, take the per-sample form.
constexpr int FR_SPAN = 32;
constexpr int FR_WIDTH = RB_MAXP * FR_SPAN + 2;
constexpr int FR_INFLIGHT = 4;         // pixel loads in flight per wave in the column loop

__device__ bool axis_table_n(float start, float bin, int grid, int b, int size, float* w, int span, int* off, int* cnt) {
    for (int j = 0; j < span; ++j) w[j] = 0.f;
    int o = -1, n = 0;
    bool ok = true;
    for (int i = 0; i < grid; ++i) {
        float v = start + b * bin + (i + 0.5f) * bin / (float)grid;
        if (v < -1.0f || v > (float)size) continue;
        if (v <= 0.f) v = 0.f;
        int lo = (int)v, hi;
        if (lo >= size - 1) { hi = lo = size - 1; v = (float)lo; } else { hi = lo + 1; }
        const float l = v - lo, h = 1.0f - l;
        if (o < 0) o = lo;
        if (hi - o >= span) { ok = false; break; }
        w[lo - o] += h;
        w[hi - o] += l;
        n = hi - o + 1;
    }
    *off = o < 0 ? 0 : o;
    *cnt = n;
    return ok;
}

template <typename T>
__global__ __launch_bounds__(256) void roi_align_fwd_rows_kernel(Pyramid p, const float* __restrict__ rois,
                                                                 int K, int PH, int PW, int sampling_ratio,
                                                                 int aligned, T* __restrict__ out) {
    __shared__ float wy[RB_MAXP][FR_SPAN];
    __shared__ float wxq[FR_WIDTH + FR_INFLIGHT][8];             // [footprint column][bin column]: one 32-byte broadcast per pixel
    __shared__ float wxb[RB_MAXP][FR_SPAN];
    __shared__ int oy[RB_MAXP], ny[RB_MAXP], ox[RB_MAXP], nx[RB_MAXP];
    __shared__ int overflow;
    const int k = p.order ? p.order[blockIdx.x] : (int)blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const RoiGeom g = roi_geom(rois + (size_t)k * 5, p, PH, PW, sampling_ratio, aligned);
    const int H = p.H[g.lvl], W = p.W[g.lvl], C = p.C;
    const bool bad_batch = g.batch < 0 || g.batch >= p.N;
    if (PH > RB_MAXP || PW > RB_MAXP || PW > 7 || bad_batch) {       // (7 column accumulators in registers)
        roi_fwd_samples<T>(p, g, k, PH, PW, out);
        return;
    }
    if (tid == 0) overflow = 0;
    for (int i = tid; i < (FR_WIDTH + FR_INFLIGHT) * 8; i += 256) (&wxq[0][0])[i] = 0.f;
    __syncthreads();
    if (tid < PH) {
        if (!axis_table_n(g.start_h, g.bin_h, g.grid_h, tid, H, wy[tid], FR_SPAN, &oy[tid], &ny[tid])) overflow = 1;
    } else if (tid >= 64 && tid < 64 + PW) {
        const int b = tid - 64;
        if (!axis_table_n(g.start_w, g.bin_w, g.grid_w, b, W, wxb[b], FR_SPAN, &ox[b], &nx[b])) overflow = 1;
    }
    __syncthreads();
    int X0 = 1 << 30, X1 = -1;
    for (int b = 0; b < PW; ++b)
        if (nx[b] > 0) { X0 = min(X0, ox[b]); X1 = max(X1, ox[b] + nx[b]); }
    const int FW = X1 - X0;
    if (overflow || FW > FR_WIDTH) {
        roi_fwd_samples<T>(p, g, k, PH, PW, out);
        return;
    }
    if (tid >= 64 && tid < 64 + PW) {
        const int b = tid - 64;
        for (int j = 0; j < nx[b]; ++j) wxq[ox[b] - X0 + j][b] = wxb[b][j];
    }
    __syncthreads();
    const T* base = reinterpret_cast<const T*>(p.feat[g.lvl]) + (size_t)g.batch * H * W * C;
    const float inv = 1.0f / g.count;
    // a wave owns bin rows ph = w, w + 4; lane = 4 channels.  The column loop keeps FR_INFLIGHT pixel loads in flight (one load per
    // trip left the wave waiting a full L2 latency per pixel: ~900 cycles per trip at 16 waves per CU)
    for (int ph = wave; ph < PH; ph += 4) {
        const int r0 = oy[ph], nr = ny[ph];
        for (int c0 = blockIdx.y * 256 + lane * 4; c0 < min(C, (int)(blockIdx.y + 1) * 256); c0 += 256) {
            f32x4 acc[7];
#pragma unroll
            for (int b = 0; b < 7; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int jr = 0; jr < nr; ++jr) {
                const float wyv = wy[ph][jr];
                if (wyv == 0.f) continue;
                f32x4 t[7];
#pragma unroll
                for (int b = 0; b < 7; ++b) t[b] = f32x4{0.f, 0.f, 0.f, 0.f};
                const T* row = base + ((size_t)(r0 + jr) * W + X0) * C + c0;
                for (int q0 = 0; q0 < FW; q0 += FR_INFLIGHT) {
                    f32x4 v[FR_INFLIGHT];
#pragma unroll
                    for (int u = 0; u < FR_INFLIGHT; ++u)       // (columns past the footprint: re-read the last one, its weights are zero)
                        v[u] = load4(row + (size_t)min(q0 + u, FW - 1) * C);
#pragma unroll
                    for (int u = 0; u < FR_INFLIGHT; ++u) {
                        const int q = q0 + u;     // wxq has FR_WIDTH + FR_INFLIGHT rows: rows >= FW are zero
                        const f32x4 wa = *reinterpret_cast<const f32x4*>(&wxq[q][0]);
                        const f32x4 wb = *reinterpret_cast<const f32x4*>(&wxq[q][4]);
                        t[0] += wa[0] * v[u]; t[1] += wa[1] * v[u]; t[2] += wa[2] * v[u]; t[3] += wa[3] * v[u];
                        t[4] += wb[0] * v[u]; t[5] += wb[1] * v[u]; t[6] += wb[2] * v[u];
                    }
                }
#pragma unroll
                for (int b = 0; b < 7; ++b) acc[b] += wyv * t[b];
            }
#pragma unroll
            for (int b = 0; b < 7; ++b)
                if (b < PW) store4(out + ((size_t)k * PH * PW + ph * PW + b) * C + c0, acc[b] * inv);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void roi_align_bwd_kernel(Pyramid p, const float* __restrict__ rois,
                                                            int K, int PH, int PW, int sampling_ratio,
                                                            int aligned, const T* __restrict__ gout) {
    __shared__ float wy[RB_MAXP][RB_SPAN], wx[RB_MAXP][RB_SPAN];
    __shared__ int oy[RB_MAXP], ny[RB_MAXP], ox[RB_MAXP], nx[RB_MAXP];
    __shared__ int overflow;
    __shared__ T slab[RB_MAXP * RB_MAXP * 256];
    const int k = p.order ? p.order[blockIdx.x] : (int)blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* roi = rois + (size_t)k * 5;
    const RoiGeom g = roi_geom(roi, p, PH, PW, sampling_ratio, aligned);
    const int H = p.H[g.lvl], W = p.W[g.lvl], C = p.C;
    if (g.batch < 0 || g.batch >= p.N) return;
    if (PH > RB_MAXP || PW > RB_MAXP) {
        roi_bwd_scatter<T>(p, g, k, PH, PW, gout);
        return;
    }
    if (tid == 0) overflow = 0;
    __syncthreads();
    if (tid < PH) {
        if (!axis_table(g.start_h, g.bin_h, g.grid_h, tid, H, wy[tid], &oy[tid], &ny[tid])) overflow = 1;
    } else if (tid >= 64 && tid < 64 + PW) {
        const int b = tid - 64;
        if (!axis_table(g.start_w, g.bin_w, g.grid_w, b, W, wx[b], &ox[b], &nx[b])) overflow = 1;
    }
    __syncthreads();
    if (overflow) {
        roi_bwd_scatter<T>(p, g, k, PH, PW, gout);
        return;
    }
    int Y0 = 1 << 30, Y1 = -1, X0 = 1 << 30, X1 = -1;
    for (int b = 0; b < PH; ++b)
        if (ny[b] > 0) { Y0 = min(Y0, oy[b]); Y1 = max(Y1, oy[b] + ny[b]); }
    for (int b = 0; b < PW; ++b)
        if (nx[b] > 0) { X0 = min(X0, ox[b]); X1 = max(X1, ox[b] + nx[b]); }
    if (Y1 < 0 || X1 < 0) return;              // no sample inside the map
    const int FW = X1 - X0, npix = (Y1 - Y0) * FW;
    const float inv = 1.0f / g.count;
    float* base = p.dfeat[g.lvl] + (size_t)g.batch * H * W * C;
    const int bins = PH * PW;
    for (int cb = 0; cb < C; cb += 256) {
        const int cw = min(256, C - cb);       // channels of this chunk (a multiple of 4)
        __syncthreads();
        for (int q = tid; q < bins * 64; q += 256) {          // 4 channels per piece
            const int bin = q >> 6, c4 = (q & 63) << 2;
            if (c4 < cw) {
                const T* src = gout + ((size_t)k * bins + bin) * C + cb + c4;
                T* dst = slab + bin * 256 + c4;
                dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
            }
        }
        __syncthreads();
        for (int pi = wave; pi < npix; pi += 4) {
            const int py = Y0 + pi / FW, px = X0 + pi % FW;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            bool any = false;
            for (int ph = 0; ph < PH; ++ph) {
                const int jy = py - oy[ph];
                if (jy < 0 || jy >= ny[ph]) continue;
                const float wyv = wy[ph][jy];
                if (wyv == 0.f) continue;
                for (int pw = 0; pw < PW; ++pw) {
                    const int jx = px - ox[pw];
                    if (jx < 0 || jx >= nx[pw]) continue;
                    const float wv = wyv * wx[pw][jx];
                    if (wv == 0.f) continue;
                    any = true;
                    const T* row = slab + (ph * PW + pw) * 256;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (lane + 64 * j < cw) acc[j] += wv * load1(row + lane + 64 * j);
                }
            }
            if (!any) continue;
            float* dst = base + ((size_t)py * W + px) * C + cb;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (lane + 64 * j < cw) unsafeAtomicAdd(dst + lane + 64 * j, acc[j] * inv);
        }
    }
}


// ------------------------------------------------------------------------------------------------ backward by tiles
// The same gradient as roi_align_bwd_kernel, organised by OUTPUT: one workgroup owns an 8 x 8 pixel tile of one image on
// one pyramid level, walks the RoIs of that (level, image) - `order` lists the RoIs sorted by (level, image, ...) and
// `range` holds the first position of every (level, image) group - keeps the tile's 64 x C gradient in registers (fp32)
// and writes it ONCE as bf16: no fp32 gradient maps (1.4 GB at 1024 x 2048, bs 4 x 2 views), no zero fill, no atomics,
// no cast pass, and a summation order fixed by `order` (deterministic).  Per overlapping RoI the bins' 1-D weights on
// the tile's 8 rows / 8 columns are rebuilt from the samples (the per-axis half of make_sample), the RoI's gradient
// slab [PH*PW][256] is staged in LDS, and each pixel gathers  sum_{ph,pw} WY[ph][y] WX[pw][x] g[ph][pw][c] / count.
constexpr int TILE = 8;

struct TileGrid {
    int first[OADG_MAX_LEVELS + 1];     // first workgroup of each level
    int tx[OADG_MAX_LEVELS], ty[OADG_MAX_LEVELS];
};

__device__ void axis_tile_weights(float start, float bin, int grid, int b, int size, int t0, float* w) {
    for (int j = 0; j < TILE; ++j) w[j] = 0.f;
    for (int i = 0; i < grid; ++i) {
        float v = start + b * bin + (i + 0.5f) * bin / (float)grid;
        if (v < -1.0f || v > (float)size) continue;
        if (v <= 0.f) v = 0.f;
        int lo = (int)v, hi;
        if (lo >= size - 1) { hi = lo = size - 1; v = (float)lo; } else { hi = lo + 1; }
        const float l = v - lo, h = 1.0f - l;
        if (lo >= t0 && lo < t0 + TILE) w[lo - t0] += h;
        if (hi >= t0 && hi < t0 + TILE) w[hi - t0] += l;
    }
}

// Round 3.  The first tile kernel (round 2) took ~3 ms per launch at BASELINE config 2 and stayed an opt-in.  Measured with
// ablations (tools/probe/roi_bwd_probe.py):
//  * 0.8 ms was the per-tile scan - every one of the 21,760 tile workgroups evaluated roi_geom (log2f, sqrtf, divisions)
//    for the ~440 RoIs of its (level, image) group;
//  * even an EMPTY kernel over 21,760 workgroups of 512 threads took 0.66 ms (workgroup launch + argument loads + three
//    barriers: ~10 us of lifetime each, 42 rounds over the chip);
//  * the pair loop ran at ~8 us per (tile, RoI) pair: per pixel a chain of dependent LDS round trips (bin bounds, wy,
//    wx, slab value) with trip counts of 1-3, the slab staged behind a workgroup barrier, the pair's tables built by one
//    wave while seven waited.
// Now:
//  * roi_tile_box_kernel writes the tile rectangle of every RoI once (int4, in processing order): the scan of a tile is
//    one coalesced 16-byte load and four compares per RoI;
//  * a workgroup walks TPW = 8 consecutive tiles of one (level, image): 2,720 workgroups instead of 21,760;
//  * the row / column weight tables of up to 32 pairs are built at once by all 512 threads (16 lanes per pair);
//  * separable accumulation: T[x] = sum_pw WX[pw][x] g[ph][pw][c], then acc[y][x] += (WY[ph][y] / count) T[x], on 4 x 4
//    blocks of the bins that reach the tile (bin rows / columns outside the tile are never touched); the weights sit
//    lane-indexed in two VGPRs and reach the multiplies through v_readlane (no memory latency, no per-pixel control
//    flow);
//  * no slab in LDS and no barrier inside the pair loop: lane c of a wave reads g[ph][pw][c] of its 64-channel chunk
//    straight from global memory (128-byte wave rows, L2 resident: a slab is used by ~12 tiles), the first block of the
//    NEXT pair in flight while the current pair is accumulated; the 8 waves (4 channel chunks x 2 column halves, 8 rows x
//    4 columns per lane in registers) run through the tile's pair list independently.
// Deterministic (the RoIs of a tile are visited in `order`, fixed fp32 summation order), one bf16 rounding at the end.
constexpr int TB_THREADS = 512;
constexpr int TPW = 8;               // tiles per workgroup (at most: fewer when the maps are small, see the launcher)
constexpr int HB = 32;               // pairs per table batch

struct TileTables {
    float wy[RB_MAXP][TILE], wx[RB_MAXP][TILE];         // [bin][tile row / column]; wy already divided by the sample count
    int plo, phi, qlo, qhi;                             // bin rows / columns with a non-zero weight somewhere on the tile
};

// tile rectangle [tx0, tx1] x [ty0, ty1] (inclusive, 8-pixel tiles of the RoI's level; conservative) and (level, image) of
// the RoI at processing position i: {tx0 | tx1 << 16, ty0 | ty1 << 16, level, image}; an empty rectangle has tx1 < tx0
__global__ void roi_tile_box_kernel(Pyramid p, const float* __restrict__ rois, int K, int PH, int PW, int sampling_ratio,
                                    int aligned, const int* __restrict__ order, int4* __restrict__ boxes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K) return;
    const int k = order[i];
    const RoiGeom g = roi_geom(rois + (size_t)k * 5, p, PH, PW, sampling_ratio, aligned);
    const int H = p.H[g.lvl], W = p.W[g.lvl];
    const float rh = g.bin_h * PH, rw = g.bin_w * PW;
    float ylo = floorf(g.start_h) - 1.f, yhi = ceilf(g.start_h + rh) + 1.f;
    float xlo = floorf(g.start_w) - 1.f, xhi = ceilf(g.start_w + rw) + 1.f;
    int4 b;
    if (g.batch < 0 || g.batch >= p.N || !(yhi >= 0.f) || !(xhi >= 0.f) || !(ylo < (float)H) || !(xlo < (float)W)) {
        b = make_int4(1, 1, g.lvl, -1);                // tx1 (0) < tx0 (1): touches nothing (NaN coordinates land here too)
    } else {
        ylo = fmaxf(ylo, 0.f); xlo = fmaxf(xlo, 0.f);
        yhi = fminf(yhi, (float)(H - 1)); xhi = fminf(xhi, (float)(W - 1));
        const int ty0 = (int)ylo / TILE, ty1 = (int)yhi / TILE, tx0 = (int)xlo / TILE, tx1 = (int)xhi / TILE;
        b = make_int4(tx0 | (tx1 << 16), ty0 | (ty1 << 16), g.lvl, g.batch);
    }
    boxes[i] = b;
}

__device__ __forceinline__ float lane_f(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

// Round 5: the tiles of a workgroup are processed as ONE batch.  Measured before (tools/probe/roi_bwd_probe.py: 36 k (tile, RoI)
// pairs on 21,760 tiles - 1.7 per tile, 14 per workgroup of 8 tiles; 128 VGPRs = 2 workgroups per CU): 117 us per workgroup
// = 14.6 us per TILE, almost none of it arithmetic - every tile paid its own chain of hit test -> 2 barriers -> tables of
// its 1-2 pairs (16 of 512 threads busy) -> barrier -> first slab load at full memory latency -> store.  Now the hit tests
// of all tiles happen at once (8 ballots, one prefix scan over the 64 (tile, wave) counts), the (tile, RoI) pairs of the
// whole workgroup form one list in tile-major position order - the SAME summation order per tile as before: bit-identical
// gradients - whose tables are built 32 pairs at a time by all threads, and every wave walks the list with the next pair's
// slab block in flight ACROSS tile boundaries, writing a tile's 8 x 4 pixels when the list moves on to the next tile.
// A (level, image) group of more than 512 RoIs takes the same walk tile by tile, 512 RoIs at a time.
__global__ __launch_bounds__(TB_THREADS, 4) void roi_align_bwd_tiles_kernel(Pyramid p, TileGrid tg, const float* __restrict__ rois,
                                                                         int PH, int PW, int sampling_ratio, int aligned,
                                                                         const unsigned short* __restrict__ gout,
                                                                         const int* __restrict__ order,
                                                                         const int* __restrict__ range,
                                                                         const int4* __restrict__ boxes, int tpw) {
    __shared__ TileTables tab[HB];
    __shared__ int hits[TB_THREADS * TPW];              // RoI | tile-of-the-workgroup << 24, tile-major, position order
    __shared__ int wcnt[TPW * 8], wbase[TPW * 8 + 1];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int chunk = wave & 3, xh = wave >> 2;         // channel chunk of 64, column half (tile columns 4 xh .. 4 xh + 3)
    int lvl = 0;
    while (lvl + 1 < p.levels && (int)blockIdx.x >= tg.first[lvl + 1]) ++lvl;
    const int H = p.H[lvl], W = p.W[lvl], C = p.C;
    const int txl = tg.tx[lvl];
    const int per_img = txl * tg.ty[lvl];
    const int wg_per_img = (per_img + tpw - 1) / tpw;
    const int local = blockIdx.x - tg.first[lvl];
    const int n = local / wg_per_img, t_first = (local - n * wg_per_img) * tpw;
    const int n_t = min(tpw, per_img - t_first);        // tiles of this workgroup
    const int r0 = range[lvl * p.N + n], r1 = range[lvl * p.N + n + 1];
    const int bins = PH * PW;
    unsigned short* dst = reinterpret_cast<unsigned short*>(const_cast<void*>(p.feat[lvl])) + (size_t)n * H * W * C;
    // (the 256-channel chunks of a wider map - R101-DC5's 2048-channel C5 - are workgroups of their own on blockIdx.y)
    const int cb = blockIdx.y * 256;
    const int cw = min(256, C - cb);
    const bool chan_ok = chunk * 64 + lane < cw;
    const unsigned short* gcol = gout + cb + chunk * 64 + lane;        // this lane's channel of g[k][bin][.]

    float acc[TILE][TILE / 2];
#pragma unroll
    for (int y = 0; y < TILE; ++y)
#pragma unroll
        for (int x = 0; x < TILE / 2; ++x) acc[y][x] = 0.f;
    int cur_t = 0;                                      // the tile (of the workgroup) the accumulators belong to

    // a tile is complete: its 8 x 4 pixels of this wave's 64 channels, one bf16 rounding; the accumulators start over
    auto flush = [&](int tt) {
        const int t = t_first + tt;
        const int tyi = t / txl, txi = t - tyi * txl;
        const int y0 = tyi * TILE, x0 = txi * TILE;
#pragma unroll
        for (int y = 0; y < TILE; ++y)
#pragma unroll
            for (int x = 0; x < TILE / 2; ++x) {
                const int py = y0 + y, px = x0 + xh * 4 + x;
                if (py < H && px < W) dst[((size_t)py * W + px) * C + cb + chunk * 64 + lane] = f32_to_bf16(acc[y][x]);
                acc[y][x] = 0.f;
            }
    };
    // 4 x 4 bins from (pb, qb); bins beyond the RoI's last row / column are CLAMPED, not skipped (no branches in the load
    // stream): their weights on this tile are zero by construction of the ranges
    auto load_block = [&](float (&v)[4][4], int kk, int pb, int qb) {
        const unsigned short* gk = gcol + (size_t)kk * bins * C;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pr = min(pb + i, PH - 1) * PW;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[i][j] = bf16_to_f32(gk[(size_t)(pr + min(qb + j, PW - 1)) * C]);
        }
    };
    // hits[0 .. nh) hold (tile, RoI) pairs in tile-major order: tables 32 pairs at a time, then every wave walks them
    auto process = [&](int nh) {
        for (int hb = 0; hb < nh; hb += HB) {
            __syncthreads();                           // hits written / the previous table batch consumed
            // ---- tables of pairs hb .. hb+31: 16 lanes per pair (0-7: bin rows, 8-15: bin columns)
            {
                const int hi_ = hb + (tid >> 4), sub = tid & 15;
                const bool rows = sub < 8;
                const int bin = sub & 7, nb = rows ? PH : PW;
                bool nz = false;
                if (hi_ < nh) {
                    TileTables& tb = tab[tid >> 4];
                    float* w = rows ? tb.wy[bin] : tb.wx[bin];
                    if (bin < nb) {
                        const int hv = hits[hi_];
                        const int t = t_first + (hv >> 24);
                        const int tyi = t / txl, txi = t - tyi * txl;
                        const RoiGeom g = roi_geom(rois + (size_t)(hv & 0xffffff) * 5, p, PH, PW, sampling_ratio, aligned);
                        if (rows) axis_tile_weights(g.start_h, g.bin_h, g.grid_h, bin, H, tyi * TILE, w);
                        else axis_tile_weights(g.start_w, g.bin_w, g.grid_w, bin, W, txi * TILE, w);
                        const float inv = rows ? 1.0f / g.count : 1.0f;
#pragma unroll
                        for (int j = 0; j < TILE; ++j) {
                            const float v = w[j];
                            nz = nz || v != 0.f;
                            if (rows) w[j] = v * inv;
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < TILE; ++j) w[j] = 0.f;             // bins beyond PH / PW: zeros
                    }
                }
                const unsigned mm = (unsigned)(__ballot(nz) >> ((lane >> 4) * 16)) & 0xffffu;
                if (hi_ < nh && (sub == 0 || sub == 8)) {
                    const unsigned f = (sub ? mm >> 8 : mm) & 0xffu;
                    const int lo = f ? __ffs((int)f) - 1 : nb, hi2 = f ? 32 - __clz((int)f) : 0;
                    TileTables& tb = tab[tid >> 4];
                    if (sub == 0) { tb.plo = lo; tb.phi = hi2; } else { tb.qlo = lo; tb.qhi = hi2; }
                }
            }
            __syncthreads();
            // ---- every wave walks the batch's pairs on its own: first 4 x 4 bin block of pair i+1 in flight
            const int nb_ = min(HB, nh - hb);
            if (chan_ok) {
                float nxt[4][4];
                load_block(nxt, hits[hb] & 0xffffff, __builtin_amdgcn_readfirstlane(tab[0].plo),
                           __builtin_amdgcn_readfirstlane(tab[0].qlo));
                for (int i2 = 0; i2 < nb_; ++i2) {
                    const TileTables& tb = tab[i2];
                    const int hv = __builtin_amdgcn_readfirstlane(hits[hb + i2]);
                    const int kk = hv & 0xffffff, tp = hv >> 24;
                    const int plo = __builtin_amdgcn_readfirstlane(tb.plo), phi = __builtin_amdgcn_readfirstlane(tb.phi);
                    const int qlo = __builtin_amdgcn_readfirstlane(tb.qlo), qhi = __builtin_amdgcn_readfirstlane(tb.qhi);
                    float cur[4][4];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) cur[i][j] = nxt[i][j];
                    if (i2 + 1 < nb_) {
                        const TileTables& tn = tab[i2 + 1];
                        load_block(nxt, hits[hb + i2 + 1] & 0xffffff, __builtin_amdgcn_readfirstlane(tn.plo),
                                   __builtin_amdgcn_readfirstlane(tn.qlo));
                    }
                    while (cur_t < tp) { flush(cur_t); ++cur_t; }              // the list moved on: earlier tiles are complete
                    if (plo >= phi || qlo >= qhi) continue;
                    // (the weights come as 16-byte LDS broadcasts - 4 column weights / 8 row weights per read - instead of
                    //  one v_readlane per weight: 96 -> 24 weight fetches per 4 x 4 bin block, off the VALU)
                    for (int pb = plo; pb < phi; pb += 4)
                        for (int qb = qlo; qb < qhi; qb += 4) {
                            if (pb != plo || qb != qlo) load_block(cur, kk, pb, qb);    // (rare: > 4 bins on a tile)
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                if (pb + i >= phi) continue;
                                float T[TILE / 2] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    if (qb + j >= qhi) continue;
                                    const f32x4 wxv = *reinterpret_cast<const f32x4*>(&tb.wx[qb + j][xh * 4]);
#pragma unroll
                                    for (int x = 0; x < TILE / 2; ++x) T[x] += wxv[x] * cur[i][j];
                                }
#pragma unroll
                                for (int yh = 0; yh < 2; ++yh) {
                                    const f32x4 wyv = *reinterpret_cast<const f32x4*>(&tb.wy[pb + i][yh * 4]);
#pragma unroll
                                    for (int y = 0; y < 4; ++y)
#pragma unroll
                                        for (int x = 0; x < TILE / 2; ++x) acc[yh * 4 + y][x] += wyv[y] * T[x];
                                }
                            }
                        }
                }
            }
        }
    };

    if (r1 - r0 <= TB_THREADS) {
        // ---- the common case: every thread tests ITS RoI's tile rectangle against all tiles of the workgroup at once
        int4 b = make_int4(1, 0, -1, -1);
        int myk = 0;
        const bool have = r0 + tid < r1;
        if (have) { b = boxes[r0 + tid]; myk = order[r0 + tid]; }
        const bool mine = have && b.z == lvl && b.w == n;
        unsigned long long mm[TPW];
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) {
            const int t = t_first + tt;
            const int tyi = t / txl, txi = t - tyi * txl;
            const bool hit = mine && tt < n_t && txi >= (b.x & 0xffff) && txi <= (b.x >> 16) && tyi >= (b.y & 0xffff) &&
                             tyi <= (b.y >> 16);
            mm[tt] = __ballot(hit);
            if (lane == 0) wcnt[tt * 8 + wave] = __popcll(mm[tt]);
        }
        __syncthreads();
        if (wave == 0) {                                 // exclusive prefix over the 64 (tile, wave) counts
            const int v = wcnt[lane];
            int incl = v;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(incl, d, 64);
                if (lane >= d) incl += o;
            }
            wbase[lane] = incl - v;
            if (lane == 63) wbase[64] = incl;
        }
        __syncthreads();
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt)
            if ((mm[tt] >> lane) & 1ull)
                hits[wbase[tt * 8 + wave] + __popcll(mm[tt] & ((1ull << lane) - 1ull))] = myk | (tt << 24);
        process(wbase[64]);
    } else {
        // ---- a group of more than 512 RoIs: tile by tile, 512 RoIs at a time (the order per tile is the position order)
        for (int tt = 0; tt < n_t; ++tt) {
            const int t = t_first + tt;
            const int tyi = t / txl, txi = t - tyi * txl;
            for (int rb = r0; rb < r1; rb += TB_THREADS) {
                bool hit = false;
                int k = -1;
                if (rb + tid < r1) {
                    const int4 b = boxes[rb + tid];
                    hit = b.z == lvl && b.w == n && txi >= (b.x & 0xffff) && txi <= (b.x >> 16) &&
                          tyi >= (b.y & 0xffff) && tyi <= (b.y >> 16);
                    if (hit) k = order[rb + tid];
                }
                const unsigned long long m = __ballot(hit);
                __syncthreads();                               // the previous batch's hits / tables are no longer read
                if (lane == 0) wcnt[wave] = __popcll(m);
                __syncthreads();
                int base = 0, nh = 0;
#pragma unroll
                for (int w2 = 0; w2 < TB_THREADS / 64; ++w2) {
                    if (w2 < wave) base += wcnt[w2];
                    nh += wcnt[w2];
                }
                if (hit) hits[base + __popcll(m & ((1ull << lane) - 1ull))] = k | (tt << 24);
                process(nh);
            }
        }
    }
    if (chan_ok)
        while (cur_t < n_t) { flush(cur_t); ++cur_t; }
}

// sort key of a RoI for the processing order of the two kernels above: (pyramid level, image, 16-feature-pixel cell
// of the RoI centre, row-major).  Proposals cluster around objects: workgroups that run together then read the same
// feature rows and - backward - add into the same gradient lines while those are still in L2.
__device__ __forceinline__ long long roi_order_key(const float* __restrict__ rois, int k, int n_img, int levels,
                                                   float finest_scale) {
    const float* r = rois + (size_t)k * 5;
    const float w = fmaxf(r[3] - r[1], 0.f), h = fmaxf(r[4] - r[2], 0.f);
    float l = floorf(log2f(sqrtf(w * h) / finest_scale + 1e-6f));
    l = fminf(fmaxf(l, 0.f), (float)(levels - 1));
    const int lvl = (int)l;
    const float cell = 64.f * (float)(1 << lvl);
    int qx = (int)((r[1] + r[3]) * 0.5f / cell), qy = (int)((r[2] + r[4]) * 0.5f / cell);
    qx = qx < 0 ? 0 : (qx > 1023 ? 1023 : qx);
    qy = qy < 0 ? 0 : (qy > 1023 ? 1023 : qy);
    int b = (int)r[0];
    b = b < 0 ? 0 : (b >= n_img ? n_img - 1 : b);
    return (((long long)lvl * n_img + b) * 1024 + qy) * 1024 + qx;
}

__global__ void roi_order_key_kernel(const float* __restrict__ rois, int K, int n_img, int levels, float finest_scale,
                                     long long* __restrict__ keys) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < K) keys[k] = roi_order_key(rois, k, n_img, levels, finest_scale);
}

// argsort of the keys (ties by index: a stable sort) + the first position of every (level, image) group in ONE launch,
// by counting: every workgroup holds all K composite keys (key << 16 | index) in LDS and thread i counts the smaller ones
// - K reads of one LDS address per step (a broadcast), K <= 8192.  torch: key kernel + sort (6 rocPRIM launches) +
// int cast + searchsorted, in the stretch of the step where the device waits for the host.
__global__ __launch_bounds__(256) void roi_order_rank_kernel(const float* __restrict__ rois, int K, int n_img, int levels,
                                                             float finest_scale, int* __restrict__ order,
                                                             int* __restrict__ range) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long rk_keys[];
    const int Kp = (K + 7) & ~7;                            // padded with keys that are smaller than nothing
    for (int j = threadIdx.x; j < Kp; j += 256)
        rk_keys[j] = j < K ? ((unsigned long long)roi_order_key(rois, j, n_img, levels, finest_scale) << 16) | (unsigned)j
                           : ~0ull;
    __syncthreads();
    // (round 4: eight keys per trip as four 16-byte broadcast reads and four independent counters - the one-key loop was a
    //  dependent chain of 4198 LDS round trips per thread: 124 us on the critical path between the RoI sampler and RoIAlign)
    auto count_below = [&](unsigned long long bound) {
        const ulonglong2* k2 = reinterpret_cast<const ulonglong2*>(rk_keys);
        int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        for (int j = 0; j < Kp / 2; j += 4) {
            const ulonglong2 a = k2[j], b = k2[j + 1], c = k2[j + 2], d = k2[j + 3];
            c0 += (a.x < bound ? 1 : 0) + (a.y < bound ? 1 : 0);
            c1 += (b.x < bound ? 1 : 0) + (b.y < bound ? 1 : 0);
            c2 += (c.x < bound ? 1 : 0) + (c.y < bound ? 1 : 0);
            c3 += (d.x < bound ? 1 : 0) + (d.y < bound ? 1 : 0);
        }
        return (c0 + c1) + (c2 + c3);
    };
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < K) order[count_below(rk_keys[i])] = i;
    const int groups = levels * n_img;
    if (blockIdx.x == gridDim.x - 1)                        // (the last workgroup is the least loaded one)
        for (int g = threadIdx.x; g <= groups; g += 256)
            range[g] = count_below((unsigned long long)g << 36);     // group g's smallest key: g << 20, then << 16
}

int fill_pyramid(Pyramid& p, const void* const* feats, float* const* dfeats, const int* heights,
                 const int* widths, const float* scales, int levels, int N, int C, float finest_scale) {
    if (levels < 1 || levels > OADG_MAX_LEVELS || N < 1 || C < 4 || (C & 3)) return OADG_EARG;
    p.levels = levels; p.N = N; p.C = C; p.finest_scale = finest_scale; p.order = nullptr;
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
                       void* out, const int* order, void* stream) {
    if (!feats || !heights || !widths || !scales || !rois || !out) return OADG_EARG;
    if (K < 0 || PH < 1 || PW < 1 || (dtype != 0 && dtype != 1)) return OADG_EARG;
    Pyramid p;
    const int rc = fill_pyramid(p, feats, nullptr, heights, widths, scales, levels, N, C, finest_scale);
    if (rc) return rc;
    p.order = order;
    if (K == 0) return OADG_OK;
    hipStream_t st = (hipStream_t)stream;
    static const bool rows = !(getenv("OADG_ROI_FWD_ROWS") && getenv("OADG_ROI_FWD_ROWS")[0] == '0');     // (A/B probes)
    const dim3 grid(K, (C + 255) / 256);
    if (rows && dtype == 0)
        hipLaunchKernelGGL((roi_align_fwd_rows_kernel<float>), grid, dim3(256), 0, st, p, rois, K, PH, PW, sampling_ratio,
                           aligned, (float*)out);
    else if (rows)
        hipLaunchKernelGGL((roi_align_fwd_rows_kernel<unsigned short>), grid, dim3(256), 0, st, p, rois, K, PH, PW,
                           sampling_ratio, aligned, (unsigned short*)out);
    else if (dtype == 0)
        hipLaunchKernelGGL((roi_align_fwd_kernel<float>), grid, dim3(256), 0, st, p, rois, K, PH, PW,
                           sampling_ratio, aligned, (float*)out);
    else
        hipLaunchKernelGGL((roi_align_fwd_kernel<unsigned short>), grid, dim3(256), 0, st, p, rois, K,
                           PH, PW, sampling_ratio, aligned, (unsigned short*)out);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

// keys[k] (int64) = locality sort key of RoI k; `order` of the two functions above = argsort(keys) as int32
int oadg_roi_order_keys(const float* rois, int K, int n_img, int levels, float finest_scale, long long* keys,
                        void* stream) {
    if (!rois || !keys || K < 0 || n_img < 1 || levels < 1 || levels > OADG_MAX_LEVELS) return OADG_EARG;
    if (K == 0) return OADG_OK;
    hipLaunchKernelGGL(roi_order_key_kernel, dim3((K + 255) / 256), dim3(256), 0, (hipStream_t)stream, rois, K, n_img,
                       levels, finest_scale, keys);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

// order [K] = argsort(keys) (stable) as int32 and range [levels * n_img + 1] = the first position in `order` of every
// (level, image) group, in one launch; K <= 8192 (OADG_EARG above: sort the keys of oadg_roi_order_keys instead)
int oadg_roi_order(const float* rois, int K, int n_img, int levels, float finest_scale, int* order, int* range,
                   void* stream) {
    if (!rois || !order || !range || K < 1 || K > 8192 || n_img < 1 || levels < 1 || levels > OADG_MAX_LEVELS) return OADG_EARG;
    hipLaunchKernelGGL(roi_order_rank_kernel, dim3((K + 255) / 256), dim3(256), (size_t)((K + 7) & ~7) * sizeof(unsigned long long),
                       (hipStream_t)stream, rois, K, n_img, levels, finest_scale, order, range);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

// Accumulates into the fp32 gradient maps dfeats[l] ([N, H_l, W_l, C], caller zero-initialises).
int oadg_roi_align_bwd(float* const* dfeats, const int* heights, const int* widths, const float* scales,
                       int levels, int N, int C, int dtype, float finest_scale, const float* rois, int K,
                       int PH, int PW, int sampling_ratio, int aligned, const void* grad_out,
                       const int* order, void* stream) {
    if (!dfeats || !heights || !widths || !scales || !rois || !grad_out) return OADG_EARG;
    if (K < 0 || PH < 1 || PW < 1 || (dtype != 0 && dtype != 1)) return OADG_EARG;
    Pyramid p;
    const int rc = fill_pyramid(p, nullptr, dfeats, heights, widths, scales, levels, N, C, finest_scale);
    if (rc) return rc;
    p.order = order;
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

// bf16 gradient maps written in full (no zero fill needed) by output tiles; `order` = RoI indices sorted by the keys of
// oadg_roi_order_keys, `range` [levels * N + 1] (device) = first position in `order` of every (level, image) group.
// PH, PW <= 8.
int oadg_roi_align_bwd_tiles(void* const* dmaps, const int* heights, const int* widths, const float* scales,
                             int levels, int N, int C, float finest_scale, const float* rois, int K, int PH, int PW,
                             int sampling_ratio, int aligned, const void* grad_out, const int* order, const int* range,
                             void* tile_boxes, void* stream) {
    if (!dmaps || !heights || !widths || !scales || !rois || !grad_out || !order || !range || !tile_boxes) return OADG_EARG;
    if (K < 0 || PH < 1 || PW < 1 || PH > RB_MAXP || PW > RB_MAXP) return OADG_EARG;
    Pyramid p;
    const int rc = fill_pyramid(p, (const void* const*)dmaps, nullptr, heights, widths, scales, levels, N, C, finest_scale);
    if (rc) return rc;
    TileGrid tg;
    // tiles per workgroup: 8 on an FPN pyramid (21,760 tiles at BASELINE config 2 -> 2,720 workgroups; an empty workgroup
    // costs ~10 us of lifetime), fewer when there are not enough tiles to fill the chip that way - R101-DC5's single
    // 46 x 80 level has 240 tiles: at 8 per workgroup 30 workgroups looped over 8 tiles x 8 channel chunks (11.3 ms);
    // now 240 x 8 workgroups (tile x chunk)
    const int nchunk = (C + 255) / 256;
    long tiles = 0;
    for (int l = 0; l < levels; ++l) tiles += (long)N * ((widths[l] + TILE - 1) / TILE) * ((heights[l] + TILE - 1) / TILE);
    int tpw = (int)(tiles * nchunk / 2048);
    tpw = tpw < 1 ? 1 : (tpw > TPW ? TPW : tpw);
    int total = 0;
    for (int l = 0; l < levels; ++l) {
        tg.first[l] = total;
        tg.tx[l] = (widths[l] + TILE - 1) / TILE;
        tg.ty[l] = (heights[l] + TILE - 1) / TILE;
        total += N * ((tg.tx[l] * tg.ty[l] + tpw - 1) / tpw);          // workgroups: tpw consecutive tiles of one image
    }
    tg.first[levels] = total;
    if (K > 0) {
        hipLaunchKernelGGL(roi_tile_box_kernel, dim3((K + 255) / 256), dim3(256), 0, (hipStream_t)stream, p, rois, K, PH, PW,
                           sampling_ratio, aligned, order, (int4*)tile_boxes);
        OADG_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(roi_align_bwd_tiles_kernel, dim3(total, nchunk), dim3(TB_THREADS), 0, (hipStream_t)stream, p, tg, rois, PH,
                       PW, sampling_ratio, aligned, (const unsigned short*)grad_out, order, range, (const int4*)tile_boxes, tpw);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

}  // extern "C"
