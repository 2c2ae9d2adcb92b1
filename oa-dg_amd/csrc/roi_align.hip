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
    const int k = p.order ? p.order[blockIdx.x] : (int)blockIdx.x;
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

__global__ __launch_bounds__(256) void roi_align_bwd_tiles_kernel(Pyramid p, TileGrid tg, const float* __restrict__ rois,
                                                                  int PH, int PW, int sampling_ratio, int aligned,
                                                                  const unsigned short* __restrict__ gout,
                                                                  const int* __restrict__ order,
                                                                  const int* __restrict__ range) {
    __shared__ float wy[RB_MAXP][TILE], wx[RB_MAXP][TILE];
    __shared__ unsigned short slab[RB_MAXP * RB_MAXP * 256];
    __shared__ int hits[256];
    __shared__ int nhit;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int lvl = 0;
    while (lvl + 1 < p.levels && (int)blockIdx.x >= tg.first[lvl + 1]) ++lvl;
    const int local = blockIdx.x - tg.first[lvl];
    const int per_img = tg.tx[lvl] * tg.ty[lvl];
    const int n = local / per_img, t = local - n * per_img;
    const int y0 = (t / tg.tx[lvl]) * TILE, x0 = (t % tg.tx[lvl]) * TILE;
    const int H = p.H[lvl], W = p.W[lvl], C = p.C;
    const int r0 = range[lvl * p.N + n], r1 = range[lvl * p.N + n + 1];
    const int bins = PH * PW;
    unsigned short* dst = reinterpret_cast<unsigned short*>(const_cast<void*>(p.feat[lvl])) + (size_t)n * H * W * C;

    for (int cb = 0; cb < C; cb += 256) {
        const int cw = min(256, C - cb);
        float acc[2 * TILE][4];
#pragma unroll
        for (int i = 0; i < 2 * TILE; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
        for (int rb = r0; rb < r1; rb += 256) {
            // which of the next 256 RoIs of this (level, image) can touch the tile?  (conservative pixel bounds)
            __syncthreads();
            if (tid == 0) nhit = 0;
            __syncthreads();
            bool hit = false;
            int k = -1;
            if (rb + tid < r1) {
                k = order[rb + tid];
                const RoiGeom g = roi_geom(rois + (size_t)k * 5, p, PH, PW, sampling_ratio, aligned);
                const float rh = g.bin_h * PH, rw = g.bin_w * PW;
                const float ylo = floorf(g.start_h) - 1.f, yhi = ceilf(g.start_h + rh) + 1.f;
                const float xlo = floorf(g.start_w) - 1.f, xhi = ceilf(g.start_w + rw) + 1.f;
                hit = g.lvl == lvl && g.batch == n && yhi >= (float)y0 && ylo < (float)(y0 + TILE) &&
                      xhi >= (float)x0 && xlo < (float)(x0 + TILE);
            }
            // compact in position order (a fixed summation order): ballot + prefix over the four waves
            const unsigned long long m = __ballot(hit);
            __shared__ int wcnt[4];
            if (lane == 0) wcnt[wave] = __popcll(m);
            __syncthreads();
            int base = 0;
            for (int w2 = 0; w2 < wave; ++w2) base += wcnt[w2];
            if (hit) hits[base + __popcll(m & ((1ull << lane) - 1ull))] = k;
            if (tid == 0) nhit = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
            __syncthreads();
            const int nh = nhit;
            for (int hidx = 0; hidx < nh; ++hidx) {
                const int kk = hits[hidx];
                const RoiGeom g = roi_geom(rois + (size_t)kk * 5, p, PH, PW, sampling_ratio, aligned);
                __syncthreads();                       // previous RoI's tables / slab are no longer read
                if (tid < PH) axis_tile_weights(g.start_h, g.bin_h, g.grid_h, tid, H, y0, wy[tid]);
                else if (tid >= 64 && tid < 64 + PW) axis_tile_weights(g.start_w, g.bin_w, g.grid_w, tid - 64, W, x0, wx[tid - 64]);
                for (int q = tid; q < bins * 64; q += 256) {          // 4 channels per piece
                    const int bin = q >> 6, c4 = (q & 63) << 2;
                    if (c4 < cw) {
                        const unsigned short* src = gout + ((size_t)kk * bins + bin) * C + cb + c4;
                        *reinterpret_cast<bf16x4*>(slab + bin * 256 + c4) = *reinterpret_cast<const bf16x4*>(src);
                    }
                }
                __syncthreads();
                const float inv = 1.0f / g.count;
#pragma unroll
                for (int i = 0; i < 2 * TILE; ++i) {
                    const int ry = wave * 2 + (i >> 3), rx = i & 7;       // this wave: tile rows 2w, 2w+1
                    for (int ph = 0; ph < PH; ++ph) {
                        const float wyv = wy[ph][ry];
                        if (wyv == 0.f) continue;
                        for (int pw = 0; pw < PW; ++pw) {
                            const float wv = wyv * wx[pw][rx];
                            if (wv == 0.f) continue;
                            const unsigned short* row = slab + (ph * PW + pw) * 256;
                            const float ws = wv * inv;
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (lane + 64 * j < cw) acc[i][j] += ws * bf16_to_f32(row[lane + 64 * j]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 2 * TILE; ++i) {
            const int py = y0 + wave * 2 + (i >> 3), px = x0 + (i & 7);
            if (py >= H || px >= W) continue;
            unsigned short* o = dst + ((size_t)py * W + px) * C + cb;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (lane + 64 * j < cw) o[lane + 64 * j] = f32_to_bf16(acc[i][j]);
        }
    }
}

// sort key of a RoI for the processing order of the two kernels above: (pyramid level, image, 16-feature-pixel cell
// of the RoI centre, row-major).  Proposals cluster around objects: workgroups that run together then read the same
// feature rows and - backward - add into the same gradient lines while those are still in L2.
__global__ void roi_order_key_kernel(const float* __restrict__ rois, int K, int n_img, int levels, float finest_scale,
                                     long long* __restrict__ keys) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
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
    keys[k] = (((long long)lvl * n_img + b) * 1024 + qy) * 1024 + qx;
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
    if (dtype == 0)
        hipLaunchKernelGGL((roi_align_fwd_kernel<float>), dim3(K), dim3(256), 0, st, p, rois, K, PH, PW,
                           sampling_ratio, aligned, (float*)out);
    else
        hipLaunchKernelGGL((roi_align_fwd_kernel<unsigned short>), dim3(K), dim3(256), 0, st, p, rois, K,
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
                             void* stream) {
    if (!dmaps || !heights || !widths || !scales || !rois || !grad_out || !order || !range) return OADG_EARG;
    if (K < 0 || PH < 1 || PW < 1 || PH > RB_MAXP || PW > RB_MAXP) return OADG_EARG;
    Pyramid p;
    const int rc = fill_pyramid(p, (const void* const*)dmaps, nullptr, heights, widths, scales, levels, N, C, finest_scale);
    if (rc) return rc;
    TileGrid tg;
    int total = 0;
    for (int l = 0; l < levels; ++l) {
        tg.first[l] = total;
        tg.tx[l] = (widths[l] + TILE - 1) / TILE;
        tg.ty[l] = (heights[l] + TILE - 1) / TILE;
        total += N * tg.tx[l] * tg.ty[l];
    }
    tg.first[levels] = total;
    hipLaunchKernelGGL(roi_align_bwd_tiles_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, p, tg, rois, PH, PW,
                       sampling_ratio, aligned, (const unsigned short*)grad_out, order, range);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

}  // extern "C"
