// Fused backward of the convolution epilogue  y = relu(conv + bias [+ residual]):
//   g = dy * (y > 0)            (torch threshold_backward; relu of resnet.py:285-300 / rpn_head.py:62)
//   dbias[k] = sum over pixels of g[., k]
// in ONE pass over dy (and y) instead of a mask pass, a cast pass and a strided column reduction.  HBM-bound:
// reads M*K*(sizeof dy + 2) bytes, writes M*K*2 bytes (nothing when only the column sums are wanted).
// Deterministic: per-block column partials in a fixed row order, then a fixed-order reduction over the blocks.
#include "common.h"
#include "../../include/oadg_hip.h"

namespace {

constexpr int EB_THREADS = 256;
constexpr int EB_MAX_BLOCKS = 1024;

__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {   // round to nearest even (torch's fp32 -> bf16)
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// Thread t owns channel group cg = t % G (8 channels, 16 B of bf16) and row phase rp = t / G of the block's row
// range; consecutive threads read consecutive 16-B pieces of a pixel row (coalesced).
template <bool DY_F32, bool MASK, bool WRITE>
__global__ __launch_bounds__(EB_THREADS) void relu_bias_bwd_kernel(const void* __restrict__ dy_,
                                                                    const unsigned short* __restrict__ y,
                                                                    unsigned short* __restrict__ g, float* __restrict__ part,
                                                                    long M, int K, long rows_per_block) {
    __shared__ float red[EB_THREADS][9];
    const int G = K >> 3;                                  // channel groups per row
    const int RP = EB_THREADS / G > 0 ? EB_THREADS / G : 1;  // row phases per pass (G <= 256)
    const int t = threadIdx.x;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
    // G > 256 (K > 2048): a thread strides over several channel groups of the row
    for (int cg = t % (G < EB_THREADS ? G : EB_THREADS); cg < G; cg += EB_THREADS) {
        const int rp = G < EB_THREADS ? t / G : 0;
        if (rp >= RP) break;
#pragma unroll 2
        for (long r = r0 + rp; r < r1; r += RP) {
            const long off = r * K + ((long)cg << 3);
            float v[8];
            if (DY_F32) {
                const float4 a = *reinterpret_cast<const float4*>((const float*)dy_ + off);
                const float4 b = *reinterpret_cast<const float4*>((const float*)dy_ + off + 4);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            } else {
                const uint4 a = *reinterpret_cast<const uint4*>((const unsigned short*)dy_ + off);
                const unsigned w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    v[2 * i] = __uint_as_float(w[i] << 16);
                    v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
                }
            }
            if (MASK) {
                const uint4 m = *reinterpret_cast<const uint4*>(y + off);
                const unsigned w[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (!(bf2f((unsigned short)(w[i] & 0xffffu)) > 0.f)) v[2 * i] = 0.f;
                    if (!(bf2f((unsigned short)(w[i] >> 16)) > 0.f)) v[2 * i + 1] = 0.f;
                }
            }
            if (WRITE) {
                unsigned o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned short lo = f2bf(v[2 * i]), hi = f2bf(v[2 * i + 1]);
                    o[i] = (unsigned)lo | ((unsigned)hi << 16);
                    if (DY_F32) {      // sum what the consumers of g will see
                        v[2 * i] = bf2f(lo);
                        v[2 * i + 1] = bf2f(hi);
                    }
                }
                *reinterpret_cast<uint4*>(g + off) = make_uint4(o[0], o[1], o[2], o[3]);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += v[i];
        }
        if (G >= EB_THREADS) {       // no row phases to combine: this thread owns the whole column group
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                part[(long)blockIdx.x * K + (cg << 3) + i] = acc[i];
                acc[i] = 0.f;
            }
        }
    }
    if (G >= EB_THREADS) return;
#pragma unroll
    for (int i = 0; i < 8; ++i) red[t][i] = acc[i];
    __syncthreads();
    if (t < G) {
        float s[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) s[i] = 0.f;
        for (int rp = 0; rp < RP; ++rp)
#pragma unroll
            for (int i = 0; i < 8; ++i) s[i] += red[rp * G + t][i];
#pragma unroll
        for (int i = 0; i < 8; ++i) part[(long)blockIdx.x * K + (t << 3) + i] = s[i];
    }
}

// out[k] = sum_b part[b][k] in a fixed order: 16 block slices x 16 channels per workgroup.
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                             int nblocks, int K) {
    __shared__ float red[16][17];
    const int c = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int k = blockIdx.x * 16 + c;
    float s = 0.f;
    if (k < K) {
        const int per = (nblocks + 15) >> 4;
        const int b0 = sl * per, b1 = b0 + per < nblocks ? b0 + per : nblocks;
#pragma unroll 8
        for (int b = b0; b < b1; ++b) s += part[(long)b * K + k];
    }
    red[sl][c] = s;
    __syncthreads();
    if (sl == 0 && k < K) {
        float tot = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) tot += red[i][c];
        out[k] = tot;
    }
}

// colsum_reduce_kernel for a table of jobs in one launch (same slices, same order: the same sums bit for bit), plus the
// finish of a BN scale gradient that prep_weights_bwd*_kernel left as a raw dot product (w_krsc bit 1)
__global__ __launch_bounds__(256) void colsum_reduce_multi_kernel(const oadg_colsum_job* __restrict__ jobs, int n) {
    __shared__ float red[16][17];
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const oadg_colsum_job j = jobs[lo];
    const int c = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int k = ((int)blockIdx.x - j.first_block) * 16 + c;
    const int K = j.K, nblocks = j.rows;
    float s = 0.f;
    if (k < K) {
        const int per = (nblocks + 15) >> 4;
        const int b0 = sl * per, b1 = b0 + per < nblocks ? b0 + per : nblocks;
#pragma unroll 8
        for (int b = b0; b < b1; ++b) s += j.part[(long)b * K + k];
    }
    red[sl][c] = s;
    __syncthreads();
    if (sl == 0 && k < K) {
        float tot = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) tot += red[i][c];
        j.out[k] = tot;
        if (j.dgamma) j.dgamma[k] = (j.dgamma[k] - tot * j.mean[k]) * rsqrtf(j.var[k] + j.eps);
    }
}

int plan_blocks(long M, int K, long* rows_per_block) {
    const int G = K >> 3;
    const int RP = EB_THREADS / G > 0 ? EB_THREADS / G : 1;
    long rpb = (M + EB_MAX_BLOCKS - 1) / EB_MAX_BLOCKS;
    const long min_rows = (long)RP * 4;                    // at least 4 passes per block
    if (rpb < min_rows) rpb = min_rows;
    rpb = (rpb + RP - 1) / RP * RP;
    *rows_per_block = rpb;
    return (int)((M + rpb - 1) / rpb);
}

// ---- FPN top-down step (fpn.py:166-175): out = lat + nearest_upsample(top), and its backward ------------------
// ATen's nearest index: src = min(floor(dst * (float)in / out), in - 1).
__device__ __forceinline__ int nearest_src(int dst, float scale, int in_size) {
    const int s = (int)floorf((float)dst * scale);
    return s < in_size - 1 ? s : in_size - 1;
}

__device__ __forceinline__ void unpack8(const uint4 a, float* v) {
    const unsigned w[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(w[i] << 16);
        v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
__device__ __forceinline__ uint4 pack8(const float* v) {
    unsigned o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (unsigned)f2bf(v[2 * i]) | ((unsigned)f2bf(v[2 * i + 1]) << 16);
    return make_uint4(o[0], o[1], o[2], o[3]);
}

// linear piece index -> (n, h, w, c8) of an [N][H][W][C8] tensor.  The 64-bit div / mod chain (six ~100-instruction
// sequences per piece) made these element-wise passes VALU-bound; below 2^32 pieces it is three 32-bit divisions.
__device__ __forceinline__ void split_nhwc(long i, int H, int W, int C8, int& n, int& h, int& w, int& c8) {
    if (i <= 0xffffffffL) {
        const unsigned u = (unsigned)i;
        const unsigned p = u / (unsigned)C8;
        c8 = (int)(u - p * (unsigned)C8);
        const unsigned q = p / (unsigned)W;
        w = (int)(p - q * (unsigned)W);
        const unsigned m = q / (unsigned)H;
        h = (int)(q - m * (unsigned)H);
        n = (int)m;
    } else {
        c8 = (int)(i % C8);
        long p = i / C8;
        w = (int)(p % W);
        p /= W;
        h = (int)(p % H);
        n = (int)(p / H);
    }
}

__global__ __launch_bounds__(256) void fpn_topdown_fwd_kernel(const unsigned short* __restrict__ lat,
                                                              const unsigned short* __restrict__ top,
                                                              unsigned short* __restrict__ out, int N, int H, int W,
                                                              int Ht, int Wt, int C8, float sh, float sw) {
    const long total = (long)N * H * W * C8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int n, h, w, c8;
        split_nhwc(i, H, W, C8, n, h, w, c8);
        const long src = (((long)n * Ht + nearest_src(h, sh, Ht)) * Wt + nearest_src(w, sw, Wt)) * C8 + c8;
        float a[8], b[8];
        unpack8(reinterpret_cast<const uint4*>(lat)[i], a);
        unpack8(reinterpret_cast<const uint4*>(top)[src], b);
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] += b[k];
        reinterpret_cast<uint4*>(out)[i] = pack8(a);
    }
}

// d top[n, ht, wt, :] = sum of g over the destination pixels whose nearest source is (ht, wt)
__global__ __launch_bounds__(256) void fpn_topdown_bwd_kernel(const unsigned short* __restrict__ g,
                                                              unsigned short* __restrict__ dtop, int N, int H, int W,
                                                              int Ht, int Wt, int C8, float sh, float sw) {
    const long total = (long)N * Ht * Wt * C8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int n, ht, wt, c8;
        split_nhwc(i, Ht, Wt, C8, n, ht, wt, c8);
        // candidate destination range: a superset of {d : nearest_src(d) == s}, filtered exactly below
        int h0 = (int)floorf((float)ht / sh) - 1, h1 = (int)ceilf((float)(ht + 1) / sh) + 1;
        int w0 = (int)floorf((float)wt / sw) - 1, w1 = (int)ceilf((float)(wt + 1) / sw) + 1;
        if (ht == Ht - 1) h1 = H - 1;          // the clamp of nearest_src folds every overshoot into the last source
        if (wt == Wt - 1) w1 = W - 1;
        h0 = h0 < 0 ? 0 : h0;
        w0 = w0 < 0 ? 0 : w0;
        h1 = h1 > H - 1 ? H - 1 : h1;
        w1 = w1 > W - 1 ? W - 1 : w1;
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        for (int h = h0; h <= h1; ++h) {
            if (nearest_src(h, sh, Ht) != ht) continue;
            for (int w = w0; w <= w1; ++w) {
                if (nearest_src(w, sw, Wt) != wt) continue;
                float v[8];
                unpack8(reinterpret_cast<const uint4*>(g)[(((long)n * H + h) * W + w) * C8 + c8], v);
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[k] += v[k];
            }
        }
        reinterpret_cast<uint4*>(dtop)[i] = pack8(acc);
    }
}

}  // namespace

extern "C" size_t oadg_relu_bias_bwd_workspace_bytes(long M, int K) {
    if (M <= 0 || K <= 0 || (K & 7)) return 0;
    long rpb;
    return (size_t)plan_blocks(M, K, &rpb) * K * sizeof(float);
}

extern "C" int oadg_relu_bias_bwd(const void* dy, int dy_is_f32, const void* y, void* g, float* dbias,
                                  void* workspace, size_t workspace_bytes, long M, int K, void* stream) {
    if (!dy || !dbias || !workspace || M <= 0 || K <= 0 || (K & 7)) return OADG_EARG;
    if (!g && (y || dy_is_f32)) return OADG_EARG;         // a masked / converted gradient has to be written somewhere
    long rpb;
    const int nb = plan_blocks(M, K, &rpb);
    if (workspace_bytes < (size_t)nb * K * sizeof(float)) return OADG_ESIZE;
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)workspace;
    const unsigned short* yy = (const unsigned short*)y;
    unsigned short* gg = (unsigned short*)g;
#define EB_LAUNCH(F32, MASK, WRITE) \
    hipLaunchKernelGGL((relu_bias_bwd_kernel<F32, MASK, WRITE>), dim3(nb), dim3(EB_THREADS), 0, st, dy, yy, gg, part, M, K, rpb)
    if (dy_is_f32) {
        if (y) EB_LAUNCH(true, true, true); else EB_LAUNCH(true, false, true);
    } else if (y) {
        EB_LAUNCH(false, true, true);
    } else if (g) {
        EB_LAUNCH(false, false, true);
    } else {
        EB_LAUNCH(false, false, false);
    }
#undef EB_LAUNCH
    OADG_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((K + 15) / 16), dim3(256), 0, st, part, dbias, nb, K);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

extern "C" int oadg_fpn_topdown_fwd(const void* lat, const void* top, void* out, int N, int H, int W, int Ht, int Wt,
                                    int C, void* stream) {
    if (!lat || !top || !out || N <= 0 || H <= 0 || W <= 0 || Ht <= 0 || Wt <= 0 || C <= 0 || (C & 7)) return OADG_EARG;
    const long total = (long)N * H * W * (C >> 3);
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(fpn_topdown_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)lat, (const unsigned short*)top, (unsigned short*)out, N, H, W, Ht, Wt,
                       C >> 3, (float)Ht / (float)H, (float)Wt / (float)W);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

extern "C" int oadg_fpn_topdown_bwd(const void* g, void* dtop, int N, int H, int W, int Ht, int Wt, int C,
                                    void* stream) {
    if (!g || !dtop || N <= 0 || H <= 0 || W <= 0 || Ht <= 0 || Wt <= 0 || C <= 0 || (C & 7)) return OADG_EARG;
    const long total = (long)N * Ht * Wt * (C >> 3);
    const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(fpn_topdown_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)g, (unsigned short*)dtop, N, H, W, Ht, Wt, C >> 3,
                       (float)Ht / (float)H, (float)Wt / (float)W);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

// ResNet stem tail (resnet.py:631-637 `relu(norm1(conv1(x)))` -> `maxpool`, BN folded): out = maxpool3x3/s2/p1(relu(x + b)).
// x + b is rounded to bf16 before the ReLU like the unfused element-wise chain; add, rounding and ReLU are monotone, so
// they are applied once to the window maximum - bit-identical to add -> relu -> max_pool2d, in one pass instead of three.
namespace {
__global__ __launch_bounds__(256) void bias_relu_maxpool_kernel(const unsigned short* __restrict__ x,
                                                                const float* __restrict__ bias,
                                                                unsigned short* __restrict__ out, int N, int H, int W,
                                                                int Ho, int Wo, int C8) {
    const long total = (long)N * Ho * Wo * C8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int n, ho, wo, c8;
        split_nhwc(i, Ho, Wo, C8, n, ho, wo, c8);
        float m[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) m[k] = -INFINITY;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int h = 2 * ho - 1 + dy;
            if ((unsigned)h >= (unsigned)H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int w = 2 * wo - 1 + dx;
                if ((unsigned)w >= (unsigned)W) continue;
                float a[8];
                unpack8(reinterpret_cast<const uint4*>(x)[(((long)n * H + h) * W + w) * C8 + c8], a);
#pragma unroll
                for (int k = 0; k < 8; ++k) m[k] = fmaxf(m[k], a[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float v = m[k];
            // autocast: the bias is cast to bf16, the sum is rounded to bf16
            if (bias) v = bf16_to_f32(f32_to_bf16(v + bf16_to_f32(f32_to_bf16(bias[c8 * 8 + k]))));
            m[k] = fmaxf(v, 0.f);
        }
        reinterpret_cast<uint4*>(out)[i] = pack8(m);
    }
}
}  // namespace

extern "C" int oadg_bias_relu_maxpool_nhwc_bf16(const void* x, const float* bias, void* out, int N, int H, int W, int C,
                                                void* stream) {
    if (!x || !out || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7)) return OADG_EARG;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long total = (long)N * Ho * Wo * (C >> 3);
    const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
    hipLaunchKernelGGL(bias_relu_maxpool_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)x, bias, (unsigned short*)out, N, H, W, Ho, Wo, C >> 3);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

// out[k] = sum_r part[r][k] in a fixed order (deterministic): the second stage of every column-sum producer
extern "C" int oadg_colsum_reduce(const float* part, long rows, int K, float* out, void* stream) {
    if (!part || !out || rows < 1 || K < 1 || rows > 0x7fffffffL) return OADG_EARG;
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((K + 15) / 16), dim3(256), 0, (hipStream_t)stream, part, out, (int)rows,
                       K);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

// ---- first FC of the RoI head on RoIAlign's own (ph, pw, c) feature order ---------------------------------------------
// The reference flattens NCHW RoI features, so column c*P + p of the FC weight meets channel c of bin p
// (convfc_bbox_head.py: x.flatten(1), P = 7*7).  RoIAlign here writes [K][ph][pw][C]; instead of permuting 105 MB of
// features forward (and their gradient backward) every step, the weight's columns are permuted inside the two passes
// autocast makes anyway: fp32 -> bf16 forward (mode 0) and bf16 -> fp32 for the gradient (mode 1).
namespace {
constexpr int FCP_TC = 64;        // channels per workgroup tile

template <int MODE>
__global__ __launch_bounds__(256) void fc_weight_permute_kernel(const void* __restrict__ src, void* __restrict__ dst, int C,
                                                                int P) {
    extern __shared__ float fcp_tile[];                       // [FCP_TC][P + 1]
    const int o = blockIdx.y, c0 = blockIdx.x * FCP_TC, n = FCP_TC * P, ld = P + 1;
    const size_t cp_base = ((size_t)o * C + c0) * P;          // [o][c0 ..][p]: FCP_TC * P contiguous elements
    if (MODE == 0) {
        const float* s = reinterpret_cast<const float*>(src) + cp_base;
        for (int e = threadIdx.x; e < n; e += 256) fcp_tile[(e / P) * ld + e % P] = s[e];
        __syncthreads();
        unsigned short* d = reinterpret_cast<unsigned short*>(dst);
        for (int e = threadIdx.x; e < n; e += 256) {
            const int p = e / FCP_TC, c = e % FCP_TC;
            d[((size_t)o * P + p) * C + c0 + c] = f32_to_bf16(fcp_tile[c * ld + p]);
        }
    } else {
        const unsigned short* s = reinterpret_cast<const unsigned short*>(src);
        for (int e = threadIdx.x; e < n; e += 256) {
            const int p = e / FCP_TC, c = e % FCP_TC;
            fcp_tile[c * ld + p] = bf16_to_f32(s[((size_t)o * P + p) * C + c0 + c]);
        }
        __syncthreads();
        float* d = reinterpret_cast<float*>(dst) + cp_base;
        for (int e = threadIdx.x; e < n; e += 256) d[e] = fcp_tile[(e / P) * ld + e % P];
    }
}
}  // namespace

extern "C" int oadg_fc_weight_permute(const void* src, void* dst, int O, int C, int P, int mode, void* stream) {
    if (!src || !dst || O < 1 || C < FCP_TC || (C % FCP_TC) || P < 1 || P > 256 || (mode != 0 && mode != 1) || O > 65535)
        return OADG_EARG;
    const size_t smem = (size_t)FCP_TC * (P + 1) * sizeof(float);
    if (smem > 48 * 1024) {          // 14 x 14 bins and more: above the default dynamic-LDS limit (P = 256: 65,792 bytes)
        static bool attr = false;
        if (!attr) {
            hipError_t e = hipFuncSetAttribute((const void*)fc_weight_permute_kernel<0>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, FCP_TC * 257 * sizeof(float));
            if (e == hipSuccess)
                e = hipFuncSetAttribute((const void*)fc_weight_permute_kernel<1>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, FCP_TC * 257 * sizeof(float));
            if (e != hipSuccess) return (int)e;
            attr = true;
        }
    }
    const dim3 grid(C / FCP_TC, O);
    if (mode == 0)
        hipLaunchKernelGGL(fc_weight_permute_kernel<0>, grid, dim3(256), smem, (hipStream_t)stream, src, dst, C, P);
    else
        hipLaunchKernelGGL(fc_weight_permute_kernel<1>, grid, dim3(256), smem, (hipStream_t)stream, src, dst, C, P);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

// the reductions of several producers in one launch: jobs_dev [n] on the DEVICE in ascending first_block order, job i owns
// blocks [first_block_i, first_block_i + (K_i + 15) / 16)
extern "C" int oadg_colsum_reduce_multi(const oadg_colsum_job* jobs_dev, int n, int total_blocks, void* stream) {
    if (!jobs_dev || n < 1 || total_blocks < 1) return OADG_EARG;
    hipLaunchKernelGGL(colsum_reduce_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev,
                       n);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}
