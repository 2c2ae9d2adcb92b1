// fp32 convolution on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 products and sums, the rate of the
// vector ALUs - 1/16 of bf16) for the fp32 PARITY path: forward, data gradient (any stride: the transposed gather form)
// and weight gradient of every convolution the reference runs through cuDNN in fp32
//   mmdet/models/backbones/resnet.py:263-302,379-660, necks/fpn.py:112-129,151-205, dense_heads/rpn_head.py:54-68
// The reference trains in fp32; the benchmarked configuration here is bf16 autocast on
// csrc/conv_mfma.hip.  Rounds 1-4 ran fp32 steps (the fp32 whole-step parity tests, `bench.py --dtype fp32`) on MIOpen
// through F.conv2d; these kernels take that over so that no library convolution is left on either path.  They are written
// for exactness and generality (any K, C % 4 == 0, any stride / dilation / padding), not for the roofline: NHWC operands,
// 128 pixels x 64 channels x 32 reduction elements per workgroup through one LDS stage filled by global_load_lds.
//
// GEMM view (forward): M = N*Ho*Wo pixels, N = K channels, reduction (r, s, c).  Transposed mode (data gradient):
// the "input" is dy [N][Hs][Ws][Cs], the output pixel (n, oh, ow) of dx gathers, for every tap (r, s), source pixel
// ((oh + pad - r*dil) / stride, (ow + pad - s*dil) / stride) when both divisions are exact and in range - the adjoint of
// the forward gather for any stride, no zero-inserted tensor; weights [C][R][S][K].
#include "common.h"
#include "../../include/oadg_hip.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

struct F32Args {
    const float* x;        // [N][H][W][C]
    const float* w;        // [K][R][S][C]
    const float* bias;     // [K] or null
    float* y;              // [N][Ho][Wo][K]
    const float* zeros;    // >= 16 bytes of zeros
    int N, H, W, C, K, R, S, Ho, Wo, stride, pad, dil, transposed;
    long M;
};

constexpr int FM = 128, FN = 64, FK = 32;       // pixels x channels x reduction elements (one 128-byte row each)

__global__ __launch_bounds__(256) void conv_f32_kernel(F32Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[(FM + FN) * FK * 4];
    unsigned char* sa = smem;
    unsigned char* sb = smem + FM * FK * 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_tiles = (a.K + FN - 1) / FN;
    const long mt = blockIdx.x / n_tiles;
    const int nt = blockIdx.x % n_tiles;
    const long m0 = mt * FM;
    const int k0 = nt * FN;
    // loader: piece q = i*256 + tid -> row q >> 3, 16-byte slot q & 7 (LDS image lane-linear, swizzle on the source)
    int oh[4], ow[4], seg[4];
    const float* img[4];
    bool live[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = i * 256 + tid, row = q >> 3;
        seg[i] = (q & 7) ^ ((row >> 1) & 7);
        const long m = m0 + row;
        live[i] = m < a.M;
        const long mm = live[i] ? m : 0;
        const int wo = (int)(mm % a.Wo);
        const long t = mm / a.Wo;
        const int ho = (int)(t % a.Ho);
        const long n = t / a.Ho;
        oh[i] = ho; ow[i] = wo;
        img[i] = a.x + (size_t)n * a.H * a.W * a.C;
    }
    int brow[2], bseg[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = i * 256 + tid;
        brow[i] = q >> 3;
        bseg[i] = (q & 7) ^ ((brow[i] >> 1) & 7);
    }
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int l31 = lane & 31, lh = lane >> 5;
    const int cchunks = (a.C + FK - 1) / FK;
    for (int r = 0; r < a.R; ++r)
        for (int s = 0; s < a.S; ++s)
            for (int cc = 0; cc < cchunks; ++cc) {
                const int c0 = cc * FK;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int hi, wi;
                    bool ok = live[i] && c0 + seg[i] * 4 < a.C;
                    if (a.transposed) {
                        const int nh = oh[i] + a.pad - r * a.dil, nw = ow[i] + a.pad - s * a.dil;
                        hi = nh / a.stride; wi = nw / a.stride;
                        ok = ok && nh >= 0 && nw >= 0 && hi * a.stride == nh && wi * a.stride == nw && hi < a.H && wi < a.W;
                    } else {
                        hi = oh[i] * a.stride - a.pad + r * a.dil;
                        wi = ow[i] * a.stride - a.pad + s * a.dil;
                        ok = ok && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
                    }
                    const float* src = ok ? img[i] + ((size_t)hi * a.W + wi) * a.C + c0 + seg[i] * 4 : a.zeros;
                    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(sa + i * 4096 + wave * 1024), 16, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int k = k0 + brow[i];
                    const bool ok = k < a.K && c0 + bseg[i] * 4 < a.C;
                    const float* src = ok ? a.w + (((size_t)k * a.R + r) * a.S + s) * a.C + c0 + bseg[i] * 4 : a.zeros;
                    __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(sb + i * 4096 + wave * 1024), 16, 0, 0);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                // a lane's four reduction elements per 16-byte read: k = 8 jj + 4 lh + e (the same pairing for both operands)
                // (blocked summation: the 32 products of a stage are chained in a fresh accumulator and THEN added to the
                //  running sum - one long fp32 chain over R*S*C = 4608 products drifts ~10x further from the reference's
                //  blocked library sums than this: measured on the R101-DC5 whole step, gradient norms 7e-4 -> 8e-5)
                f32x16 part[2];
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 16; ++q) part[j][q] = 0.f;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int sg = 2 * jj + lh;
                    const int arow = wave * 32 + l31;
                    const f32x4 fa = *reinterpret_cast<const f32x4*>(sa + arow * 128 + ((sg ^ ((arow >> 1) & 7)) << 4));
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int bro = j * 32 + l31;
                        const f32x4 fb = *reinterpret_cast<const f32x4*>(sb + bro * 128 + ((sg ^ ((bro >> 1) & 7)) << 4));
#pragma unroll
                        for (int e = 0; e < 4; ++e) part[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[e], fb[e], part[j], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[j] += part[j];
                __syncthreads();
            }
    // D[row = pixel][col = channel]: a lane holds column l31 and rows (r & 3) + 8 (r >> 2) + 4 lh of each 32 x 32 tile
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int k = k0 + j * 32 + l31;
        if (k >= a.K) continue;
        const float bv = a.bias ? a.bias[k] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m < a.M) a.y[(size_t)m * a.K + k] = acc[j][r] + bv;
        }
    }
}

// dW partials: one wave = one 32 (k) x 32 (c) tile of one tap over one pixel range; operands straight from global memory
// (a lane's MFMA operand is ONE element: row l31 of the tile, reduction index lh - the 32 lanes of a half read 128
// contiguous bytes of one pixel); fp32 partial tiles summed by a second kernel in split order (deterministic).
struct W32Args {
    const float* x;        // [N][H][W][C]
    const float* dy;       // [N][Ho][Wo][K]
    float* part;           // [splits][K][R*S][C]
    int N, H, W, C, K, R, S, Ho, Wo, stride, pad, dil, splits;
    long P, per_split;
};

__global__ __launch_bounds__(64) void conv_wgrad_f32_kernel(W32Args a) {
    const int lane = threadIdx.x, l31 = lane & 31, lh = lane >> 5;
    const int kt_n = (a.K + 31) / 32, ct_n = (a.C + 31) / 32, RS = a.R * a.S;
    long b = blockIdx.x;
    const int ct = (int)(b % ct_n); b /= ct_n;
    const int kt = (int)(b % kt_n); b /= kt_n;
    const int rs = (int)(b % RS);
    const int split = (int)(b / RS);
    const int r = rs / a.S, s = rs - r * a.S;
    const int k = kt * 32 + l31, c = ct * 32 + l31;
    const long p0 = (long)split * a.per_split, p1 = p0 + a.per_split < a.P ? p0 + a.per_split : a.P;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    f32x16 part;
#pragma unroll
    for (int i = 0; i < 16; ++i) part[i] = 0.f;
    int since = 0;
    for (long pb = p0; pb < p1; pb += 8) {
        if (since == 8) {          // blocked summation: 64 pixels per chain (see conv_f32_kernel)
            acc += part;
#pragma unroll
            for (int i = 0; i < 16; ++i) part[i] = 0.f;
            since = 0;
        }
        ++since;
        float av[4], bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long p = pb + 2 * u + lh;
            av[u] = bv[u] = 0.f;
            if (p < p1) {
                const int wo = (int)(p % a.Wo);
                const long t = p / a.Wo;
                const int ho = (int)(t % a.Ho);
                const long n = t / a.Ho;
                const int hi = ho * a.stride - a.pad + r * a.dil, wi = wo * a.stride - a.pad + s * a.dil;
                if (k < a.K) av[u] = a.dy[(size_t)p * a.K + k];
                if (c < a.C && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W)
                    bv[u] = a.x[(((size_t)n * a.H + hi) * a.W + wi) * a.C + c];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) part = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], part, 0, 0, 0);
    }
    acc += part;
    if (c < a.C) {
        float* out = a.part + (size_t)split * a.K * RS * a.C;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kk = kt * 32 + (i & 3) + 8 * (i >> 2) + 4 * lh;
            if (kk < a.K) out[((size_t)kk * RS + rs) * a.C + c] = acc[i];
        }
    }
}

__global__ void wgrad_f32_reduce_kernel(const float* __restrict__ part, int splits, long n, float* __restrict__ dw) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += part[(size_t)k * n + i];
    dw[i] = s;
}

}  // namespace

extern "C" int oadg_conv2d_f32(const float* x, const float* w, const float* bias, float* y, const void* zeros16, int N,
                               int H, int W, int C, int K, int R, int S, int stride, int pad, int dil, int transposed,
                               int out_h, int out_w, void* stream) {
    if (!x || !w || !y || !zeros16 || N < 1 || H < 1 || W < 1 || C < 4 || (C & 3) || K < 1 || R < 1 || S < 1 || stride < 1 ||
        dil < 1 || pad < 0)
        return OADG_EARG;
    F32Args a;
    a.x = x; a.w = w; a.bias = bias; a.y = y; a.zeros = (const float*)zeros16;
    a.N = N; a.H = H; a.W = W; a.C = C; a.K = K; a.R = R; a.S = S; a.stride = stride; a.pad = pad; a.dil = dil;
    a.transposed = transposed ? 1 : 0;
    if (transposed) {
        if (out_h < 1 || out_w < 1) return OADG_EARG;
        a.Ho = out_h; a.Wo = out_w;
    } else {
        a.Ho = (H + 2 * pad - dil * (R - 1) - 1) / stride + 1;
        a.Wo = (W + 2 * pad - dil * (S - 1) - 1) / stride + 1;
        if (a.Ho < 1 || a.Wo < 1) return OADG_EARG;
    }
    a.M = (long)N * a.Ho * a.Wo;
    const long blocks = ((a.M + FM - 1) / FM) * ((K + FN - 1) / FN);
    if (blocks > 0x7fffffffL) return OADG_EARG;
    hipLaunchKernelGGL(conv_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

extern "C" int oadg_conv2d_wgrad_f32_splits(int N, int Ho, int Wo, int C, int K, int R, int S) {
    const long P = (long)N * Ho * Wo;
    const long tiles = (long)((K + 31) / 32) * ((C + 31) / 32) * R * S;
    long s = (4096 + tiles - 1) / tiles;           // ~4096 waves: every SIMD busy
    const long most = (P + 511) / 512;             // at least 512 pixels per split
    if (s > most) s = most;
    if (s < 1) s = 1;
    if (s > 1024) s = 1024;
    return (int)s;
}

// dw [K][R][S][C] fp32 (overwritten); workspace: splits * K * R * S * C floats
extern "C" int oadg_conv2d_wgrad_f32(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes,
                                     int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                                     void* stream) {
    if (!x || !dy || !dw || !workspace || N < 1 || C < 1 || K < 1 || R < 1 || S < 1 || stride < 1 || dil < 1 || pad < 0)
        return OADG_EARG;
    W32Args a;
    a.x = x; a.dy = dy; a.part = (float*)workspace;
    a.N = N; a.H = H; a.W = W; a.C = C; a.K = K; a.R = R; a.S = S; a.stride = stride; a.pad = pad; a.dil = dil;
    a.Ho = (H + 2 * pad - dil * (R - 1) - 1) / stride + 1;
    a.Wo = (W + 2 * pad - dil * (S - 1) - 1) / stride + 1;
    if (a.Ho < 1 || a.Wo < 1) return OADG_EARG;
    a.P = (long)N * a.Ho * a.Wo;
    a.splits = oadg_conv2d_wgrad_f32_splits(N, a.Ho, a.Wo, C, K, R, S);
    a.per_split = ((a.P + a.splits - 1) / a.splits + 7) / 8 * 8;
    const long n = (long)K * R * S * C;
    if (workspace_bytes < (size_t)a.splits * n * sizeof(float)) return OADG_ESIZE;
    const long blocks = (long)a.splits * R * S * ((K + 31) / 32) * ((C + 31) / 32);
    if (blocks > 0x7fffffffL) return OADG_EARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(conv_wgrad_f32_kernel, dim3((unsigned)blocks), dim3(64), 0, st, a);
    OADG_LAUNCH_CHECK();
    hipLaunchKernelGGL(wgrad_f32_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)workspace,
                       a.splits, n, dw);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}
