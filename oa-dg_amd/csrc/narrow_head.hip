// 1x1 convolutions with at most 16 output channels: the RPN head (mmdet/models/dense_heads/rpn_head.py:54-68, `rpn_cls` +
// `rpn_reg`: 3 + 12 channels per pixel of a 256-channel map), forward, data gradient and weight gradient.
//
// Rounds 1-3 ran the fused head as ONE 128-output-channel convolution on the tile kernels (output channels zero-padded
// to their tile): 15 live channels in 128 written and re-read by every pass - the head's output (358 MB per step), its
// gradient (written by the loss, read by the data gradient, the weight gradient and the bias-gradient pass).  These
// kernels keep the head's maps 16 channels wide (one 32-byte row per pixel): 1.56 GB less traffic per step.
//
// All three are streaming kernels around v_mfma_f32_16x16x32_bf16 with the 16 head channels on one MFMA dimension:
//   forward        D[k][px]  = sum_c  w[k][c]      x[px][c]     A = w   (registers, whole launch), B = 16 pixels from HBM
//   data gradient  D[c][px]  = sum_k  wT[c][k]     dy[px][k]    A = w^T (registers), B = dy rows (k padded 16 -> 32 by zeros)
//   weight grad.   D[k][c]  += sum_px dy[px][k]    x[px][c]     both operands pixel-major in HBM: staged as they lie (LDS-DMA)
//                                                               and transposed by ds_read_b64_tr_b16, as conv_wgrad_kernel does
// The data gradient carries the epilogue of the tile kernels' data-gradient launches (hip_conv.GradToken): ReLU mask of the
// tensor it flows into (mask BITS), column sums of what it stores (that tensor's producer's bias gradient).
#include "common.h"
#include "oadg_hip.h"

namespace {

typedef float f32x4n __attribute__((ext_vector_type(4)));
typedef short v4sn __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4sn* lds_v4sn_ptr;
typedef __attribute__((address_space(3))) void* lds_ptr_n;
typedef const __attribute__((address_space(1))) void* gbl_ptr_n;

__device__ __forceinline__ void glds16n(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((gbl_ptr_n)g, (lds_ptr_n)l, 16, 0, 0);
}

#define ZERO8 (bf16x8{0, 0, 0, 0, 0, 0, 0, 0})

// ---------------------------------------------------------------------------------------------------------------- forward
// The reduction index of MFMA ks is a permutation of the channels (the same for both operands): lane quarter fq takes the
// 8 channels at 64 (ks / 2) + 16 fq + 8 (ks % 2), so that a lane's two loads of an MFMA pair are 32 contiguous bytes and
// the four quarters of a pixel cover one 128-byte line per pair (instead of 64 bytes of two lines).
__device__ __forceinline__ int kofs(int ks, int fq) { return 64 * (ks >> 1) + 16 * fq + 8 * (ks & 1); }

// One wave = one 16-pixel tile per trip (grid-stride: the chip streams through one compact window of x); the 8 (C = 256)
// fragment loads of a tile are independent 16-byte loads, two tiles in flight per wave.
template <int C>
__global__ __launch_bounds__(256) void n16_fwd_kernel(const unsigned short* __restrict__ x,
                                                      const unsigned short* __restrict__ w16,
                                                      const float* __restrict__ bias16, unsigned short* __restrict__ y,
                                                      long M) {
    constexpr int KS = C / 32;
    const int lane = threadIdx.x & 63, fr = lane & 15, fq = lane >> 4;
    const long gw = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
    bf16x8 a[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a[ks] = *reinterpret_cast<const bf16x8*>(w16 + (size_t)fr * C + kofs(ks, fq));
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = bias16 ? bias16[4 * fq + r] : 0.f;
    const long n_tiles = (M + 15) >> 4;
    for (long t = gw; t < n_tiles; t += 2 * nw) {
        bf16x8 b[2][KS];
        long p[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            p[u] = (t + u * nw) * 16 + fr;
            const bool ok = (t + u * nw) < n_tiles && p[u] < M;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                b[u][ks] = ok ? *reinterpret_cast<const bf16x8*>(x + (size_t)p[u] * C + kofs(ks, fq)) : ZERO8;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if ((t + u * nw) >= n_tiles) break;
            f32x4n acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks], b[u][ks], acc, 0, 0, 0);
            if (p[u] < M) {
                uint2 pk;
                pk.x = (unsigned)f32_to_bf16(acc[0] + bv[0]) | ((unsigned)f32_to_bf16(acc[1] + bv[1]) << 16);
                pk.y = (unsigned)f32_to_bf16(acc[2] + bv[2]) | ((unsigned)f32_to_bf16(acc[3] + bv[3]) << 16);
                *reinterpret_cast<uint2*>(y + (size_t)p[u] * 16 + 4 * fq) = pk;        // 32-byte pixel rows: 512 B per wave
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------- data gradient
// Per 16-pixel tile one 16-byte load of dy per lane (k-groups 2, 3 are the zero padding of the reduction), C / 16 MFMAs
// against the register-resident w^T fragments.  The A rows of the four MFMAs of a 64-channel quad are a permutation of
// its channels - row 4 fq + r of MFMA j is channel 64 q + 32 (j / 2) + 8 fq + 4 (j % 2) + r - so that a lane ends up with
// two runs of 8 consecutive channels of its pixel per quad: two 16-byte stores, in each of which the four quarters of a
// pixel write 64 contiguous bytes (the whole 128-byte line between them).
template <int C>
__global__ __launch_bounds__(256) void n16_dgrad_kernel(const unsigned short* __restrict__ dy,
                                                        const unsigned short* __restrict__ wt,
                                                        unsigned short* __restrict__ dx,
                                                        const unsigned char* __restrict__ bits,
                                                        float* __restrict__ colsum_part, long M) {
    constexpr int NQ = C / 64, NW32 = C / 32;
    __shared__ float red[4][C];
    const int lane = threadIdx.x & 63, fr = lane & 15, fq = lane >> 4, wave = threadIdx.x >> 6;
    const long gw = (long)blockIdx.x * 4 + wave, nw = (long)gridDim.x * 4;
    bf16x8 a[NQ][4];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            a[q][j] = fq < 2 ? *reinterpret_cast<const bf16x8*>(wt + (size_t)(64 * q + 32 * (j >> 1) + 8 * (fr >> 2) + 4 * (j & 1) + (fr & 3)) * 16 + fq * 8)
                             : ZERO8;
    float csum[NQ][16];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int e = 0; e < 16; ++e) csum[q][e] = 0.f;
    const long n_tiles = (M + 15) >> 4;
    // one tile ahead: the next tile's dy row and mask words are in flight while this one is multiplied and stored
    auto fetch = [&](long t, bf16x8& b, unsigned (&mb)[NW32]) {
        const long p = t * 16 + fr;
        const bool ok = t < n_tiles && p < M;
        b = (ok && fq < 2) ? *reinterpret_cast<const bf16x8*>(dy + (size_t)p * 16 + fq * 8) : ZERO8;
        if (bits) {
#pragma unroll
            for (int j = 0; j < NW32; j += 4) {
                const uint4 v = ok ? *reinterpret_cast<const uint4*>(bits + (size_t)p * (C / 8) + j * 4) : make_uint4(0, 0, 0, 0);
                mb[j] = v.x; mb[j + 1] = v.y; mb[j + 2] = v.z; mb[j + 3] = v.w;
            }
        }
    };
    bf16x8 b, bn;
    unsigned mb[NW32], mbn[NW32];
    fetch(gw, bn, mbn);
    for (long t = gw; t < n_tiles; t += nw) {
        const long p = t * 16 + fr;
        const bool ok = p < M;
        b = bn;
#pragma unroll
        for (int j = 0; j < NW32; ++j) mb[j] = mbn[j];
        fetch(t + nw, bn, mbn);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            unsigned short o[16];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4n acc = {0.f, 0.f, 0.f, 0.f};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[q][j], b, acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) o[4 * j + r] = f32_to_bf16(acc[r]);
            }
            if (bits) {
                // bit e of byte j = channel 8 j + e (little-endian words): o[8 h ..] = channels 64 q + 32 h + 8 fq .. + 7 are
                // byte fq of word 2 q + h
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const unsigned byte = (mb[2 * q + h] >> (8 * fq)) & 0xffu;
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (!((byte >> e) & 1u)) o[8 * h + e] = 0;
                }
            }
            if (ok) {
#pragma unroll
                for (int e = 0; e < 16; ++e) csum[q][e] += bf16_to_f32(o[e]);
                uint4 pk[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    pk[h].x = (unsigned)o[8 * h + 0] | ((unsigned)o[8 * h + 1] << 16);
                    pk[h].y = (unsigned)o[8 * h + 2] | ((unsigned)o[8 * h + 3] << 16);
                    pk[h].z = (unsigned)o[8 * h + 4] | ((unsigned)o[8 * h + 5] << 16);
                    pk[h].w = (unsigned)o[8 * h + 6] | ((unsigned)o[8 * h + 7] << 16);
                }
                uint4* dst = reinterpret_cast<uint4*>(dx + (size_t)p * C + 64 * q + 8 * fq);
                dst[0] = pk[0];                 // (the four quarters of a pixel: 64 contiguous bytes per store)
                dst[4] = pk[1];
            }
        }
    }
    if (colsum_part) {          // one row of partial column sums per workgroup (fixed order: deterministic)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = csum[q][e];           // sum over the 16 pixels of a lane row: rotations within the DPP row
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
                v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
                if (fr == 0) red[wave][64 * q + 32 * (e >> 3) + 8 * fq + (e & 7)] = v;
            }
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 256)
            colsum_part[(size_t)blockIdx.x * C + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    }
}

// -------------------------------------------------------------------------------------------------------- weight gradient
// part[split][k][c] = sum over the split's pixels of dy[px][k] x[px][c]; bpart[split][k] = sum of dy[px][k] (bias gradient).
// 64-pixel chunks, two LDS stages of (x tile [64][C] + dy tile [64][16]); wave w owns the channel groups [w C/64, (w+1) C/64).
constexpr int NWP = 64;
template <int C>
__global__ __launch_bounds__(256) void n16_wgrad_kernel(const unsigned short* __restrict__ x,
                                                        const unsigned short* __restrict__ dy, float* __restrict__ part,
                                                        float* __restrict__ bpart, const unsigned short* __restrict__ zeros,
                                                        long M, int chunks_per_split) {
    constexpr int XB = NWP * C * 2, DB = NWP * 16 * 2, STAGE = XB + DB, SPR = C / 8, NX = XB / 4096, CGW = C / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_n[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long nchunks = (M + NWP - 1) / NWP;
    const long ch0 = (long)blockIdx.x * chunks_per_split;
    const long ch1 = ch0 + chunks_per_split < nchunks ? ch0 + chunks_per_split : nchunks;
    // loader geometry: x piece q = i * 256 + tid -> row q / SPR, 16-byte slot q % SPR (source slot swizzled by the row)
    int xrow[NX], xsrc[NX];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int q = i * 256 + tid;
        xrow[i] = q / SPR;
        xsrc[i] = xrow[i] * C + (((q % SPR) ^ ((xrow[i] & 3) << 2)) * 8);
    }
    auto stage = [&](long ch, int buf) {
        unsigned char* sx = smem_n + buf * STAGE;
        unsigned char* sd = sx + XB;
        const long p0 = ch * NWP;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const bool ok = p0 + xrow[i] < M;
            glds16n(ok ? (const void*)(x + (size_t)p0 * C + xsrc[i]) : (const void*)zeros, sx + i * 4096 + wave * 1024);
        }
        if (wave < 2) {                       // dy tile: 64 rows x 32 bytes = two 1-KiB LDS-DMA instructions
            const int row = wave * 32 + (lane >> 1);
            const bool ok = p0 + row < M;
            glds16n(ok ? (const void*)(dy + (size_t)(p0 + row) * 16 + (lane & 1) * 8) : (const void*)zeros, sd + wave * 1024);
        }
    };
    // fragment addresses (ds_read_b64_tr_b16: lane s of a 16-lane group supplies {row s >> 2, columns 4 (s & 3) ..} of a
    // 4 x 16 block and receives column s): pixel row = ks * 32 + fq * 8 + u * 4 + (s >> 2)
    const int s16 = lane & 15, fq = lane >> 4;
    const int rowb = fq * 8 + (s16 >> 2);
    const int dyoff = rowb * 32 + (s16 & 3) * 8;
    int xoff[CGW];
#pragma unroll
    for (int j = 0; j < CGW; ++j) {
        const int col = (wave * CGW + j) * 16 + (s16 & 3) * 4;
        xoff[j] = rowb * (C * 2) + ((((col >> 3)) ^ ((rowb & 3) << 2)) << 4) + (col & 7) * 2;      // (rowb + 4 u: same row & 3)
    }
    auto frag = [&](const unsigned char* base, int off, int ks, int rowbytes) {
        bf16x8 out;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const v4sn r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4sn_ptr)(base + off + (ks * 32 + u * 4) * rowbytes));
            out[u * 4 + 0] = r[0]; out[u * 4 + 1] = r[1]; out[u * 4 + 2] = r[2]; out[u * 4 + 3] = r[3];
        }
        return out;
    };
    f32x4n acc[CGW];
#pragma unroll
    for (int j = 0; j < CGW; ++j) acc[j] = f32x4n{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    if (ch0 < ch1) {
        stage(ch0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (long ch = ch0; ch < ch1; ++ch) {
            const int cur = (int)((ch - ch0) & 1);
            if (ch + 1 < ch1) stage(ch + 1, cur ^ 1);
            const unsigned char* sx = smem_n + cur * STAGE;
            const unsigned char* sd = sx + XB;
#pragma unroll
            for (int ks = 0; ks < NWP / 32; ++ks) {
                const bf16x8 fa = frag(sd, dyoff, ks, 32);
                if (wave == 0) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) bsum += bf16_to_f32((unsigned short)fa[e]);
                }
#pragma unroll
                for (int j = 0; j < CGW; ++j) {
                    const bf16x8 fb = frag(sx, xoff[j], ks, C * 2);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc[j], 0, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    // D rows = k (4 fq + r), columns = c
    float* out = part + (size_t)blockIdx.x * 16 * C;
#pragma unroll
    for (int j = 0; j < CGW; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) out[(size_t)(4 * fq + r) * C + (wave * CGW + j) * 16 + s16] = acc[j][r];
    if (wave == 0) {            // lane (k = s16, pixel group fq) -> sum over the four pixel groups
        float v = bsum;
        v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
        if (fq == 0) bpart[(size_t)blockIdx.x * 16 + s16] = v;
    }
}

long n16_splits(long M) {
    const long nchunks = (M + NWP - 1) / NWP;
    long s = nchunks / 8;                       // at least 8 chunks per workgroup
    if (s > 512) s = 512;
    if (s < 1) s = 1;
    return s;
}

}  // namespace

extern "C" {

int oadg_conv1x1_n16_fwd(const void* x, const void* w16, const float* bias16, void* y, long M, int C, void* stream) {
    if (!x || !w16 || !y || M < 1) return OADG_EARG;
    const long tiles = (M + 15) / 16;
    const unsigned grid = (unsigned)(tiles / 8 < 2048 ? (tiles + 7) / 8 : 2048);
    hipStream_t st = (hipStream_t)stream;
    if (C == 256)
        hipLaunchKernelGGL(n16_fwd_kernel<256>, dim3(grid), dim3(256), 0, st, (const unsigned short*)x,
                           (const unsigned short*)w16, bias16, (unsigned short*)y, M);
    else if (C == 128)
        hipLaunchKernelGGL(n16_fwd_kernel<128>, dim3(grid), dim3(256), 0, st, (const unsigned short*)x,
                           (const unsigned short*)w16, bias16, (unsigned short*)y, M);
    else
        return OADG_EARG;
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

long oadg_conv1x1_n16_dgrad_rows(long M) {
    const long tiles = (M + 15) / 16;           // (165 registers: two workgroups per CU resident - one round of 512)
    return tiles / 4 < 512 ? (tiles + 3) / 4 : 512;
}

int oadg_conv1x1_n16_dgrad(const void* dy, const void* wt, void* dx, const void* mask_bits, float* colsum_part, long M, int C,
                           void* stream) {
    if (!dy || !wt || !dx || M < 1) return OADG_EARG;
    const unsigned grid = (unsigned)oadg_conv1x1_n16_dgrad_rows(M);
    hipStream_t st = (hipStream_t)stream;
    if (C == 256)
        hipLaunchKernelGGL(n16_dgrad_kernel<256>, dim3(grid), dim3(256), 0, st, (const unsigned short*)dy,
                           (const unsigned short*)wt, (unsigned short*)dx, (const unsigned char*)mask_bits, colsum_part, M);
    else if (C == 128)
        hipLaunchKernelGGL(n16_dgrad_kernel<128>, dim3(grid), dim3(256), 0, st, (const unsigned short*)dy,
                           (const unsigned short*)wt, (unsigned short*)dx, (const unsigned char*)mask_bits, colsum_part, M);
    else
        return OADG_EARG;
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

long oadg_conv1x1_n16_wgrad_splits(long M) { return M < 1 ? 0 : n16_splits(M); }

int oadg_conv1x1_n16_wgrad(const void* x, const void* dy, float* part, float* bias_part, const void* zeros16, long M, int C,
                           void* stream) {
    if (!x || !dy || !part || !bias_part || !zeros16 || M < 1) return OADG_EARG;
    const long splits = n16_splits(M), nchunks = (M + NWP - 1) / NWP;
    const int cps = (int)((nchunks + splits - 1) / splits);
    const unsigned grid = (unsigned)((nchunks + cps - 1) / cps);
    if ((long)grid != splits && (long)grid > splits) return OADG_EARG;
    hipStream_t st = (hipStream_t)stream;
#define OADG_N16W(CC)                                                                                                     \
    do {                                                                                                                  \
        constexpr int LDS = 2 * (NWP * CC * 2 + NWP * 32);                                                                \
        static bool attr = false;                                                                                         \
        if (!attr) {                                                                                                      \
            hipError_t e = hipFuncSetAttribute((const void*)n16_wgrad_kernel<CC>,                                         \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, LDS);                          \
            if (e != hipSuccess) return (int)e;                                                                           \
            attr = true;                                                                                                  \
        }                                                                                                                 \
        hipLaunchKernelGGL(n16_wgrad_kernel<CC>, dim3(grid), dim3(256), LDS, st, (const unsigned short*)x,                \
                           (const unsigned short*)dy, part, bias_part, (const unsigned short*)zeros16, M, cps);           \
    } while (0)
    if (C == 256) OADG_N16W(256);
    else if (C == 128) OADG_N16W(128);
    else return OADG_EARG;
#undef OADG_N16W
    OADG_LAUNCH_CHECK();
    // (workgroups past the last chunk do not exist: grid <= splits; the caller sums the first `grid` rows - see
    //  oadg_conv1x1_n16_wgrad_rows)
    return OADG_OK;
}

long oadg_conv1x1_n16_wgrad_rows(long M) {
    if (M < 1) return 0;
    const long splits = n16_splits(M), nchunks = (M + NWP - 1) / NWP;
    const long cps = (nchunks + splits - 1) / splits;
    return (nchunks + cps - 1) / cps;
}

}  // extern "C"
