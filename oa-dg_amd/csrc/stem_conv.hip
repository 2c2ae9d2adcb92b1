// ResNet stem convolution (mmdet/models/backbones/resnet.py:585-596 `conv1`: 7x7, stride 2, padding 3, 3 -> 64
// channels; called at :631-637) on the matrix cores, for the NHWC bf16 images the device pipeline produces.
//
// Implicit GEMM with the reduction index k = (r, j, c): filter row r (7), j = 8 input pixels of that row starting one
// pixel left of the filter (pixel 0 carries a zero weight), c = 4 channels per pixel (the image's 3 + one zero) ->
// K = 7 x 32, exactly one v_mfma_f32_16x16x32_bf16 per filter row.  The input patch of a workgroup is staged in LDS as
// 4-channel (8-byte) pixels, so the 8 k-values of a lane (2 neighbouring pixels) are ONE aligned ds_read_b128:
//   B[k-group g][pixel p] = patch[row 2*ho + r][column 2*wo + 2g .. 2g+1],   A[channel][k-group g] = prepared weights.
// A workgroup (4 waves) computes 4 output rows x 64 output columns x 64 channels; a wave owns one output row
// (4 pixel tiles x 4 channel tiles = 16 accumulators).  The weights (64 x 224 bf16 = 28 KiB) stay in registers.
// Output: round_bf16(conv) without bias - the BN shift, ReLU and max-pool follow in bias_relu_maxpool_kernel - staged
// through LDS so that every store instruction writes whole 128-byte pixel rows.
#include "common.h"
#include "../../include/oadg_hip.h"

namespace {

constexpr int ST_TH = 4, ST_TW = 64;                       // output tile (rows x columns) per workgroup
constexpr int ST_PR = (ST_TH - 1) * 2 + 7;                 // 13 patch rows
constexpr int ST_PC = 136;                                 // patch columns: 2*63 + 8 = 134, padded to 136 pixels
constexpr int ST_ROWB = ST_PC * 8;                         // bytes per patch row (8-byte pixels)

typedef float f32x4v __attribute__((ext_vector_type(4)));

template <bool EVENW>
__global__ __launch_bounds__(256) void stem_conv7x7s2_kernel(const unsigned short* __restrict__ x,
                                                            const unsigned short* __restrict__ wp,
                                                            unsigned short* __restrict__ y, int N, int H, int W,
                                                            int Ho, int Wo) {
    __shared__ __attribute__((aligned(16))) unsigned char patch[ST_PR * ST_ROWB];      // 14,144 B
    __shared__ __attribute__((aligned(16))) unsigned short otile[4][16 * 64];          // per wave: 16 pixels x 64 ch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wo0 = blockIdx.x * ST_TW, ho0 = blockIdx.y * ST_TH, n = blockIdx.z;
    // ---- stage: image rows 2*ho0 - 3 .. +12, pixels 2*wo0 - 4 .. +133 (3 channels, 6 bytes) -> 8-byte LDS pixels.
    // Item i = (patch row, dword d of the row segment) holds elements 2d, 2d+1 (element e = pixel e / 3, channel e % 3).
    // W even: rows and the segment start are dword aligned, so a dword lies entirely inside or outside the image row and
    // one predicated 4-byte load per item does; W odd: two 2-byte loads.  All loads of a thread are issued before the
    // first LDS write (no control flow between them).
    constexpr int NDW = (134 * 3 + 1) / 2;            // 201 dwords per row
    constexpr int NIT = (ST_PR * NDW + 255) / 256;    // 11 items per thread
    const int px0 = 2 * wo0 - 4;
    unsigned v[NIT];
    unsigned ok[NIT];                                  // bit 0 / 1: low / high element inside the image
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = it * 256 + tid;
        const int pr = i / NDW, d = i - pr * NDW;
        const int row = 2 * ho0 - 3 + pr;
        const bool rowok = i < ST_PR * NDW && (unsigned)row < (unsigned)H;
        const int e0 = 2 * d, p0 = e0 / 3, p1 = (e0 + 1) / 3;
        const bool in0 = rowok && (unsigned)(px0 + p0) < (unsigned)W;
        const bool in1 = rowok && (unsigned)(px0 + p1) < (unsigned)W;
        const long base = ((long)n * H + row) * W * 3 + (long)px0 * 3 + e0;
        if (EVENW) {
            v[it] = in0 ? *reinterpret_cast<const unsigned*>(x + base) : 0u;
            ok[it] = in0 ? 3u : 0u;
        } else {
            const unsigned lo = in0 ? x[base] : 0u, hi = in1 ? x[base + 1] : 0u;
            v[it] = lo | (hi << 16);
            ok[it] = (in0 ? 1u : 0u) | (in1 ? 2u : 0u);
        }
    }
    // ---- zero the patch (channel 3 of every pixel and everything outside the image stay zero), then scatter
    for (int i = tid; i < ST_PR * ST_ROWB / 16; i += 256) reinterpret_cast<uint4*>(patch)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    {
        unsigned short* p16 = reinterpret_cast<unsigned short*>(patch);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = it * 256 + tid;
            const int pr = i / NDW, d = i - pr * NDW;
            const int e0 = 2 * d, p0 = e0 / 3, c0 = e0 - p0 * 3, p1 = (e0 + 1) / 3, c1 = (e0 + 1) - p1 * 3;
            if (ok[it] & 1u) p16[pr * (ST_ROWB / 2) + p0 * 4 + c0] = (unsigned short)(v[it] & 0xffffu);
            if (ok[it] & 2u) p16[pr * (ST_ROWB / 2) + p1 * 4 + c1] = (unsigned short)(v[it] >> 16);
        }
    }
    // ---- weights: A[channel = nt*16 + (lane & 15)][k = r*32 + (lane >> 4)*8 ..] = 8 contiguous bf16 of wp [64][7][32]
    bf16x8 wfrag[7][4];
#pragma unroll
    for (int r = 0; r < 7; ++r)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
            wfrag[r][nt] = *reinterpret_cast<const bf16x8*>(wp + ((size_t)(nt * 16 + (lane & 15)) * 7 + r) * 32 +
                                                           (lane >> 4) * 8);
    __syncthreads();
    f32x4v acc[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = f32x4v{0.f, 0.f, 0.f, 0.f};
    const int colb = (2 * (lane & 15) + 2 * (lane >> 4)) * 8;           // byte offset of this lane's 2 pixels in a tile
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        const unsigned char* prow = patch + (2 * wave + r) * ST_ROWB + colb;
        bf16x8 pf[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) pf[mt] = *reinterpret_cast<const bf16x8*>(prow + mt * 32 * 8);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfrag[r][nt], pf[mt], acc[mt][nt], 0, 0, 0);
    }
    // ---- epilogue: lane holds channels nt*16 + (lane >> 4)*4 + i of pixel mt*16 + (lane & 15)
    const int ho = ho0 + wave;
    unsigned short* ot = otile[wave];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            uint2 pk;
            pk.x = (unsigned)f32_to_bf16(acc[mt][nt][0]) | ((unsigned)f32_to_bf16(acc[mt][nt][1]) << 16);
            pk.y = (unsigned)f32_to_bf16(acc[mt][nt][2]) | ((unsigned)f32_to_bf16(acc[mt][nt][3]) << 16);
            // 16-byte slot (8 channels) XOR (pixel & 7): the 16 pixels of a lane group spread over the banks
            const int slot = (nt * 2 + (lane >> 5)) ^ (lane & 7);
            *reinterpret_cast<uint2*>(ot + (lane & 15) * 64 + slot * 8 + ((lane >> 4) & 1) * 4) = pk;
        }
        // wave-private tile: only this wave's lanes wrote it (LDS operations of a wave complete in order)
        if (ho < Ho) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int q = h * 64 + lane, p = q >> 3, sg = q & 7;        // pixel p of the tile, 16-byte slot sg
                const int wo = wo0 + mt * 16 + p;
                if (wo < Wo)
                    *reinterpret_cast<uint4*>(y + (((size_t)n * Ho + ho) * Wo + wo) * 64 + sg * 8) =
                        *reinterpret_cast<const uint4*>(ot + p * 64 + ((sg ^ (p & 7)) * 8));
            }
        }
    }
}

}  // namespace

extern "C" int oadg_stem_conv7x7s2_nhwc_bf16(const void* x, const void* wp, void* y, int N, int H, int W, void* stream) {
    if (!x || !wp || !y || N < 1 || H < 1 || W < 1) return OADG_EARG;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    if ((long)N * H * W * 3 >= (1L << 31)) return OADG_EARG;
    dim3 grid((Wo + ST_TW - 1) / ST_TW, (Ho + ST_TH - 1) / ST_TH, N);
    if (W & 1)
        hipLaunchKernelGGL(stem_conv7x7s2_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream,
                           (const unsigned short*)x, (const unsigned short*)wp, (unsigned short*)y, N, H, W, Ho, Wo);
    else
        hipLaunchKernelGGL(stem_conv7x7s2_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream,
                           (const unsigned short*)x, (const unsigned short*)wp, (unsigned short*)y, N, H, W, Ho, Wo);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}
