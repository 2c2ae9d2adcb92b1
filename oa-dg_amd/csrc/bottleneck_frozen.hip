// A FROZEN ResNet bottleneck block as ONE kernel (mmdet/models/backbones/resnet.py:263-302 `Bottleneck.forward` with
// `_freeze_stages` :613-628 - stage 1 of the OA-DG configs: frozen_stages = 1, BN in eval mode and folded):
//     y = relu( conv3( relu(conv2( relu(conv1(x) + b1) ) + b2) ) + b3 + x ),   conv1 1x1 256 -> 64, conv2 3x3 64 -> 64,
//     conv3 1x1 64 -> 256, identity shortcut, NHWC bf16, fp32 accumulation, bf16 rounding where the three-kernel path
//     rounds (after every bias add and after the residual add).
// The three launches of the unfused path move 2.15 GB per block at 256 x 512 x 8 images (x read by conv1 AND as conv3's
// residual, the 64-channel maps written and re-read twice) at 4.3 - 5.3 TB/s; nothing of a frozen block is needed again
// (no backward pass), so the fused kernel reads x once and writes y once: 1.07 GB.
//
// One workgroup (8 waves) per 16 x 16 output tile, persistent over the tiles of its XCD:
//   stage 1  conv1 on the 18 x 18 halo (324 pixels = 21 column tiles of the MFMA): B operands straight from HBM in fragment
//            layout (channels permuted so that the four lane quarters of a pixel read whole 128-byte lines), A = w1 from
//            LDS; + b1, ReLU, ZERO outside the image (conv2's zero padding) -> t1 [halo pixel][64] bf16 in LDS;
//   stage 2  conv2 as 9 taps x 2 k-steps of v_mfma_f32_16x16x32_bf16 per output row (one column tile = one row of the
//            tile: the tap shift is an LDS address offset); wave = (16 output channels, 8 rows), its 18 weight fragments
//            live in registers for the whole launch; -> t2 [pixel][64] bf16 in LDS;
//   stage 3  conv3: wave = (64 output channels, 8 rows); the A rows are a permutation of the channels so that a lane ends
//            up with two runs of 8 consecutive channels per pixel (16-byte stores, 64 contiguous bytes per pixel per
//            store); + b3, + x (L2: stage 1 has just read it), ReLU.
// LDS layouts (round 6): a ds_read_b128 is served in four groups of 16 lanes that are NOT the lane quarters - {0-3, 12-15,
// 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS) - so a group mixes 8 rows of one k-slot with 8
// other rows of the NEXT k-slot.  The 16-byte row padding of rounds 4-5 (stride 144 B = 36 banks) put 7 of those 16 lanes on
// a bank another lane of the group used: every fragment read took two LDS cycles (SQ_LDS_BANK_CONFLICT 47 % of the LDS
// cycles).  t1 / t2 rows are now 160 B apart (40 banks: the only paddings for which both group shapes are conflict-free for
// ANY first row - conv2's tap shifts make the first row arbitrary - are 32 / 96 bytes); the weight images, whose rows are
// always read 16-aligned, are XOR-swizzled without padding: w3 slot ^ ((row >> 1) & 7) in 128-byte rows (the 256-tile
// convolution's scheme), w1 (512-byte rows, lane quarters 32 bytes apart) slot ^ (row & 15).
#include "common.h"
#include "oadg_hip.h"

namespace {

typedef float f32x4q __attribute__((ext_vector_type(4)));

constexpr int FB_TS = 16, FB_HS = FB_TS + 2, FB_NH = FB_HS * FB_HS;        // tile side, halo side, halo pixels (324)
constexpr int FB_NHT = (FB_NH + 15) / 16;                                  // 21 column tiles of 16 halo pixels
constexpr int FB_C = 256, FB_MID = 64;
constexpr int FB_PSTR = FB_MID * 2 + 32;                                   // bytes per pixel row of t1 / t2 (see above)
constexpr int FB_W1STR = FB_C * 2;                                         // bytes per output-channel row of w1 (swizzled)
constexpr int FB_T1 = FB_NHT * 16 * FB_PSTR, FB_T2 = FB_TS * FB_TS * FB_PSTR, FB_W1 = FB_MID * FB_W1STR;
constexpr int FB_W3STR = FB_MID * 2;                                       // w3 rows: 64 k, swizzled
constexpr int FB_W3 = FB_C * FB_W3STR;
constexpr int FB_LDS = FB_T1 + FB_T2 + FB_W1 + FB_W3 + FB_C * 4;           // 161,280 bytes (+ b3 as floats)
static_assert(FB_LDS <= 160 * 1024, "frozen block: LDS");

struct FrozenBlockArgs {
    const unsigned short* x;
    const unsigned short *w1, *w2, *w3;     // [64][256], [64][3][3][64], [256][64] bf16 (BN folded)
    const float *b1, *b2, *b3;
    unsigned short* y;
    int N, H, W, tiles_x, tiles_y;
};

// channel offset of k-step ks (of 8) for lane quarter fq: see csrc/narrow_head.hip kofs()
__device__ __forceinline__ int fb_kofs(int ks, int fq) { return 64 * (ks >> 1) + 16 * fq + 8 * (ks & 1); }

__device__ __forceinline__ unsigned pack2(float a, float b) {
    return (unsigned)f32_to_bf16(a) | ((unsigned)f32_to_bf16(b) << 16);
}

__global__ __launch_bounds__(512) void bottleneck_frozen_kernel(FrozenBlockArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fb_smem[];
    unsigned char* t1 = fb_smem;
    unsigned char* t2 = fb_smem + FB_T1;
    unsigned char* w1s = fb_smem + FB_T1 + FB_T2;
    unsigned char* w3s = w1s + FB_W1;               // rows in the order stage 3 reads them (see there)
    float* b3s = reinterpret_cast<float*>(w3s + FB_W3);
    const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sub = wave & 3, half = wave >> 2;         // stage 2: 16 output channels `sub`; stage 3: 64 channels `sub`
    // ---- launch-long operands: w1 -> LDS, this wave's w2 fragments -> registers
    for (int i = tid; i < FB_MID * (FB_C / 8); i += 512) {
        const int row = i / (FB_C / 8), piece = i - row * (FB_C / 8);
        *reinterpret_cast<uint4*>(w1s + row * FB_W1STR + ((piece ^ (row & 15)) << 4)) =
            *reinterpret_cast<const uint4*>(a.w1 + (size_t)row * FB_C + piece * 8);
    }
    for (int i = tid; i < FB_C * (FB_MID / 8); i += 512) {
        // LDS row R = 64 s + 16 j + m holds the channel that row m of stage 3's MFMA j computes (a permutation inside each
        // group of 64 channels: a lane then owns two runs of 8 consecutive channels of its pixel)
        const int R = i / (FB_MID / 8), piece = i - R * (FB_MID / 8);
        const int s_ = R >> 6, j = (R >> 4) & 3, m = R & 15;
        const int ch = 64 * s_ + 32 * (j >> 1) + 8 * (m >> 2) + 4 * (j & 1) + (m & 3);
        *reinterpret_cast<uint4*>(w3s + R * FB_W3STR + ((piece ^ ((R >> 1) & 7)) << 4)) =
            *reinterpret_cast<const uint4*>(a.w3 + (size_t)ch * FB_MID + piece * 8);
    }
    if (tid < FB_C) b3s[tid] = a.b3[tid];
    bf16x8 w2f[9][2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            w2f[tap][ks] = *reinterpret_cast<const bf16x8*>(a.w2 + ((size_t)(16 * sub + fr) * 9 + tap) * FB_MID + 32 * ks + 8 * fq);
    float b2v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) b2v[r] = a.b2[16 * sub + 4 * fq + r];
    __syncthreads();

    // ---- persistent over the tiles of this workgroup's XCD (block b runs on XCD b % 8): at any moment the 32 workgroups
    //      of an XCD work on 32 consecutive tiles (one tile row of a 512-pixel-wide map): halo rows are shared in its L2
    const int total = a.N * a.tiles_y * a.tiles_x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int tper = (total + 7) / 8;
    // the first halo column tile of a wave's NEXT tile is requested right after the barrier that ends stage 1 and arrives
    // while stages 2 and 3 run (those stages touch HBM only through L2-resident residual rows and the stores)
    bf16x8 bx[2][8];
    bool valid[2] = {false, false};
    struct TileAt { int n, y0, x0; bool ok; };
    auto tile_at = [&](int it) {
        TileAt t;
        const int tile = xcd * tper + it;
        t.ok = it < tper && tile < total;
        const int per_img = a.tiles_y * a.tiles_x;
        t.n = tile / per_img;
        const int trem = tile - t.n * per_img, ty = trem / a.tiles_x;
        t.y0 = ty * FB_TS;
        t.x0 = (trem - ty * a.tiles_x) * FB_TS;
        return t;
    };
    auto fetch = [&](const TileAt& t, int j, int set) {
        const int hp = 16 * j + fr, hy = hp / FB_HS, hx = hp - hy * FB_HS;
        const int gy = t.y0 - 1 + hy, gx = t.x0 - 1 + hx;
        valid[set] = t.ok && hp < FB_NH && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
        const unsigned short* px = a.x + (((size_t)t.n * a.H + gy) * a.W + gx) * FB_C;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            bx[set][ks] = valid[set] ? *reinterpret_cast<const bf16x8*>(px + fb_kofs(ks, fq)) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    };
    fetch(tile_at(slot), wave, 0);
    for (int it = slot; it < tper; it += per_xcd) {
        const TileAt cur = tile_at(it);
        if (!cur.ok) break;
        const int n = cur.n, y0 = cur.y0, x0 = cur.x0;
        const unsigned short* xn = a.x + (size_t)n * a.H * a.W * FB_C;

        // ================================================================ stage 1: t1 = relu(conv1(x halo) + b1)
        {
            const int ntile = wave + 16 < FB_NHT ? 3 : 2;        // column tiles wave, wave + 8, wave + 16 (< 21)
            fetch(cur, wave + 8, 1);                             // (set 0 = column tile `wave`: requested a tile ago)
            float b1v[4][4];                                     // channels 16 rt + 4 fq + r of this lane
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) b1v[rt][r] = a.b1[16 * rt + 4 * fq + r];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                if (u >= ntile) break;
                const int set = u & 1, j = wave + 8 * u;
                f32x4q acc[4];
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) acc[rt] = f32x4q{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt) {
                        const bf16x8 wa = *reinterpret_cast<const bf16x8*>(
                            w1s + (16 * rt + fr) * FB_W1STR + (((fb_kofs(ks, fq) >> 3) ^ fr) << 4));
                        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, bx[set][ks], acc[rt], 0, 0, 0);
                    }
                    if (ks & 1) __builtin_amdgcn_sched_barrier(0);     // (bounds the LDS reads in flight: registers)
                }
                const bool ok = valid[set];
                const int hp = 16 * j + fr;
                if (u == 0 && ntile == 3) fetch(cur, wave + 16, 0);     // (set 0 is free again)
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    const int c0 = 16 * rt + 4 * fq;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = fmaxf(acc[rt][r] + b1v[rt][r], 0.f);
                        if (!ok) v[r] = 0.f;                             // outside the image: conv2's zero padding
                    }
                    uint2 pk;
                    pk.x = pack2(v[0], v[1]);
                    pk.y = pack2(v[2], v[3]);
                    *reinterpret_cast<uint2*>(t1 + hp * FB_PSTR + c0 * 2) = pk;
                }
            }
        }
        __syncthreads();
        fetch(tile_at(it + per_xcd), wave, 0);
        // ================================================================ stage 2: t2 = relu(conv2(t1) + b2)
#pragma unroll 1
        for (int cp = 0; cp < 4; ++cp) {                                  // two rows of the tile per trip
            const int py = 8 * half + 2 * cp;
            f32x4q acc[2] = {f32x4q{0.f, 0.f, 0.f, 0.f}, f32x4q{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const bf16x8 bv = *reinterpret_cast<const bf16x8*>(
                            t1 + ((py + q + dy) * FB_HS + fr + dx) * FB_PSTR + (32 * ks + 8 * fq) * 2);
                        acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[tap][ks], bv, acc[q], 0, 0, 0);
                    }
                }
                if (tap % 3 == 2) __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                uint2 pk;
                pk.x = pack2(fmaxf(acc[q][0] + b2v[0], 0.f), fmaxf(acc[q][1] + b2v[1], 0.f));
                pk.y = pack2(fmaxf(acc[q][2] + b2v[2], 0.f), fmaxf(acc[q][3] + b2v[3], 0.f));
                *reinterpret_cast<uint2*>(t2 + ((py + q) * FB_TS + fr) * FB_PSTR + (16 * sub + 4 * fq) * 2) = pk;
            }
        }
        __syncthreads();
        // ================================================================ stage 3: y = relu(conv3(t2) + b3 + x)
        {
#pragma unroll 1
            for (int pass = 0; pass < 4; ++pass) {                       // two rows of the tile per trip
                const int py0 = 8 * half + 2 * pass;
                uint4 res[2][2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const unsigned short* px = xn + ((size_t)(y0 + py0 + q) * a.W + x0 + fr) * FB_C + 64 * sub + 8 * fq;
                    const bool in = y0 + py0 + q < a.H && x0 + fr < a.W;       // (edge tiles of maps that are no multiple of 16)
                    res[q][0] = in ? *reinterpret_cast<const uint4*>(px) : make_uint4(0, 0, 0, 0);
                    res[q][1] = in ? *reinterpret_cast<const uint4*>(px + 32) : make_uint4(0, 0, 0, 0);
                }
                f32x4q acc[2][4];
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[q][j] = f32x4q{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const bf16x8 bv = *reinterpret_cast<const bf16x8*>(
                            t2 + ((py0 + q) * FB_TS + fr) * FB_PSTR + (32 * ks + 8 * fq) * 2);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int wrow = 64 * sub + 16 * j + fr;
                            const bf16x8 wa = *reinterpret_cast<const bf16x8*>(
                                w3s + wrow * FB_W3STR + (((4 * ks + fq) ^ ((wrow >> 1) & 7)) << 4));
                            acc[q][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, bv, acc[q][j], 0, 0, 0);
                        }
                    }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    unsigned short* py_ = a.y + ((size_t)n * a.H * a.W + (size_t)(y0 + py0 + q) * a.W + x0 + fr) * FB_C +
                                          64 * sub + 8 * fq;
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const unsigned rw[4] = {res[q][h].x, res[q][h].y, res[q][h].z, res[q][h].w};
                        const float4 ba = *reinterpret_cast<const float4*>(b3s + 64 * sub + 32 * h + 8 * fq);
                        const float4 bb = *reinterpret_cast<const float4*>(b3s + 64 * sub + 32 * h + 8 * fq + 4);
                        const float b3e[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w};
                        float o[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            // the three-kernel path rounds conv3 + b3 to bf16 before the residual add: so does this
                            const float t = bf16_to_f32(f32_to_bf16(acc[q][2 * h + (e >> 2)][e & 3] + b3e[e]));
                            const float rv = bf16_to_f32((unsigned short)(rw[e >> 1] >> ((e & 1) * 16)));
                            o[e] = fmaxf(t + rv, 0.f);
                        }
                        uint4 pk;
                        pk.x = pack2(o[0], o[1]); pk.y = pack2(o[2], o[3]); pk.z = pack2(o[4], o[5]); pk.w = pack2(o[6], o[7]);
                        if (y0 + py0 + q < a.H && x0 + fr < a.W) *reinterpret_cast<uint4*>(py_ + 32 * h) = pk;
                    }
                }
            }
        }
        // (no barrier here: the next tile's stage 1 writes t1, which nobody reads after the barrier above; its stage 2
        //  writes t2 only after the barrier that follows stage 1, i.e. after every wave has left this stage 3)
    }
}


// ---------------------------------------------------------------------------------------------------- the stage's FIRST block
// 64 -> 64 -> 64 -> 256 with the 1x1 downsample convolution on the shortcut (resnet.py:631-646 on the max-pool output):
//     y = relu( bf16(conv3(relu(conv2(relu(conv1(x) + b1)) + b2)) + b3) + bf16(convd(x) + bd) )
// The four launches move 2.55 GB at 8 x 256 x 512 (x read three times, three 64-channel maps and the downsample map
// written and re-read); fused: x read once (134 MB), y written once (537 MB).  Same tile scheme; the halo of x (64
// channels) is staged in LDS (t0) - stage 1 and the downsample MFMAs of stage 3 read it there - and the next tile's halo
// pieces are requested into registers while stages 2 and 3 run.
constexpr int FF_W1 = FB_MID * FB_PSTR;                                    // w1 [64][64] rows padded like t1
constexpr int FF_LDS = 2 * FB_T1 + FB_T2 + FF_W1 + 2 * FB_C * 4;           // t0, t1, t2, w1, b3, bd: 160,768 bytes
static_assert(FF_LDS <= 160 * 1024, "frozen first block: LDS");
constexpr int FF_NP = (FB_NH * 8 + 511) / 512;                             // 16-byte halo pieces per thread (6)

struct FrozenFirstArgs {
    const unsigned short* x;                 // [N][H][W][64]
    const unsigned short *w1, *w2, *w3, *wd; // [64][64], [64][3][3][64], [256][64], [256][64] bf16 (BN folded)
    const float *b1, *b2, *b3, *bd;
    unsigned short* y;                       // [N][H][W][256]
    int N, H, W, tiles_x, tiles_y;
};

__global__ __launch_bounds__(512) void bottleneck_frozen_first_kernel(FrozenFirstArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fb_smem[];
    unsigned char* t0 = fb_smem;                    // the halo of x of the current tile ...
    unsigned char* t1 = fb_smem + FB_T1;            // ... conv1's output; the two buffers swap roles every tile (below)
    unsigned char* t2 = fb_smem + 2 * FB_T1;
    unsigned char* w1s = t2 + FB_T2;
    float* b3s = reinterpret_cast<float*>(w1s + FF_W1);
    float* bds = b3s + FB_C;
    const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sub = wave & 3, half = wave >> 2;
    for (int i = tid; i < FB_MID * (FB_MID / 8); i += 512) {
        const int row = i / (FB_MID / 8), piece = i - row * (FB_MID / 8);
        *reinterpret_cast<uint4*>(w1s + row * FB_PSTR + piece * 16) =
            *reinterpret_cast<const uint4*>(a.w1 + (size_t)row * FB_MID + piece * 8);
    }
    if (tid < FB_C) { b3s[tid] = a.b3[tid]; bds[tid] = a.bd[tid]; }
    for (int i = tid; i < (FB_NHT * 16 - FB_NH) * (FB_PSTR / 16); i += 512)      // the 12 padding rows of the first t0
        *reinterpret_cast<uint4*>(t0 + FB_NH * FB_PSTR + i * 16) = make_uint4(0, 0, 0, 0);
    bf16x8 w2f[9][2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            w2f[tap][ks] = *reinterpret_cast<const bf16x8*>(a.w2 + ((size_t)(16 * sub + fr) * 9 + tap) * FB_MID + 32 * ks + 8 * fq);
    bf16x8 w3f[4][2], wdf[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ch = 64 * sub + 32 * (j >> 1) + 8 * (fr >> 2) + 4 * (j & 1) + (fr & 3);     // A row fr of stage 3's MFMA j
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            w3f[j][ks] = *reinterpret_cast<const bf16x8*>(a.w3 + (size_t)ch * FB_MID + 32 * ks + 8 * fq);
            wdf[j][ks] = *reinterpret_cast<const bf16x8*>(a.wd + (size_t)ch * FB_MID + 32 * ks + 8 * fq);
        }
    }
    float b2v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) b2v[r] = a.b2[16 * sub + 4 * fq + r];

    const int total = a.N * a.tiles_y * a.tiles_x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
    const int tper = (total + 7) / 8;
    struct TileAt { int n, y0, x0; bool ok; };
    auto tile_at = [&](int it) {
        TileAt t;
        const int tile = xcd * tper + it;
        t.ok = it < tper && tile < total;
        const int per_img = a.tiles_y * a.tiles_x;
        t.n = tile / per_img;
        const int trem = tile - t.n * per_img, ty = trem / a.tiles_x;
        t.y0 = ty * FB_TS;
        t.x0 = (trem - ty * a.tiles_x) * FB_TS;
        return t;
    };
    // halo pieces of a tile: piece q = hp * 8 + slot of 8 channels -> registers (zeros outside the image)
    uint4 hx[FF_NP];
    auto fetch = [&](const TileAt& t) {
#pragma unroll
        for (int u = 0; u < FF_NP; ++u) {
            const int q = u * 512 + tid, hp = q >> 3, sl = q & 7;
            const int hy = hp / FB_HS, hxx = hp - hy * FB_HS;
            const int gy = t.y0 - 1 + hy, gx = t.x0 - 1 + hxx;
            const bool ok = t.ok && hp < FB_NH && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            hx[u] = ok ? *reinterpret_cast<const uint4*>(a.x + (((size_t)t.n * a.H + gy) * a.W + gx) * FB_MID + sl * 8)
                       : make_uint4(0, 0, 0, 0);
        }
    };
    auto put = [&](unsigned char* dst) {
#pragma unroll
        for (int u = 0; u < FF_NP; ++u) {
            const int q = u * 512 + tid, hp = q >> 3, sl = q & 7;
            if (hp < FB_NH) *reinterpret_cast<uint4*>(dst + hp * FB_PSTR + sl * 16) = hx[u];
        }
    };
    fetch(tile_at(slot));
    put(t0);
    __syncthreads();
    for (int it = slot; it < tper; it += per_xcd) {
        const TileAt cur = tile_at(it);
        if (!cur.ok) break;
        const int n = cur.n, y0 = cur.y0, x0 = cur.x0;
        // (t0 holds this tile's halo: written during the previous tile's stage 3 - or above for the first tile)
        fetch(tile_at(it + per_xcd));                        // the next tile's halo: in flight during stages 1 and 2
        // ================================================================ stage 1: t1 = relu(conv1(t0) + b1)
        {
            float b1v[4][4];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int r = 0; r < 4; ++r) b1v[rt][r] = a.b1[16 * rt + 4 * fq + r];
#pragma unroll 1
            for (int j = wave; j < FB_NHT; j += 8) {
                const int hp = 16 * j + fr, hy = hp / FB_HS, hxx = hp - hy * FB_HS;
                const bool ok = hp < FB_NH && (unsigned)(y0 - 1 + hy) < (unsigned)a.H && (unsigned)(x0 - 1 + hxx) < (unsigned)a.W;
                f32x4q acc[4];
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) acc[rt] = f32x4q{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8 bv = *reinterpret_cast<const bf16x8*>(t0 + hp * FB_PSTR + (32 * ks + 8 * fq) * 2);
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt) {
                        const bf16x8 wa = *reinterpret_cast<const bf16x8*>(w1s + (16 * rt + fr) * FB_PSTR + (32 * ks + 8 * fq) * 2);
                        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, bv, acc[rt], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    const int c0 = 16 * rt + 4 * fq;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        v[r] = fmaxf(acc[rt][r] + b1v[rt][r], 0.f);
                        if (!ok) v[r] = 0.f;                             // outside the image: conv2's zero padding
                    }
                    uint2 pk;
                    pk.x = pack2(v[0], v[1]);
                    pk.y = pack2(v[2], v[3]);
                    *reinterpret_cast<uint2*>(t1 + hp * FB_PSTR + c0 * 2) = pk;
                }
            }
        }
        __syncthreads();
        // ================================================================ stage 2: t2 = relu(conv2(t1) + b2)
#pragma unroll 1
        for (int cp = 0; cp < 4; ++cp) {
            const int py = 8 * half + 2 * cp;
            f32x4q acc[2] = {f32x4q{0.f, 0.f, 0.f, 0.f}, f32x4q{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const bf16x8 bv = *reinterpret_cast<const bf16x8*>(
                            t1 + ((py + q + dy) * FB_HS + fr + dx) * FB_PSTR + (32 * ks + 8 * fq) * 2);
                        acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[tap][ks], bv, acc[q], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                uint2 pk;
                pk.x = pack2(fmaxf(acc[q][0] + b2v[0], 0.f), fmaxf(acc[q][1] + b2v[1], 0.f));
                pk.y = pack2(fmaxf(acc[q][2] + b2v[2], 0.f), fmaxf(acc[q][3] + b2v[3], 0.f));
                *reinterpret_cast<uint2*>(t2 + ((py + q) * FB_TS + fr) * FB_PSTR + (16 * sub + 4 * fq) * 2) = pk;
            }
        }
        __syncthreads();
        // ================================================================ stage 3: y = relu(bf16(conv3(t2) + b3) + bf16(convd(x) + bd))
        put(t1);                     // conv1's output is dead: its buffer receives the NEXT tile's halo (frees the registers)
#pragma unroll 1
        for (int row = 0; row < 8; ++row) {                  // one row of the tile per trip (registers)
            const int py0 = 8 * half + row;
            f32x4q acc[4], accd[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[j] = f32x4q{0.f, 0.f, 0.f, 0.f}; accd[j] = f32x4q{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 bv = *reinterpret_cast<const bf16x8*>(t2 + (py0 * FB_TS + fr) * FB_PSTR + (32 * ks + 8 * fq) * 2);
                const bf16x8 xv = *reinterpret_cast<const bf16x8*>(
                    t0 + ((py0 + 1) * FB_HS + fr + 1) * FB_PSTR + (32 * ks + 8 * fq) * 2);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3f[j][ks], bv, acc[j], 0, 0, 0);
                    accd[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wdf[j][ks], xv, accd[j], 0, 0, 0);
                }
            }
            unsigned short* py_ = a.y + ((size_t)n * a.H * a.W + (size_t)(y0 + py0) * a.W + x0 + fr) * FB_C + 64 * sub + 8 * fq;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float4 ba = *reinterpret_cast<const float4*>(b3s + 64 * sub + 32 * h + 8 * fq);
                const float4 bb = *reinterpret_cast<const float4*>(b3s + 64 * sub + 32 * h + 8 * fq + 4);
                const float4 da = *reinterpret_cast<const float4*>(bds + 64 * sub + 32 * h + 8 * fq);
                const float4 db = *reinterpret_cast<const float4*>(bds + 64 * sub + 32 * h + 8 * fq + 4);
                const float b3e[8] = {ba.x, ba.y, ba.z, ba.w, bb.x, bb.y, bb.z, bb.w};
                const float bde[8] = {da.x, da.y, da.z, da.w, db.x, db.y, db.z, db.w};
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    // both maps exist as bf16 tensors on the four-launch path: rounded before the add here too
                    const float t = bf16_to_f32(f32_to_bf16(acc[2 * h + (e >> 2)][e & 3] + b3e[e]));
                    const float d = bf16_to_f32(f32_to_bf16(accd[2 * h + (e >> 2)][e & 3] + bde[e]));
                    o[e] = fmaxf(t + d, 0.f);
                }
                uint4 pk;
                pk.x = pack2(o[0], o[1]); pk.y = pack2(o[2], o[3]); pk.z = pack2(o[4], o[5]); pk.w = pack2(o[6], o[7]);
                if (y0 + py0 < a.H && x0 + fr < a.W) *reinterpret_cast<uint4*>(py_ + 32 * h) = pk;
            }
        }
        __syncthreads();             // stage 3 has read t0 / t2, the next halo is complete in t1: swap the two buffers
        unsigned char* tmp = t0; t0 = t1; t1 = tmp;
    }
}

}  // namespace

extern "C" {

// x, y [N][H][W][256] bf16 (NHWC); w1 [64][256], w2 [64][3][3][64], w3 [256][64] bf16 with the BN scales folded in,
// b1 / b2 [64], b3 [256] fp32 (the folded BN shifts).  Any H, W >= 1 (edge tiles are masked).  y must not alias x.
int oadg_bottleneck_frozen_256(const void* x, const void* w1, const float* b1, const void* w2, const float* b2,
                               const void* w3, const float* b3, void* y, int N, int H, int W, void* stream) {
    if (!x || !w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !y || x == y || N < 1 || H < 1 || W < 1) return OADG_EARG;
    if ((long)N * H * W * FB_C >= (1L << 40)) return OADG_EARG;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)bottleneck_frozen_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           FB_LDS);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    FrozenBlockArgs a;
    a.x = (const unsigned short*)x; a.w1 = (const unsigned short*)w1; a.w2 = (const unsigned short*)w2;
    a.w3 = (const unsigned short*)w3; a.b1 = b1; a.b2 = b2; a.b3 = b3; a.y = (unsigned short*)y;
    a.N = N; a.H = H; a.W = W; a.tiles_x = (W + FB_TS - 1) / FB_TS; a.tiles_y = (H + FB_TS - 1) / FB_TS;
    const long total = (long)N * a.tiles_x * a.tiles_y;
    if (total > 0x7fffffffL) return OADG_EARG;
    const int grid = total >= 256 ? 256 : (int)((total + 7) / 8) * 8;         // one workgroup per CU, a multiple of 8 XCDs
    hipLaunchKernelGGL(bottleneck_frozen_kernel, dim3(grid), dim3(512), FB_LDS, (hipStream_t)stream, a);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

// The stage's first block: x [N][H][W][64] (the max-pool output), y [N][H][W][256]; w1 [64][64], w2 [64][3][3][64],
// w3 [256][64], wd [256][64] (the downsample convolution) bf16 with the BN scales folded in; b1, b2 [64], b3, bd [256].
int oadg_bottleneck_frozen_first_64(const void* x, const void* w1, const float* b1, const void* w2, const float* b2,
                                    const void* w3, const float* b3, const void* wd, const float* bd, void* y, int N, int H,
                                    int W, void* stream) {
    if (!x || !w1 || !b1 || !w2 || !b2 || !w3 || !b3 || !wd || !bd || !y || N < 1 || H < 1 || W < 1) return OADG_EARG;
    if ((long)N * H * W * FB_C >= (1L << 40)) return OADG_EARG;
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute((const void*)bottleneck_frozen_first_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS);
        if (e != hipSuccess) return (int)e;
        attr = true;
    }
    FrozenFirstArgs a;
    a.x = (const unsigned short*)x; a.w1 = (const unsigned short*)w1; a.w2 = (const unsigned short*)w2;
    a.w3 = (const unsigned short*)w3; a.wd = (const unsigned short*)wd; a.b1 = b1; a.b2 = b2; a.b3 = b3; a.bd = bd;
    a.y = (unsigned short*)y; a.N = N; a.H = H; a.W = W; a.tiles_x = (W + FB_TS - 1) / FB_TS; a.tiles_y = (H + FB_TS - 1) / FB_TS;
    const long total = (long)N * a.tiles_x * a.tiles_y;
    if (total > 0x7fffffffL) return OADG_EARG;
    const int grid = total >= 256 ? 256 : (int)((total + 7) / 8) * 8;
    hipLaunchKernelGGL(bottleneck_frozen_first_kernel, dim3(grid), dim3(512), FF_LDS, (hipStream_t)stream, a);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

}  // extern "C"
