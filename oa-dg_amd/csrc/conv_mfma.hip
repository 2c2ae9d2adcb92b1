// NHWC bf16 implicit-GEMM convolution on the MFMA cores (gfx950), forward (+ stride-1 data gradient through
// the same kernel with transformed weights), with a fused bias / residual / ReLU epilogue.
//
// Replaces the dense 3x3 / 1x1 convolutions the reference runs through cuDNN:
//   mmdet/models/necks/fpn.py:112-129         FPN lateral 1x1 and output 3x3 convs
//   mmdet/models/dense_heads/rpn_head.py:54-68 RPN 3x3 conv
//   mmdet/models/backbones/resnet.py:166-205   Bottleneck 1x1 / 3x3 convs (+ folded eval-mode BN, :648-657)
//
// GEMM view: M = N*Ho*Wo output pixels, N = K output channels, Kg = R*S*C.  A[m][kg] is gathered from the
// NHWC input (a 64-channel slice of one filter tap = 128 contiguous bytes per pixel, zeros in the padding),
// B^T[k][kg] is the KRSC weight (contiguous).  Tile 128 x 128 x 64 per 256-thread workgroup (4 waves as 2 x 2,
// 64 x 64 per wave = 2 x 2 v_mfma_f32_32x32x16_bf16 accumulators), two LDS stages of 32 KB filled by
// global_load_lds (16 B per lane, asynchronous, no VGPR round trip).  The LDS image is lane-linear, so the
// bank-conflict swizzle lives in the per-lane SOURCE address: the 16-byte piece (row, seg) is stored at slot
// seg ^ ((row >> 1) & 7) of its 128-byte row, which makes every ds_read_b128 lane group hit 16 distinct
// 16-byte bank slots.  Algorithmic work: 2*M*K*R*S*C FLOP; HBM bytes: M*C*2 (input, taps re-read from L2) +
// M*K*2 (output) + K*R*S*C*2 (weights).
#include <algorithm>
#include <vector>
#include "common.h"
#include <type_traits>
#include <stdlib.h>
#include "oadg_hip.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;

struct ConvArgs {
    const unsigned short* x;      // [N,H,W,C] bf16
    const unsigned short* w;      // [K,R,S,C] bf16
    const float* bias;            // [K] or null
    const unsigned short* res;    // [N,Ho,Wo,K] bf16 residual or null
    unsigned short* y;            // [N,Ho,Wo,K] bf16
    const unsigned short* zeros;  // >= 16 bytes of zeros (source of padding / tail rows)
    const unsigned short* mask;   // [N,Ho,Wo,K] bf16 or null: y *= (mask > 0)  (ReLU backward of the tensor y feeds)
    const unsigned char* bits_in; // the same mask as one BIT per element ([rows][K / 8] bytes, bit e of a byte = channel
                                  // 8 j + e), written by the producer's forward launch: 1/16 of the bf16 mask's bytes
    unsigned char* bits_out;      // or null: this launch also stores (y > 0) of what it writes, in that format
    float* colsum;                // [pixel tiles][K] fp32 partial column sums of the stored y, or null
    int N, H, W, C, K, R, S, Ho, Wo, stride, pad, dil, relu;
    long M;
    // output scatter (stride-2 data gradients, one launch per parity class): pixel (n, ho, wo) of this launch is row
    // ((n * OH + ho * osh + oph) * OW + wo * osw + opw) of y / res / mask.  scatter == 0: row = m.
    int scatter, OH, OW, osh, osw, oph, opw;
    // res_up (streaming pointwise kernel only): `res` is a HALF-resolution map [N, H/2, W/2, K] read through the nearest
    // 2x upsampling (pixel (n, y, x) adds row ((n * H/2 + y/2) * W/2 + x/2)): the FPN top-down add (necks/fpn.py:166-175)
    // in the lateral convolution's epilogue.  res_up 1: H = 2^lh, W = 2^lw (shifts); 2: any even H, W (divisions).
    int res_up, lh, lw;
};

__device__ __forceinline__ long out_row(const ConvArgs& a, long m) {
    if (!a.scatter) return m;
    if (a.M <= 0x7fffffffL) {                      // 32-bit divisions: a fraction of the 64-bit sequences
        const unsigned t = (unsigned)m / (unsigned)a.Wo, wo = (unsigned)m - t * (unsigned)a.Wo;
        const unsigned n = t / (unsigned)a.Ho, ho = t - n * (unsigned)a.Ho;
        return ((long)n * a.OH + (long)ho * a.osh + a.oph) * a.OW + (long)wo * a.osw + a.opw;
    }
    const long t = m / a.Wo;
    const int wo = (int)(m - t * a.Wo);
    const long n = t / a.Ho;
    const int ho = (int)(t - n * a.Ho);
    return (n * a.OH + (long)ho * a.osh + a.oph) * a.OW + (long)wo * a.osw + a.opw;
}

// Last step of both epilogues for one 16-byte piece (8 channels of a pixel): residual add, ReLU, ReLU-backward
// mask, bf16 rounding, and the running column sums of what is stored.
// The operands are run-time (uniform) flags.  `OADG_UNIFORM_BRANCH` keeps each block behind a real scalar branch: left
// alone the compiler if-converts them - it computed the ReLU select, the column sums and the (y > 0) bits of EVERY
// piece and threw the results away with v_cndmask (round-2 ISA count: 191 selects + 64 compares + 96 adds per two
// sub-tiles of the streaming kernel with only a residual operand).
#define OADG_UNIFORM_BRANCH() asm volatile("" ::: "memory")
template <bool POST>
__device__ __forceinline__ bf16x8 finish_piece(const ConvArgs& a, bf16x8 v, const bf16x8 rv, const bf16x8 mv,
                                               float* csum, unsigned mbits = 0xffu, size_t off = 0) {
    if (POST) {
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = bf16_to_f32((unsigned short)v[e]);
        if (a.res) {
            OADG_UNIFORM_BRANCH();
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += bf16_to_f32((unsigned short)rv[e]);
            if (a.relu) {
                OADG_UNIFORM_BRANCH();
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], 0.f);
            }
        }
        if (a.mask) {
            OADG_UNIFORM_BRANCH();
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (!(bf16_to_f32((unsigned short)mv[e]) > 0.f)) f[e] = 0.f;
        }
        if (a.bits_in) {
            OADG_UNIFORM_BRANCH();
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (!((mbits >> e) & 1u)) f[e] = 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (short)f32_to_bf16(f[e]);
    }
    if (a.colsum) {
        OADG_UNIFORM_BRANCH();
#pragma unroll
        for (int e = 0; e < 8; ++e) csum[e] += bf16_to_f32((unsigned short)v[e]);
    }
    if (a.bits_out) {
        OADG_UNIFORM_BRANCH();
        unsigned b = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) b |= (bf16_to_f32((unsigned short)v[e]) > 0.f ? 1u : 0u) << e;
        a.bits_out[off >> 3] = (unsigned char)b;
    }
    return v;
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)l, 16, 0, 0);
}

// TBN = 128: 4 waves as 2 x 2, 64 x 64 outputs per wave.  TBN = 64 (layers with K % 128 != 0, e.g. the 64-channel
// convolutions of ResNet's first stage): 4 waves as 4 x 1, 32 x 64 outputs per wave.
// POST: the launch has a residual and/or a ReLU-backward mask operand (compiled out otherwise).
// NST = 2: two LDS stages (next chunk loads while this one computes), 2 workgroups per CU.  NST = 1: one stage
// (32 KiB + 8 KiB), 4 workgroups per CU - for short reductions (1x1 convolutions with C <= 256) the latencies of a
// workgroup (first HBM fetch, epilogue) are covered by the other three instead of by its own pipeline.
// PW: pointwise launch (R = S = 1, stride 1, pad 0, no scatter map): pixel m of y is row m of x, so the 64-bit
// pixel -> (n, ho, wo) divisions and the per-tap bounds tests - 12-15 % of an HBM-bound 1x1 launch - are compiled out.
template <int TBN, bool POST, int NST, bool PW = false>
__device__ __forceinline__ void conv_igemm_body(const ConvArgs& a, const long bid, unsigned char* smem) {
    constexpr int WN = TBN / 64, WM = 4 / WN, AF = BM / WM / 32, NBP = TBN * 8 / 256;
    constexpr int TSTAGE = (BM + TBN) * BK * 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // block -> (m tile, n tile): the K/BN column blocks of one pixel tile are adjacent (shared A through L2)
    // Workgroup b is dispatched to XCD b % 8 (observed; used for speed only).  Each XCD gets a CONTIGUOUS range
    // of pixel tiles (and all K/BN column tiles of a pixel tile back to back), so the halo rows that neighbouring
    // tiles of a 3x3 convolution share, and the A rows the column tiles share, are served by that XCD's L2
    // instead of being fetched through the fabric once per XCD.
    const int n_tiles = a.K / TBN;
    const long m_tiles = (a.M + BM - 1) / BM;
    long mt;
    int nt;
    {
        const long xcd = bid & 7, j = bid >> 3;              // j-th workgroup of this XCD
        const long per = (m_tiles + 7) >> 3;                 // pixel tiles per XCD
        nt = (int)(j % n_tiles);
        // pointwise launches have no halo to share: pixel tiles are dealt round-robin to the XCDs, so at any moment
        // the whole chip streams through ONE compact window of x / y / residual (DRAM page locality: +15-20 % on the
        // HBM-bound 1x1 layers, tools/probe/conv1x1_lab.hip) instead of eight windows an eighth of the tensor apart
        mt = PW ? (j / n_tiles) * 8 + xcd : xcd * per + j / n_tiles;
        if (j / n_tiles >= per || mt >= m_tiles) return;     // padding of the last XCD range
    }
    const long m0 = mt * BM;
    const int k0 = nt * TBN;

    // ---- loader geometry: piece q = i*256 + tid -> row = q >> 3 (0..127), slot = q & 7
    const int cpc = a.C / BK;                 // 64-channel chunks per filter tap
    const int nchunks = a.R * a.S * cpc;
    const unsigned short* a_base[4];
    int a_hi0[4], a_wi0[4];
    int seg[4];
    const unsigned short* b_base[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = i * 256 + tid;
        const int row = q >> 3, slot = q & 7;
        seg[i] = slot ^ ((row >> 1) & 7);
        const long m = m0 + row;
        if (PW) {
            a_base[i] = m < a.M ? a.x + (size_t)m * a.C : nullptr;
            a_hi0[i] = a_wi0[i] = 0;
        } else if (m < a.M) {
            int wo, ho, n;
            if (a.M <= 0x7fffffffL) {              // (uniform) 32-bit divisions where the pixel count allows
                const unsigned t = (unsigned)m / (unsigned)a.Wo;
                wo = (int)((unsigned)m - t * (unsigned)a.Wo);
                n = (int)(t / (unsigned)a.Ho);
                ho = (int)(t - (unsigned)n * (unsigned)a.Ho);
            } else {
                wo = (int)(m % a.Wo);
                const long t = m / a.Wo;
                ho = (int)(t % a.Ho);
                n = (int)(t / a.Ho);
            }
            a_hi0[i] = ho * a.stride - a.pad;
            a_wi0[i] = wo * a.stride - a.pad;
            // the piece's pixel at tap (0, 0) with its channel slot (integer arithmetic: padding may put it before x -
            // such taps fail the bounds test and read the zero line)
            a_base[i] = a.x + ((long)n * a.H * a.W + (long)a_hi0[i] * a.W + a_wi0[i]) * a.C + seg[i] * 8;
        } else {
            a_base[i] = nullptr;
            a_hi0[i] = -(1 << 28);                  // fails every bounds test
            a_wi0[i] = 0;
        }
        b_base[i] = a.w + (size_t)(k0 + (row < TBN ? row : 0)) * a.R * a.S * a.C + seg[i] * 8;
    }

    // chunk-major K order (taps inner): the R*S taps of a 128-byte input line follow each other, see the 256-tile kernel.
    // The stages are issued in chunk order, so tap and channel chunk are counters and a stage adds one uniform offset per
    // operand to the per-piece bases (round 2: divisions by R*S / S and 64-bit multiplies per piece made the loop of the
    // 3x3 launches as long in VALU + SALU issue as in MFMA cycles).
    int st_r = 0, st_s = 0, st_rs = 0, st_c0 = 0;
    auto stage = [&](int kc, int buf) {
        unsigned char* sa = smem + buf * TSTAGE;
        unsigned char* sb = sa + BM * BK * 2;
        if (PW) {
            const int c0 = kc * BK;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                glds16(a_base[i] ? a_base[i] + c0 + seg[i] * 8 : a.zeros, sa + i * 4096 + wave * 1024);
#pragma unroll
            for (int i = 0; i < NBP; ++i) glds16(b_base[i] + c0, sb + i * 4096 + wave * 1024);
            return;
        }
        const int dr = st_r * a.dil, ds = st_s * a.dil;
        const long aoff = ((long)dr * a.W + ds) * a.C + st_c0;           // uniform
        const long boff = (long)st_rs * a.C + st_c0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = (unsigned)(a_hi0[i] + dr) < (unsigned)a.H && (unsigned)(a_wi0[i] + ds) < (unsigned)a.W;
            glds16(ok ? a_base[i] + aoff : a.zeros, sa + i * 4096 + wave * 1024);
        }
#pragma unroll
        for (int i = 0; i < NBP; ++i) glds16(b_base[i] + boff, sb + i * 4096 + wave * 1024);
        ++st_rs;
        if (++st_s == a.S) {
            st_s = 0;
            if (++st_r == a.R) { st_r = 0; st_rs = 0; st_c0 += BK; }
        }
    };

    f32x16 acc[AF][2];
#pragma unroll
    for (int i = 0; i < AF; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int l31 = lane & 31, lh = lane >> 5;
    for (int kc = 0; kc < nchunks; ++kc) {
        const int cur = NST == 2 ? (kc & 1) : 0;
        if (NST == 2 && kc + 1 < nchunks) stage(kc + 1, cur ^ 1);
        const unsigned char* sa = smem + cur * TSTAGE;
        const unsigned char* sb = sa + BM * BK * 2;
        // Two-stage form (two workgroups per CU, 256 registers per wave to spend): the fragments of k-step kk + 1 are requested
        // BEFORE the MFMAs of k-step kk (round 6: the loop used to be { 4 ds_read_b128, s_waitcnt lgkmcnt(0), 4 MFMAs } per
        // k-step - every MFMA group behind a full LDS round trip).  The one-stage form runs four workgroups per CU on 128
        // registers: its neighbours cover the round trip, and the second fragment set would spill; pipelined at THREE
        // workgroups per CU (146 registers) it lost on the shape it exists for (layer2's 3x3: 106.7 us against 100.2).
        constexpr bool PIPE = NST == 2;
        bf16x8 fa[PIPE ? 2 : 1][AF], fb[PIPE ? 2 : 1][2];
        auto frags = [&](const int kk, const int b) {
            const int sg = kk * 2 + lh;
#pragma unroll
            for (int i = 0; i < AF; ++i) {
                const int row = wm * (AF * 32) + i * 32 + l31;
                fa[b][i] = *reinterpret_cast<const bf16x8*>(sa + row * 128 + ((sg ^ ((row >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = wn * 64 + j * 32 + l31;
                fb[b][j] = *reinterpret_cast<const bf16x8*>(sb + row * 128 + ((sg ^ ((row >> 1) & 7)) << 4));
            }
        };
        if (PIPE) frags(0, 0);
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            if (PIPE) {
                if (kk + 1 < BK / 16) frags(kk + 1, (kk + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);  // (left alone, the scheduler sinks the reads below the MFMAs again)
            } else {
                frags(kk, 0);
            }
#pragma unroll
            for (int i = 0; i < AF; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PIPE ? (kk & 1) : 0][i], fb[PIPE ? (kk & 1) : 0][j],
                                                                        acc[i][j], 0, 0, 0);
        }
        if (NST == 1) {
            __syncthreads();                           // every wave is done reading the single stage
            if (kc + 1 < nchunks) stage(kc + 1, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue.  D[row = pixel][col = channel]: a lane holds column l31 and rows (r&3) + 8*(r>>2) + 4*lh of
    // each 32x32 sub-tile.  bias is added in fp32, the tile is staged as bf16 [128 pixels][128 channels] in the
    // (now idle) LDS stages, then written with 16-byte stores: 16 lanes cover the 256 contiguous bytes of one
    // pixel's channels.  Residual add and ReLU happen on the way out (fp32).
    unsigned short* tile = reinterpret_cast<unsigned short*>(smem);      // [128][TBN] bf16 <= 32 KiB
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = wn * 64 + j * 32 + l31;
        const float bv = a.bias ? a.bias[k0 + col] : 0.f;
#pragma unroll
        for (int i = 0; i < AF; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * (AF * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float v = acc[i][j][r] + bv;
                if (a.relu && !a.res) v = fmaxf(v, 0.f);
                tile[row * TBN + col] = f32_to_bf16(v);
            }
        }
    }
    // The residual / mask pieces this thread will combine are requested BEFORE the barrier, all at once: the
    // accumulators are dead by now, so the registers are free and every load of the epilogue is in flight together.
    constexpr int SPR = TBN / 8;                       // 16-byte slots per tile row
    constexpr int NPIECE = (BM * TBN / 8) / 256;
    bf16x8 rv[POST ? NPIECE : 1], mv[POST ? NPIECE : 1];
    unsigned mb[POST ? NPIECE : 1];
    if (POST) {
#pragma unroll
        for (int it = 0; it < NPIECE; ++it) {
            const int q = it * 256 + tid;
            const long m = m0 + q / SPR;
            const size_t off = (size_t)(PW ? (m < a.M ? m : 0) : out_row(a, m < a.M ? m : 0)) * a.K + k0 + (q % SPR) * 8;
            rv[it] = (a.res && m < a.M) ? *reinterpret_cast<const bf16x8*>(a.res + off) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            mv[it] = (a.mask && m < a.M) ? *reinterpret_cast<const bf16x8*>(a.mask + off) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            mb[it] = (a.bits_in && m < a.M) ? a.bits_in[off >> 3] : 0xffu;
        }
    } else {
        rv[0] = mv[0] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        mb[0] = 0xffu;
    }
    __syncthreads();
    float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < NPIECE; ++it) {
        const int q = it * 256 + tid;
        const int row = q / SPR, sg = q % SPR;         // sg = tid % SPR for every piece of this thread
        const long m = m0 + row;
        if (m >= a.M) continue;
        bf16x8 v = *reinterpret_cast<const bf16x8*>(tile + row * TBN + sg * 8);
        const size_t off = (size_t)(PW ? m : out_row(a, m)) * a.K + k0 + sg * 8;
        *reinterpret_cast<bf16x8*>(a.y + off) = finish_piece<POST>(a, v, rv[POST ? it : 0], mv[POST ? it : 0], csum,
                                                                   mb[POST ? it : 0], off);
    }
    if (a.colsum) {       // 256 / SPR threads share a channel slot: combine through the idle second LDS stage
        float* red = reinterpret_cast<float*>(smem + TSTAGE);           // [256 / SPR][TBN] (8 KiB behind stage 0)
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(tid / SPR) * TBN + (tid % SPR) * 8 + e] = csum[e];
        __syncthreads();
        if (tid < TBN) {
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < 256 / SPR; ++g) t += red[g * TBN + tid];
            a.colsum[(size_t)mt * a.K + k0 + tid] = t;
        }
    }
}


template <int TBN, bool POST, int NST, bool PW = false>
__global__ __launch_bounds__(256, (NST == 1 ? 4 : 2)) void conv_igemm_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    conv_igemm_body<TBN, POST, NST, PW>(a, blockIdx.x, smem);
}

// The parity classes of a stride-2 data gradient in ONE launch (round 6; rounds 1-5: one scatter launch per class = dy read
// four times from HBM, four launch ramps for 2 - 8 chunk reductions).  Each class is the stride-1 convolution over dy that
// conv_igemm_kernel runs with a scatter map; here a workgroup picks its class from its index and runs the same body with
// that class's filters / taps / output phase.  Equal-sized classes (even H, W) are INTERLEAVED at the granularity of one
// workgroup per XCD: the four class tiles over the same dy pixels run back to back on the same XCD, so dy comes from HBM
// once and from that L2 three times.  Same products in the same order per class: bit-identical to the per-class launches.
struct S2Class {
    const unsigned short* w;
    float* colsum;
    int R, S, Ho, Wo, oph, opw, block0;
    long M;
};
struct S2Args {
    ConvArgs base;
    S2Class cls[4];
    int n_cls, interleave;
};
template <int TBN, bool POST>
__global__ __launch_bounds__(256, 4) void conv_igemm_s2_kernel(S2Args g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    long bid = blockIdx.x;
    int ci = 0;
    if (g.interleave) {
        ci = (int)((bid >> 3) & 3);
        bid = ((bid >> 5) << 3) | (bid & 7);
    } else {
#pragma unroll
        for (int k = 1; k < 4; ++k)
            if (k < g.n_cls && bid >= g.cls[k].block0) ci = k;
    }
    S2Class c = g.cls[0];
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (k == ci) c = g.cls[k];
    if (!g.interleave) bid -= c.block0;
    ConvArgs a = g.base;
    a.w = c.w; a.colsum = c.colsum; a.R = c.R; a.S = c.S; a.Ho = c.Ho; a.Wo = c.Wo; a.oph = c.oph; a.opw = c.opw; a.M = c.M;
    conv_igemm_body<TBN, POST, 1, false>(a, bid, smem);
}

// ================================================================================================ pointwise, streaming
// HBM-bound 1x1 / stride 1 convolutions with C <= 256 (ResNet conv3 and its mirror, the data gradient of conv1; the
// P2 lateral): y[M][K] = x[M][C] * w[K][C]^T moves M * (C + K * (1 + residual)) * 2 bytes for 2 * M * C * K FLOP - 64 to
// 200 FLOP per byte.  The tile kernels above spend a workgroup's life on serial phases (stage -> MFMA -> epilogue loads ->
// stores); here every memory stream of the launch stays in flight while the matrix pipe works:
//   * persistent workgroups (256 threads, 2 per CU); wave v keeps the weights of 64 output channels x C in REGISTERS for
//     the whole launch (the A operand of v_mfma_f32_16x16x32_bf16, channels on the rows so that a lane ends up with 4
//     consecutive channels of a pixel); K / 256 workgroup columns share a pixel range through their XCD's L2;
//   * pixels stream through an LDS ring of NS sub-tiles (32 pixels, 16 for C = 256) filled by global_load_lds, NS - 1
//     sub-tiles ahead; L2 -> LDS traffic per output byte is C / 256 (the 128-tile kernel: (C + C) / 128);
//   * the residual / mask-bit pieces of sub-tile i + 1 are requested into a second register set before the MFMAs of
//     sub-tile i; waits are counted (s_waitcnt vmcnt(N): loads, LDS-DMA and stores retire in issue order on gfx9);
//   * sub-tile i of a workgroup is number i * ranges + range - a grid-stride order: the chip works inside ONE compact
//     window of x / y / residual at any moment.  With one contiguous range per workgroup (512 far-apart streams) the same
//     kernel ran 15-20 % slower (DRAM page locality; tools/probe/conv1x1_lab.hip).
// Epilogue per wave: accumulators + bias -> bf16 rows [pixel][64 channels] in a private staging area (padded rows) ->
// 16-byte pieces -> finish_piece (residual, ReLU, mask bits, column sums, bits out) -> 128-byte row segments to HBM.
// Results equal the 128-tile kernel's bit for bit (same products, same fp32 summation order per output).
// Round 4, the READ-HEAVY mirror (C = 512: ResNet conv1 of stage 2 / 3 / 4 inputs, the data gradient of their conv3, the P3
// lateral): the register budget of a wave holds the weights of 32 output channels x 512 (128 VGPRs, as 64 x 256 does), so
// a workgroup covers 128 output channels and K / 128 workgroup columns share a pixel range; the pixel ring is 16 pixels x
// 1 KiB wide.  Same schedule, same epilogue (four lanes per pixel instead of eight), bit-identical to the tile kernel.
template <int C>
struct PwGeo {
    static constexpr int KPW = (C >= 512 ? 32 : 64);          // output channels per wave (weights stationary in registers)
    static constexpr int SP = (C >= 256 ? 16 : 32), NS = (C == 256 ? 8 : 4), SLOTS = C / 8;
    static constexpr int LPP = KPW / 8;                        // lanes (16-byte pieces) per pixel row of a wave's staging area
    static constexpr int PPP = 64 / LPP, NP = SP / PPP;        // pixels per epilogue pass, passes per sub-tile
    static constexpr int SUB_BYTES = SP * C * 2, G = SUB_BYTES / 4096;
    static constexpr int RT = KPW / 16, CT = SP / 16, KS = C / 32, STG_STRIDE = KPW * 2 + 16, STG_BYTES = SP * STG_STRIDE;
    static constexpr int LDS = NS * SUB_BYTES + 4 * STG_BYTES + 4 * KPW * 4;
    static constexpr int GRID = 512;
};

#define OADG_VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

template <int C, bool RES, bool BIN, bool BOUT>
__global__ __launch_bounds__(256, 2) void conv_pw_stream_kernel(ConvArgs a, int ncol) {
    using Geo = PwGeo<C>;
    constexpr int KPW = Geo::KPW, SP = Geo::SP, NS = Geo::NS, NP = Geo::NP, SLOTS = Geo::SLOTS, SUB_BYTES = Geo::SUB_BYTES;
    constexpr int LPP = Geo::LPP, PPP = Geo::PPP;
    constexpr int G = Geo::G, RT = Geo::RT, CT = Geo::CT, KS = Geo::KS, STG_STRIDE = Geo::STG_STRIDE, STG_BYTES = Geo::STG_BYTES;
    // vector-memory operations per sub-tile and wave: G LDS-DMA loads, R operand loads, S stores
    constexpr int R = (RES ? NP : 0) + (BIN ? NP : 0), S = NP + (BOUT ? NP : 0);
    constexpr bool POST = RES || BIN;
    static_assert((NS - 2) * G + (NS - 1) * (R + S) <= 63, "vmcnt is a 6-bit counter");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long bid = blockIdx.x;
    const int xcd = (int)(bid & 7), j = (int)(bid >> 3);            // workgroup b runs on XCD b % 8 (speed only)
    const int col = j % ncol;                                       // the columns of a range sit on one XCD
    const long ranges = (gridDim.x >> 3) / ncol * 8;
    const long range = (long)(j / ncol) * 8 + xcd;
    const long n_sub = a.M / SP;                                    // the host guarantees M % SP == 0
    const int n_it = range < n_sub ? (int)((n_sub - range + ranges - 1) / ranges) : 0;
    const int kcol = col * (4 * KPW) + wave * KPW;                  // this wave's first output channel
    unsigned char* ring = smem;
    unsigned char* stg = smem + NS * SUB_BYTES + wave * STG_BYTES;
    float* sbias = reinterpret_cast<float*>(smem + NS * SUB_BYTES + 4 * STG_BYTES) + wave * KPW;
    float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;

    if (n_it > 0) {
        // ---- stationary operands: weights in registers, this wave's 64 biases in LDS
        bf16x8 wf[RT][KS];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                wf[rt][ks] = *reinterpret_cast<const bf16x8*>(a.w + (size_t)(kcol + rt * 16 + fr) * C + ks * 32 + fq * 8);
        if (lane < KPW) sbias[lane] = a.bias ? a.bias[kcol + lane] : 0.f;

        // ---- loader geometry (constant over sub-tiles).  The LDS image of an LDS-DMA instruction is lane-linear, so the
        // bank-conflict swizzle (16-byte slot ^ f(pixel)) is applied on the global source address
        int goff[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int L = (g * 4 + wave) * 64 + lane;
            const int px = L / SLOTS, sl = L % SLOTS;
            goff[g] = px * C + (C == 64 ? (sl ^ ((px >> 1) & 7)) : (sl ^ (px & 15))) * 8;
        }
        auto stage = [&](long sub, int slot) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const unsigned short* src = sub < n_sub ? a.x + (size_t)sub * (SP * C) + goff[g] : a.zeros;
                glds16(src, ring + slot * SUB_BYTES + (g * 4 + wave) * 1024);
            }
        };
        // slot (ks * 4 + fq) ^ f(px) = (ks * 4) ^ (fq ^ f(px)): one base per 16-pixel tile, the k-step is an XOR constant
        int bbase[CT], bsw[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int px = ct * 16 + fr;
            bbase[ct] = px * (C * 2);
            bsw[ct] = (fq ^ (C == 64 ? ((px >> 1) & 7) : (px & 15))) << 4;
        }
        // epilogue pieces of this lane: pixel lane / LPP + PPP * jj, 16-byte slot lane % LPP of the wave's KPW channels
        const int ppx = lane / LPP, psl = lane % LPP;
        // element offset of a piece = (uniform) sub-tile and piece-row terms + this lane's term: no vector multiplies
        const size_t lane_off = (size_t)ppx * a.K + kcol + psl * 8;
        const size_t sub_stride = (size_t)SP * a.K, row8 = (size_t)PPP * a.K;
        bf16x8 rv[2][NP];
        unsigned mb[2][NP];
        auto request = [&](long sub, int set) {     // residual / mask bits of sub-tile `sub` -> register set `set`
            if (!POST) return;
#pragma unroll
            for (int jj = 0; jj < NP; ++jj) {
                const size_t off = (sub < n_sub ? (size_t)sub * sub_stride + jj * row8 : 0) + lane_off;   // past the end: any valid row
                size_t roff = off;
                if (RES && a.res_up) {          // the residual row of pixel (n, y, x) is row (n, y / 2, x / 2) of the half-size map
                    const long prow = (sub < n_sub ? sub * SP + jj * PPP : 0) + ppx;
                    if (a.res_up == 1) {        // power-of-two map: shifts and masks
                        const long x_ = prow & (a.W - 1), y_ = (prow >> a.lw) & (a.H - 1), n_ = prow >> (a.lw + a.lh);
                        roff = (size_t)((((n_ << (a.lh - 1)) + (y_ >> 1)) << (a.lw - 1)) + (x_ >> 1)) * a.K + kcol + psl * 8;
                    } else {                    // any even H, W (M < 2^31: 32-bit divisions)
                        const unsigned t_ = (unsigned)prow / (unsigned)a.W, x_ = (unsigned)prow - t_ * (unsigned)a.W;
                        const unsigned n_ = t_ / (unsigned)a.H, y_ = t_ - n_ * (unsigned)a.H;
                        roff = ((size_t)(n_ * (unsigned)(a.H >> 1) + (y_ >> 1)) * (unsigned)(a.W >> 1) + (x_ >> 1)) * a.K + kcol + psl * 8;
                    }
                }
                // asm: the compiler must not see these loads, or it would wait for ALL vector memory (the LDS-DMA
                // prefetches included) at their first use
                if (RES) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rv[set][jj]) : "v"(a.res + roff) : "memory");
                if (BIN) asm volatile("global_load_ubyte %0, %1, off" : "=v"(mb[set][jj]) : "v"(a.bits_in + (off >> 3)) : "memory");
            }
        };

#pragma unroll
        for (int p = 0; p < NS - 1; ++p) stage(range + p * ranges, p);
        request(range, 0);

        // issue order per sub-tile i: [DMA of i + NS - 1] [operands of i + 1] MFMAs(i) [stores of i]
        auto body = [&](int i, auto set_c) {
            constexpr int set = decltype(set_c)::value;
            const long sub = range + i * ranges;
            // sub-tile i landed: everything issued after its DMA may stay in flight (the pipeline fill was drained before
            // the loop, so the same count holds from the first iteration on)
            OADG_VMCNT((NS - 2) * G + (NS - 1) * (R + S));
            asm volatile("s_barrier" ::: "memory");     // ... for every wave; and slot (i - 1) % NS has been read by all
            stage(sub + (NS - 1) * ranges, (i + NS - 1) % NS);
            request(sub + ranges, set ^ 1);
            const unsigned char* at = ring + (i % NS) * SUB_BYTES;
            f32x4 acc[RT][CT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                bf16x8 pf[CT];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    pf[ct] = *reinterpret_cast<const bf16x8*>(at + bbase[ct] + (bsw[ct] ^ (ks * 64)));
#pragma unroll
                for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[rt][ks], pf[ct], acc[rt][ct], 0, 0, 0);
            }
            // accumulators (+ bias, fp32) -> bf16 rows [pixel][64 channels] in this wave's staging area
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(sbias + rt * 16 + fq * 4);
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    bf16x4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[rt][ct][r] + bv[r];
                        if (a.relu && !RES) v = fmaxf(v, 0.f);
                        o[r] = (short)f32_to_bf16(v);
                    }
                    *reinterpret_cast<bf16x4*>(stg + (ct * 16 + fr) * STG_STRIDE + (rt * 16 + fq * 4) * 2) = o;
                }
            }
            // the wave re-reads its staging rows as 16-byte pieces: order the compiler's view of the two access shapes
            // (LDS executes a wave's operations in issue order)
            asm volatile("" ::: "memory");
            if (POST) {
                // the operands of THIS sub-tile (requested one sub-tile ago); what was issued since stays in flight
                OADG_VMCNT(S + G + R);
#pragma unroll
                for (int jj = 0; jj < NP; ++jj) {   // ties the registers to the wait: no use may be scheduled above it
                    if (RES) asm volatile("" : "+v"(rv[set][jj]));
                    if (BIN) asm volatile("" : "+v"(mb[set][jj]));
                }
            }
#pragma unroll
            for (int jj = 0; jj < NP; ++jj) {
                const int px = ppx + PPP * jj;
                bf16x8 v = *reinterpret_cast<const bf16x8*>(stg + px * STG_STRIDE + psl * 16);
                const size_t off = (size_t)sub * sub_stride + jj * row8 + lane_off;
                v = finish_piece<POST>(a, v, RES ? rv[set][jj] : bf16x8{0, 0, 0, 0, 0, 0, 0, 0}, bf16x8{0, 0, 0, 0, 0, 0, 0, 0},
                                       csum, BIN ? mb[set][jj] : 0xffu, off);
                asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(a.y + off), "v"(v) : "memory");
            }
        };
        OADG_VMCNT(0);                              // the first NS - 1 sub-tiles and the first operands have landed
        for (int i = 0; i < n_it; i += 2) {         // two bodies per trip: the register set index is a constant
            body(i, std::integral_constant<int, 0>{});
            if (i + 1 < n_it) body(i + 1, std::integral_constant<int, 1>{});
        }
    }
    if (a.colsum) {     // one row of partial column sums per pixel range: lanes l, l + LPP, ... share a channel slot
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t = csum[e];
#pragma unroll
            for (int sh = LPP; sh < 64; sh <<= 1) t += __shfl_xor(t, sh, 64);
            csum[e] = t;
        }
        if (lane < LPP) {
#pragma unroll
            for (int e = 0; e < 8; ++e) a.colsum[(size_t)range * a.K + kcol + lane * 8 + e] = csum[e];
        }
    }
}

// geometry test of the streaming pointwise kernel (the operands are tested by the launcher): pixel ranges, or 0
long pw_stream_ranges(long M, int C, int K, int R, int S, int stride, int pad) {
    if (R != 1 || S != 1 || stride != 1 || pad != 0 || (C != 64 && C != 128 && C != 256 && C != 512)) return 0;
    const int kwg = C >= 512 ? 128 : 256;                       // output channels per workgroup (4 waves x KPW)
    if (K % kwg != 0) return 0;
    const int ncol = K / kwg, sp = C >= 256 ? 16 : 32;
    if (ncol > 64 || (64 % ncol) != 0 || M % sp != 0) return 0;
    const long ranges = 512 / ncol;
    return (M / sp) >= 8 * ranges ? ranges : 0;                 // at least 8 sub-tiles per workgroup
}

template <int C>
int launch_pw_stream(const ConvArgs& a, hipStream_t st) {
    using Geo = PwGeo<C>;
    const int ncol = a.K / (4 * Geo::KPW);
    const bool res = a.res != nullptr, bin = a.bits_in != nullptr, bout = a.bits_out != nullptr;
#define OADG_PWS(RE, BI, BO)                                                                                       \
    do {                                                                                                           \
        static bool attr = false;                                                                                  \
        if (!attr) {                                                                                               \
            hipError_t e = hipFuncSetAttribute((const void*)conv_pw_stream_kernel<C, RE, BI, BO>,                  \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, Geo::LDS);              \
            if (e != hipSuccess) return (int)e;                                                                    \
            attr = true;                                                                                           \
        }                                                                                                          \
        hipLaunchKernelGGL((conv_pw_stream_kernel<C, RE, BI, BO>), dim3(Geo::GRID), dim3(256), Geo::LDS, st, a, ncol); \
    } while (0)
    if (res && bin && !bout) OADG_PWS(true, true, false);
    else if (res && !bin && bout) OADG_PWS(true, false, true);
    else if (res && !bin && !bout) OADG_PWS(true, false, false);
    else if (!res && bin && !bout) OADG_PWS(false, true, false);
    else if (!res && !bin && bout) OADG_PWS(false, false, true);
    else if (!res && !bin && !bout) OADG_PWS(false, false, false);
    else return OADG_EARG;                  // (mask bits in AND bits out: no caller)
#undef OADG_PWS
    return OADG_OK;
}

// ================================================================================================ 256 x 256 tile
// Deep-pipelined variant for the large layers: 256 pixels x 256 channels x 64 per 512-thread workgroup, 8 waves as
// 2 (pixel groups) x 4 (channel groups), 128 x 64 outputs per wave as four 64 x 32 quadrants of
// v_mfma_f32_16x16x32_bf16 tiles (channels on the MFMA rows, so a lane ends up with 4 consecutive channels of a
// pixel).  LDS: two K-tile buffers of four 16 KiB half-tiles each - P0/P1 = pixel rows 0-127 / 128-255,
// W0/W1 = channel rows likewise - 128 KiB, one workgroup per CU, two waves per SIMD.
//
// Schedule (per K-tile t, buffer t & 1): TWO phases of 32 MFMAs, each = { ds_read the register sub-tiles, issue two
// half-tiles of global_load_lds, s_waitcnt lgkmcnt(0), s_barrier, 32 MFMAs (two quadrants x K = 64), s_barrier }.
// Pixel group 1 runs one barrier behind group 0, so on every SIMD one wave is in its MFMA phase while the other one is
// in its load phase.
//   phase   reads (buffer t&1)   MFMA quadrants        stages (4 glds per thread)
//     A     W0 + W1 + P0         (p0,w0) (p0,w1)       W1, P1 of tile t+1   (last read: tile t-1 phases A / B)
//     B     P1                   (p1,w1) (p1,w0)       P0, W0 of tile t+2   (last read: tile t phase A)
// Ablations on the FPN 3x3 shape (tools/probe/conv256_lab.hip): with 16 MFMAs per phase (the first form of this
// kernel) a barrier interval costs ~120 cycles beyond its 16 x 16 MFMA cycles - the matrix pipe idles while the barrier
// releases the partner wave - so MFMAs + barriers alone ran at 1490 TFLOP/s; 32 per phase halve the number of
// intervals: 1650, and the whole kernel 1000 -> 1200 TFLOP/s with bit-identical results.  The reads of a phase are
// retired (lgkmcnt(0)) BEFORE its first barrier, so a half-tile may be restaged one phase after its last read;
// global_load_lds results are waited for once per tile, `s_waitcnt vmcnt(4)` before the first barrier of phase B: that
// leaves phase B's own four loads in flight and retires everything tile t+1 needs, read one phase later (the barrier
// in between publishes the other waves' loads).  Tiles past the end are staged from the zero line so that the counts
// stay uniform.
constexpr int TM = 256, TN = 256;
constexpr int HALF_BYTES = 128 * BK * 2;          // 16 KiB
constexpr int BUF_BYTES = 4 * HALF_BYTES;         // 64 KiB

typedef float f32x4v __attribute__((ext_vector_type(4)));

struct TapState {
    int t, r, s, rs, c0;
};

// NW = 2: 256 x 256 tile.  NW = 1 (round 5): 256 pixels x 128 channels for the layers with K = 128 (ResNet layer2's 3x3
// and its data gradient: 1024 pixel tiles of 18 K-tiles each, which the 128-tile kernel ran at 440 - 650 TFLOP/s) - the same
// buffers and schedule with the W1 half-tile left out: 16 MFMAs per phase, 64 accumulator registers.
template <bool POST, int NW>
__global__ __launch_bounds__(512) void conv_igemm256_kernel(ConvArgs a) {
    constexpr int TNW = NW * 128;                       // channels of the tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int n_tiles = a.K / TNW;
    const long m_tiles = (a.M + TM - 1) / TM;
    const long bid = blockIdx.x;
    long mt;
    int nt;
    {
        const long xcd = bid & 7, j = bid >> 3;
        const long per = (m_tiles + 7) >> 3;
        nt = (int)(j % n_tiles);
        mt = xcd * per + j / n_tiles;
        if (j / n_tiles >= per || mt >= m_tiles) return;
    }
    const long m0 = mt * TM;
    const int k0 = nt * TNW;
    const int nk = a.R * a.S * (a.C / BK);

    // ---- loader geometry: piece q = i*512 + tid of a half-tile -> row = q >> 3 (0..127), 16-byte slot q & 7
    const unsigned short* pb[4];      // [h*2+i]: image base of the pixel + channel slot
    int hi0[4], wi0[4];
    const unsigned short* wb[4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = i * 512 + tid;
            const int row = q >> 3, lslot = (q & 7) ^ ((row >> 1) & 7);
            const long m = m0 + h * 128 + row;
            if (m < a.M) {
                const unsigned mu = (unsigned)m;                 // M < 2^31 (checked on the host): 32-bit divisions
                const unsigned tq = mu / (unsigned)a.Wo;
                const int wo = (int)(mu - tq * (unsigned)a.Wo);
                const int n = (int)(tq / (unsigned)a.Ho);
                const int ho = (int)(tq - (unsigned)n * (unsigned)a.Ho);
                pb[h * 2 + i] = a.x + (size_t)n * a.H * a.W * a.C + lslot * 8;
                hi0[h * 2 + i] = ho * a.stride - a.pad;
                wi0[h * 2 + i] = wo * a.stride - a.pad;
            } else {
                pb[h * 2 + i] = a.x;
                hi0[h * 2 + i] = -(1 << 28);            // fails every bounds test -> zero line
                wi0[h * 2 + i] = 0;
            }
            wb[h * 2 + i] = a.w + (size_t)(k0 + (h & (NW - 1)) * 128 + row) * a.R * a.S * a.C + lslot * 8;
        }

    // K-tile order: CHANNEL CHUNK major, taps inner.  A 64-channel chunk of a pixel is one 128-byte line; the R*S taps
    // that touch it now follow each other within 1/(C/64) of the loop, and an XCD's 32 workgroups sweep 1/(C/64) of
    // their input footprint at a time (a P2 3x3: 1.2 MB of 4.7 MB + 0.3 MB of weights inside the 4 MiB L2) - with the
    // taps outermost every line had to survive the whole loop and FETCH_SIZE showed 3.5x the input bytes.
    auto advance = [&](TapState& st) {
        st.t++;
        st.rs++;
        if (++st.s == a.S) {
            st.s = 0;
            if (++st.r == a.R) { st.r = 0; st.rs = 0; st.c0 += BK; }
        }
    };
    auto stage_pix = [&](int h, const TapState& st, int buf) {
        unsigned char* dst = smem + buf * BUF_BYTES + h * HALF_BYTES + wave * 1024;
        const bool live = st.t < nk;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int hi = hi0[h * 2 + i] + st.r * a.dil, wi = wi0[h * 2 + i] + st.s * a.dil;
            const bool ok = live && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
            // offsets inside one image fit 32 bits (checked on the host)
            const unsigned short* src = ok ? pb[h * 2 + i] + ((hi * a.W + wi) * a.C + st.c0) : a.zeros;
            glds16(src, dst + i * 8192);
        }
    };
    auto stage_wgt = [&](int h, const TapState& st, int buf) {
        unsigned char* dst = smem + buf * BUF_BYTES + (2 + h) * HALF_BYTES + wave * 1024;
        const bool live = st.t < nk;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned short* src = live ? wb[h * 2 + i] + (st.rs * a.C + st.c0) : a.zeros;
            glds16(src, dst + i * 8192);
        }
    };

    // fragment addresses inside a half-tile (the same for every buffer): row = base + (lane & 15),
    // 16-byte slot = ks*4 + (lane >> 4), swizzled like the loader
    const int fr = lane & 15, fq = lane >> 4;
    int poff[4][2], woff[2][2];
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int row = wr * 64 + it * 16 + fr;
            poff[it][ks] = row * 128 + (((ks * 4 + fq) ^ ((row >> 1) & 7)) << 4);
        }
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int row = wc * 32 + jt * 16 + fr;
            woff[jt][ks] = row * 128 + (((ks * 4 + fq) ^ ((row >> 1) & 7)) << 4);
        }

    f32x4v acc[NW][2][2][4];       // [w half][w tile][p half][p tile]
#pragma unroll
    for (int x0 = 0; x0 < NW; ++x0)
#pragma unroll
        for (int x1 = 0; x1 < 2; ++x1)
#pragma unroll
            for (int x2 = 0; x2 < 2; ++x2)
#pragma unroll
                for (int x3 = 0; x3 < 4; ++x3) acc[x0][x1][x2][x3] = f32x4v{0.f, 0.f, 0.f, 0.f};

    bf16x8 pf[4][2], wf[NW][2][2];
    auto read_pix = [&](int h, int buf) {
        const unsigned char* base = smem + buf * BUF_BYTES + h * HALF_BYTES;
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) pf[it][ks] = *reinterpret_cast<const bf16x8*>(base + poff[it][ks]);
    };
    auto read_wgt = [&](int h, int buf) {
        const unsigned char* base = smem + buf * BUF_BYTES + (2 + h) * HALF_BYTES;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) wf[h][jt][ks] = *reinterpret_cast<const bf16x8*>(base + woff[jt][ks]);
    };
#define OADG_MFMA32(W_FIRST, PH_)                                                                            \
    do {                                                                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
        asm volatile("s_barrier" ::: "memory");                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        __builtin_amdgcn_s_setprio(1);                                                                       \
        _Pragma("unroll") for (int hh = 0; hh < NW; ++hh)                                                    \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                 \
                _Pragma("unroll") for (int jt = 0; jt < 2; ++jt)                                             \
                    _Pragma("unroll") for (int it = 0; it < 4; ++it)                                         \
                        acc[(hh ^ W_FIRST) & (NW - 1)][jt][PH_][it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(   \
                            wf[(hh ^ W_FIRST) & (NW - 1)][jt][ks], pf[it][ks],                               \
                            acc[(hh ^ W_FIRST) & (NW - 1)][jt][PH_][it], 0, 0, 0);                           \
        __builtin_amdgcn_s_setprio(0);                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        asm volatile("s_barrier" ::: "memory");                                                              \
    } while (0)

    // ---- prologue: tile 0 complete in buffer 0, P0 + W0 of tile 1 in buffer 1 (what phase B of "tile -1" stages)
    TapState s1{0, 0, 0, 0, 0};
    stage_pix(0, s1, 0);
    stage_wgt(0, s1, 0);
    if (NW == 2) stage_wgt(1, s1, 0);
    stage_pix(1, s1, 0);
    advance(s1);                     // s1 = tile 1
    stage_pix(0, s1, 1);
    stage_wgt(0, s1, 1);
    TapState s2 = s1;
    advance(s2);                     // s2 = tile 2
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    if (wr == 1) asm volatile("s_barrier" ::: "memory");     // stagger: group 1 runs one barrier behind

    for (int t = 0; t < nk; ++t) {
        const int buf = t & 1;
        // phase A
        read_wgt(0, buf);
        if (NW == 2) read_wgt(1, buf);
        read_pix(0, buf);
        if (NW == 2) stage_wgt(1, s1, buf ^ 1);
        stage_pix(1, s1, buf ^ 1);
        OADG_MFMA32(0, 0);
        // phase B
        read_pix(1, buf);
        stage_pix(0, s2, buf);
        stage_wgt(0, s2, buf);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        OADG_MFMA32(1, 1);
        s1 = s2;
        advance(s2);
    }
#undef OADG_MFMA32
    if (wr == 0) asm volatile("s_barrier" ::: "memory");     // balance the stagger
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the zero-line stages of the tail have landed
    asm volatile("s_barrier" ::: "memory");

    // ---- epilogue: bf16 C image [256 pixels][256 channels] in LDS (16-byte slot ^ (pixel & 15): the 16 pixels of a
    // ds_write_b64 lane group land on 16 different bank groups), then 16-byte row-contiguous stores
    {
        const int cq = lane >> 4;
#pragma unroll
        for (int wh = 0; wh < NW; ++wh)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                const int ch = wh * 128 + wc * 32 + jt * 16 + 4 * cq;
                float bv[4] = {0.f, 0.f, 0.f, 0.f};
                if (a.bias) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) bv[e] = a.bias[k0 + ch + e];
                }
#pragma unroll
                for (int ph = 0; ph < 2; ++ph)
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int p = ph * 128 + wr * 64 + it * 16 + fr;
                        unsigned short o[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float v = acc[wh][jt][ph][it][e] + bv[e];
                            if (a.relu && !a.res) v = fmaxf(v, 0.f);
                            o[e] = f32_to_bf16(v);
                        }
                        uint2 pk;
                        pk.x = (unsigned)o[0] | ((unsigned)o[1] << 16);
                        pk.y = (unsigned)o[2] | ((unsigned)o[3] << 16);
                        *reinterpret_cast<uint2*>(smem + p * (TNW * 2) + ((((ch >> 3) ^ (p & 15))) << 4) + ((ch >> 2) & 1) * 8) = pk;
                    }
            }
    }
    // residual / mask pieces requested before the barrier, all 16 (x2) loads of the thread in flight together: the
    // 128 accumulator registers are dead once the C image is written
    constexpr int NPIECE = (TM * TNW / 8) / 512, SL = TNW / 8;      // 16-byte slots per pixel row
    bf16x8 rv[POST ? NPIECE : 1], mv[POST ? NPIECE : 1];
    unsigned mb[POST ? NPIECE : 1];
    if (POST) {
#pragma unroll
        for (int it = 0; it < NPIECE; ++it) {
            const int q = it * 512 + tid;
            const long m = m0 + q / SL;
            const size_t off = (size_t)m * a.K + k0 + (q % SL) * 8;
            rv[it] = (a.res && m < a.M) ? *reinterpret_cast<const bf16x8*>(a.res + off) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            mv[it] = (a.mask && m < a.M) ? *reinterpret_cast<const bf16x8*>(a.mask + off) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            mb[it] = (a.bits_in && m < a.M) ? a.bits_in[off >> 3] : 0xffu;
        }
    } else {
        rv[0] = mv[0] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        mb[0] = 0xffu;
    }
    __syncthreads();
    float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < NPIECE; ++it) {
        const int q = it * 512 + tid;
        const int p = q / SL, sg = q % SL;             // sg = tid % SL for every piece of this thread
        const long m = m0 + p;
        if (m >= a.M) continue;
        bf16x8 v = *reinterpret_cast<const bf16x8*>(smem + p * (TNW * 2) + ((sg ^ (p & 15)) << 4));
        const size_t off = (size_t)m * a.K + k0 + sg * 8;
        *reinterpret_cast<bf16x8*>(a.y + off) = finish_piece<POST>(a, v, rv[POST ? it : 0], mv[POST ? it : 0], csum,
                                                                   mb[POST ? it : 0], off);
    }
    if (a.colsum) {       // 512 / SL threads share a channel slot: combine through the 16 KiB behind the C image
        float* red = reinterpret_cast<float*>(smem + 2 * BUF_BYTES);    // [512 / SL][TNW]
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(tid / SL) * TNW + (tid % SL) * 8 + e] = csum[e];
        __syncthreads();
        if (tid < TNW) {
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < 512 / SL; ++g) t += red[g * TNW + tid];
            a.colsum[(size_t)mt * a.K + k0 + tid] = t;
        }
    }
}

}  // namespace

namespace {
// Automatic kernel choice (tools/bench_conv.py, MI355X):
//  - the 256 x 256 phase-pipelined tile (variant 2) when there is a workgroup for every CU and the reduction is long
//    enough to amortise its one-workgroup-per-CU prologue/epilogue: 920-1080 TFLOP/s on the 3x3 layers;
//  - otherwise the single-stage 128-tile kernel at 4 workgroups per CU (variant 3): 1.15-1.4x the 256-tile kernel on
//    the 1x1 layers with C <= 256, 850 TFLOP/s even on 3x3 (the two-stage variant 1: 710) - it is never slower than
//    the two-stage kernel, which stays selectable for comparison.
//  - pointwise launches with C <= 256 and K a multiple of 256 (ResNet conv3 / the data gradient of conv1, P2 lateral):
//    the streaming kernel (variant 4), 1.1-1.2x the 128-tile kernel there (4.2-5.3 TB/s of HBM traffic).
int auto_variant(long M, int H, int W, int C, int K, int nchunks, int R = 0, int S = 0, int stride = 0, int pad = 0) {
    // (K = 128: the 256 x 128 instantiation exists - explicit variant 2 - but is NOT chosen: measured
    //  on ResNet layer2's 3x3 (8 x 128 x 256 px, C = K = 128) 555 - 650 TFLOP/s against the 128-tile kernel's 650 - 715: its
    //  phases are 16 MFMAs between barriers, the interval this kernel's own ablation found too short)
    constexpr bool narrow256 = false;
    const int tnw = K % TN == 0 ? TN : 128;
    const long big = ((M + TM - 1) / TM) * (K / tnw);
    const bool ok256 = (K % TN == 0 || (K == 128 && narrow256)) && big >= 256 && (long)H * W * C < (1L << 31) && M < (1L << 31);
    if (ok256 && nchunks >= 12) return 2;
    if (pw_stream_ranges(M, C, K, R, S, stride, pad) > 0) return 4;
    if (ok256 && nchunks >= 8) return 2;      // (512-channel 1x1 / stride 2: no streaming kernel; 97 / 108 us cold)
    // small maps with a long reduction (layer4 3x3, the P5 / P6 3x3 convolutions, 2048 -> 256): at most two 128-tile
    // workgroups per CU, so the single-stage form has nobody to cover its loads - the two-stage pipeline is 10 - 20 % faster
    // there (cold operands, round 4: 94 / 113 us at 512 x 512 x 3 x 3 on 32 x 64, 45 / 55 at 256 x 256 x 3 x 3)
    const long blocks128 = ((M + BM - 1) / BM) * ((K + 127) / 128);
    if (K % 128 == 0 && blocks128 <= 512 && nchunks >= 32) return 1;
    return 3;
}

static inline bool mask_bits_unused(const void* bits_in, const void* bits_out) { return bits_in != nullptr || bits_out != nullptr; }

int conv_launch(const void* x, const void* w, const float* bias, const void* residual, void* y, const void* zeros16,
                int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil, int relu, int variant,
                void* stream, const void* mask = nullptr, float* colsum_part = nullptr, const int* sc = nullptr,
                const void* bits_in = nullptr, void* bits_out = nullptr) {
    // sc (optional, 8 ints): out_h, out_w = output extent of this launch; OH, OW, osh, osw, oph, opw = scatter map
    if (!x || !w || !y || !zeros16) return OADG_EARG;
    if (N < 1 || H < 1 || W < 1 || R < 1 || S < 1 || stride < 1 || dil < 1 || pad < 0) return OADG_EARG;
    if (C % BK != 0 || K % 64 != 0) return OADG_EARG;   // other shapes stay on the library path
    if (variant < 0 || variant > 4 || (variant == 2 && K % 128 != 0)) return OADG_EARG;
    ConvArgs a;
    a.x = (const unsigned short*)x; a.w = (const unsigned short*)w; a.bias = bias;
    a.res = (const unsigned short*)residual; a.y = (unsigned short*)y; a.zeros = (const unsigned short*)zeros16;
    a.mask = (const unsigned short*)mask; a.colsum = colsum_part;
    a.bits_in = (const unsigned char*)bits_in; a.bits_out = (unsigned char*)bits_out;
    if ((bits_in || bits_out) && K % 8 != 0) return OADG_EARG;
    a.N = N; a.H = H; a.W = W; a.C = C; a.K = K; a.R = R; a.S = S; a.stride = stride; a.pad = pad; a.dil = dil;
    a.relu = relu & 1;
    a.Ho = (H + 2 * pad - dil * (R - 1) - 1) / stride + 1;
    a.Wo = (W + 2 * pad - dil * (S - 1) - 1) / stride + 1;
    a.scatter = 0; a.OH = a.Ho; a.OW = a.Wo; a.osh = a.osw = 1; a.oph = a.opw = 0;
    a.res_up = 0; a.lh = a.lw = 0;
    if (relu & 2) {     // (relu bit 1: `residual` is a half-resolution map added through the nearest 2x upsampling)
        auto lg = [](int v) { int l = 0; while ((1 << l) < v) ++l; return (1 << l) == v ? l : -1; };
        a.lh = lg(H); a.lw = lg(W);
        if (!residual || (H & 1) || (W & 1) || H < 2 || W < 2 || (long)N * H * W >= (1L << 31) || R != 1 || S != 1 || stride != 1 ||
            pad != 0 || sc || mask || mask_bits_unused(bits_in, bits_out))
            return OADG_EARG;
        a.res_up = (a.lh >= 1 && a.lw >= 1) ? 1 : 2;
        a.relu = relu & 1;
        if (variant == 0) variant = 4;
        if (variant != 4) return OADG_EARG;
    }
    if (sc) {
        if (stride != 1 || sc[0] < 1 || sc[1] < 1 || sc[4] < 1 || sc[5] < 1 || sc[6] < 0 || sc[7] < 0) return OADG_EARG;
        if ((long)(sc[0] - 1) * sc[4] + sc[6] >= sc[2] || (long)(sc[1] - 1) * sc[5] + sc[7] >= sc[3]) return OADG_EARG;
        a.Ho = sc[0]; a.Wo = sc[1];           // rows past the input extent read the zero line
        a.scatter = 1; a.OH = sc[2]; a.OW = sc[3]; a.osh = sc[4]; a.osw = sc[5]; a.oph = sc[6]; a.opw = sc[7];
        if (variant == 0 || variant == 2) variant = 3;      // the 128-tile kernels carry the scatter map
    }
    if (a.Ho < 1 || a.Wo < 1) return OADG_EARG;
    a.M = (long)N * a.Ho * a.Wo;
    const bool post = residual != nullptr || mask != nullptr || bits_in != nullptr;
    if (variant == 0) {
        variant = auto_variant(a.M, H, W, C, K, R * S * (C / BK), R, S, stride, pad);
        // the streaming kernel takes mask BITS only, and not together with bits out
        if (variant == 4 && (sc || mask || (bits_in && bits_out))) variant = 3;
    }
    if (variant == 2 && ((long)H * W * C >= (1L << 31) || a.M >= (1L << 31))) variant = 1;
    if (variant == 4) {
        if (sc || mask || (bits_in && bits_out) || pw_stream_ranges(a.M, C, K, R, S, stride, pad) == 0) return OADG_EARG;
        int rc = OADG_EARG;
        if (C == 64) rc = launch_pw_stream<64>(a, (hipStream_t)stream);
        else if (C == 128) rc = launch_pw_stream<128>(a, (hipStream_t)stream);
        else if (C == 256) rc = launch_pw_stream<256>(a, (hipStream_t)stream);
        else if (C == 512) rc = launch_pw_stream<512>(a, (hipStream_t)stream);
        if (rc != OADG_OK) return rc;
        OADG_LAUNCH_CHECK();
        return OADG_OK;
    }
    if (variant == 2) {
        static bool attr_set = false;
        if (!attr_set) {
            const void* fns[4] = {(const void*)conv_igemm256_kernel<false, 2>, (const void*)conv_igemm256_kernel<true, 2>,
                                  (const void*)conv_igemm256_kernel<false, 1>, (const void*)conv_igemm256_kernel<true, 1>};
            for (const void* f : fns) {
                hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF_BYTES + 16384);
                if (e != hipSuccess) return (int)e;
            }
            attr_set = true;
        }
        const int nw = K % TN == 0 ? 2 : 1;
        const long m_tiles = (a.M + TM - 1) / TM;
        const long blocks = ((m_tiles + 7) / 8) * 8 * (K / (nw * 128));
        if (blocks > 0x7fffffffL) return OADG_EARG;
#define OADG_L256(PO, NW_) \
    hipLaunchKernelGGL((conv_igemm256_kernel<PO, NW_>), dim3((unsigned)blocks), dim3(512), 2 * BUF_BYTES + 16384, (hipStream_t)stream, a)
        if (nw == 2) { if (post) OADG_L256(true, 2); else OADG_L256(false, 2); }
        else { if (post) OADG_L256(true, 1); else OADG_L256(false, 1); }
#undef OADG_L256
    } else {
        const long m_tiles = (a.M + BM - 1) / BM;
        // 128 x 64 tiles also where 128 x 128 tiles would leave compute units without a workgroup (R101-DC5's 46 x 80 maps:
        // 115 pixel tiles x K / 128 = 230 workgroups of one wave per SIMD each; P5 / P6 of the FPN): twice the workgroups, the
        // same products in the same order (bit-identical).  tbn64_max: the largest 128-tile workgroup count that still takes
        // the narrow tile.
        constexpr long tbn64_max = 256;
        const int tbn = (K % BN == 0 && m_tiles * (K / BN) > tbn64_max) ? BN : 64;
        const long blocks = ((m_tiles + 7) / 8) * 8 * (K / tbn);     // 8 equal XCD ranges (the kernel drops the padding)
        if (blocks > 0x7fffffffL) return OADG_EARG;
        const bool one = variant == 3;
        const unsigned lds2 = 2 * (BM + tbn) * BK * 2, lds1 = (BM + tbn) * BK * 2 + 8192;
        hipStream_t st = (hipStream_t)stream;
#define OADG_L128(TB, PO, NS, LDS) \
    hipLaunchKernelGGL((conv_igemm_kernel<TB, PO, NS>), dim3((unsigned)blocks), dim3(256), LDS, st, a)
#define OADG_L128PW(TB, PO, LDS) \
    hipLaunchKernelGGL((conv_igemm_kernel<TB, PO, 1, true>), dim3((unsigned)blocks), dim3(256), LDS, st, a)
        const bool pw = one && R == 1 && S == 1 && stride == 1 && pad == 0 && !a.scatter;
        if (pw) {
            if (tbn == BN) { if (post) OADG_L128PW(128, true, lds1); else OADG_L128PW(128, false, lds1); }
            else { if (post) OADG_L128PW(64, true, lds1); else OADG_L128PW(64, false, lds1); }
        } else if (tbn == BN) {
            if (one) { if (post) OADG_L128(128, true, 1, lds1); else OADG_L128(128, false, 1, lds1); }
            else { if (post) OADG_L128(128, true, 2, lds2); else OADG_L128(128, false, 2, lds2); }
        } else {
            if (one) { if (post) OADG_L128(64, true, 1, lds1); else OADG_L128(64, false, 1, lds1); }
            else { if (post) OADG_L128(64, true, 2, lds2); else OADG_L128(64, false, 2, lds2); }
        }
#undef OADG_L128
#undef OADG_L128PW
    }
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}
}  // namespace

extern "C" int oadg_conv2d_nhwc_bf16(const void* x, const void* w, const float* bias, const void* residual,
                                     void* y, const void* zeros16, int N, int H, int W, int C, int K, int R,
                                     int S, int stride, int pad, int dil, int relu, void* stream) {
    return conv_launch(x, w, bias, residual, y, zeros16, N, H, W, C, K, R, S, stride, pad, dil, relu, 0, stream);
}

// which kernel the automatic choice takes for a problem by its geometry (2, 3 or 4; 0 for unsupported shapes).  A launch
// with a bf16 mask operand, an output scatter map, or mask bits in AND bits out takes 3 where this says 4.
extern "C" int oadg_conv2d_auto_variant(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil) {
    if (N < 1 || H < 1 || W < 1 || R < 1 || S < 1 || stride < 1 || dil < 1 || pad < 0) return 0;
    if (C % BK != 0 || K % 64 != 0) return 0;
    const int Ho = (H + 2 * pad - dil * (R - 1) - 1) / stride + 1, Wo = (W + 2 * pad - dil * (S - 1) - 1) / stride + 1;
    if (Ho < 1 || Wo < 1) return 0;
    return auto_variant((long)N * Ho * Wo, H, W, C, K, R * S * (C / BK), R, S, stride, pad);
}

// Data-gradient form with the backward of the producer's epilogue fused in:  y = (conv(x, w) [+ residual]) * (mask > 0)
// and colsum_part[tile][k] = per-pixel-tile column sums of the stored y (reduce with oadg_colsum_reduce; rows =
// oadg_conv2d_pixel_tiles).  mask / colsum_part may be NULL.
extern "C" int oadg_conv2d_nhwc_bf16_ex(const void* x, const void* w, const float* bias, const void* residual,
                                        void* y, const void* zeros16, int N, int H, int W, int C, int K, int R, int S,
                                        int stride, int pad, int dil, int relu, int variant, const void* mask,
                                        float* colsum_part, const void* mask_bits, void* relu_bits_out, void* stream) {
    return conv_launch(x, w, bias, residual, y, zeros16, N, H, W, C, K, R, S, stride, pad, dil, relu, variant, stream,
                       mask, colsum_part, nullptr, mask_bits, relu_bits_out);
}

// One parity class of a strided data gradient (or any stride-1 convolution whose output pixels are written on a
// strided grid of a larger tensor): the launch covers out_h x out_w output pixels per image (input rows / columns past
// the extent of x read zeros), pixel (n, ho, wo) is stored at row ((n*OH + ho*osh + oph)*OW + wo*osw + opw) of y
// (and residual / mask are read there).  colsum_part has ceil(N*out_h*out_w / 128) rows.
extern "C" int oadg_conv2d_nhwc_bf16_scatter(const void* x, const void* w, const float* bias, const void* residual,
                                             void* y, const void* zeros16, int N, int H, int W, int C, int K, int R,
                                             int S, int pad, int dil, int relu, int out_h, int out_w, int OH, int OW,
                                             int osh, int osw, int oph, int opw, const void* mask,
                                             float* colsum_part, const void* mask_bits, void* stream) {
    const int sc[8] = {out_h, out_w, OH, OW, osh, osw, oph, opw};
    return conv_launch(x, w, bias, residual, y, zeros16, N, H, W, C, K, R, S, 1, pad, dil, relu, 3, stream, mask,
                       colsum_part, sc, mask_bits, nullptr);
}

// dx of a stride-2 convolution (3x3 / pad 1 or 1x1 / pad 0) in ONE launch: all parity classes of
// oadg_conv2d_nhwc_bf16_scatter together (conv_igemm_s2_kernel).  dy [N,Ho,Wo,K] bf16; wt = the class filters
// (oadg_prep_conv_weights wt_mode 2: class blocks [C][taps][K] in the order (0,0) (0,1) (1,0) (1,1)); dx [N,H,W,C];
// residual (1x1 only in practice: dx += in place, may alias dx) / mask / mask_bits are indexed like dx; colsum_part: the
// classes' partial rows back to back, ceil(N*ha*wa / 128) rows per class in class order.  Classes with no pixel are skipped.
extern "C" int oadg_conv2d_dgrad_s2_nhwc_bf16(const void* dy, const void* wt, const void* residual, void* dx,
                                              const void* zeros16, int N, int Ho, int Wo, int K, int C, int R, int H, int W,
                                              const void* mask, float* colsum_part, const void* mask_bits, void* stream) {
    if (!dy || !wt || !dx || !zeros16 || N < 1 || Ho < 1 || Wo < 1 || H < 1 || W < 1 || (R != 3 && R != 1)) return OADG_EARG;
    if (K % BK != 0 || C % 64 != 0 || (mask_bits && C % 8 != 0)) return OADG_EARG;
    static const int cls_tab[4][5] = {{0, 0, 1, 1, 0}, {0, 1, 1, 2, 1}, {1, 0, 2, 1, 3}, {1, 1, 2, 2, 5}};   // ph, pw, taps h, w, block
    S2Args g;
    ConvArgs& a = g.base;
    a.x = (const unsigned short*)dy; a.w = nullptr; a.bias = nullptr; a.res = (const unsigned short*)residual;
    a.y = (unsigned short*)dx; a.zeros = (const unsigned short*)zeros16; a.mask = (const unsigned short*)mask;
    a.bits_in = (const unsigned char*)mask_bits; a.bits_out = nullptr; a.colsum = nullptr;
    a.N = N; a.H = Ho; a.W = Wo; a.C = K; a.K = C; a.R = 1; a.S = 1; a.stride = 1; a.pad = 0; a.dil = 1; a.relu = 0;
    a.Ho = a.Wo = 1; a.M = 0;
    a.scatter = 1; a.OH = H; a.OW = W; a.osh = 2; a.osw = 2; a.oph = a.opw = 0;
    a.res_up = 0; a.lh = a.lw = 0;
    const int tbn = (C % BN == 0) ? BN : 64;
    g.n_cls = 0;
    long blocks = 0, rows = 0;
    bool equal = true;
    long first_blocks = -1;
    for (int k = 0; k < (R == 3 ? 4 : 1); ++k) {
        const int ph = cls_tab[k][0], pw = cls_tab[k][1];
        const int ha = (H - ph + 1) / 2, wa = (W - pw + 1) / 2;
        if (ha < 1 || wa < 1) continue;
        // rows / columns of dy past its extent read the zero line (class taps reach dy[a + 1])
        if ((long)(ha - 1) * 2 + ph >= H || (long)(wa - 1) * 2 + pw >= W) return OADG_EARG;
        S2Class& c = g.cls[g.n_cls++];
        c.w = (const unsigned short*)wt + (size_t)cls_tab[k][4] * C * K;
        c.R = cls_tab[k][2]; c.S = cls_tab[k][3]; c.Ho = ha; c.Wo = wa; c.oph = ph; c.opw = pw;
        c.M = (long)N * ha * wa;
        const long m_tiles = (c.M + BM - 1) / BM;
        c.colsum = colsum_part ? colsum_part + (size_t)rows * C : nullptr;
        rows += m_tiles;
        const long b = ((m_tiles + 7) / 8) * 8 * (C / tbn);
        if (first_blocks < 0) first_blocks = b;
        equal = equal && b == first_blocks;
        if (blocks + b > 0x7fffffffL) return OADG_EARG;
        c.block0 = (int)blocks;
        blocks += b;
    }
    if (g.n_cls == 0) return OADG_OK;
    for (int k = g.n_cls; k < 4; ++k) g.cls[k] = g.cls[0];
    g.interleave = (g.n_cls == 4 && equal) ? 1 : 0;
    const bool post = residual != nullptr || mask != nullptr || mask_bits != nullptr;
    const unsigned lds1 = (BM + tbn) * BK * 2 + 8192;
    hipStream_t st = (hipStream_t)stream;
#define OADG_LS2(TB, PO) hipLaunchKernelGGL((conv_igemm_s2_kernel<TB, PO>), dim3((unsigned)blocks), dim3(256), lds1, st, g)
    if (tbn == BN) { if (post) OADG_LS2(128, true); else OADG_LS2(128, false); }
    else { if (post) OADG_LS2(64, true); else OADG_LS2(64, false); }
#undef OADG_LS2
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

// rows of the colsum_part buffer for a problem / variant (0 = automatic)
extern "C" long oadg_conv2d_pixel_tiles(int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                                        int variant) {
    if (variant == 0) variant = oadg_conv2d_auto_variant(N, H, W, C, K, R, S, stride, pad, dil);
    if (variant == 0) return 0;
    const int Ho = (H + 2 * pad - dil * (R - 1) - 1) / stride + 1, Wo = (W + 2 * pad - dil * (S - 1) - 1) / stride + 1;
    const long M = (long)N * Ho * Wo;
    if (variant == 4) return pw_stream_ranges(M, C, K, R, S, stride, pad);      // one row per pixel range
    return variant == 2 ? (M + TM - 1) / TM : (M + BM - 1) / BM;
}

// same, with the kernel variant chosen by the caller: 0 = automatic, 1 = 128 x 128 tile (two LDS stages), 2 = 256 x 256
// tile, 3 = 128 x 128 tile (one stage, 4 workgroups per CU), 4 = streaming pointwise kernel (EARG unless eligible)
extern "C" int oadg_conv2d_nhwc_bf16_variant(const void* x, const void* w, const float* bias, const void* residual,
                                             void* y, const void* zeros16, int N, int H, int W, int C, int K, int R,
                                             int S, int stride, int pad, int dil, int relu, int variant,
                                             void* stream) {
    return conv_launch(x, w, bias, residual, y, zeros16, N, H, W, C, K, R, S, stride, pad, dil, relu, variant, stream);
}

// ================================================================================================ weight gradient
// dW[k][r][s][c] = sum_p dy[p][k] * x[p @ tap(r,s)][c]   (p runs over the N*Ho*Wo output pixels)
//
// GEMM view: M = K (output channels), N = C (input channels), reduction over pixels.  Both operands are stored
// pixel-major in HBM (NHWC), i.e. TRANSPOSED with respect to what an MFMA fragment wants (8 consecutive reduction
// indices per lane).  The tiles are therefore staged as they are ([64 pixels][128 channels], filled by
// global_load_lds) and read with ds_read_b64_tr_b16, the gfx950 transposing LDS read: inside a 16-lane group
// source lane 4j+q supplies the 8-byte segment {row j, columns 4q..4q+3} of a 4 x 16 block and lane l receives
// column l (4 consecutive pixels of one channel); two reads make one 8-pixel fragment.  Slot swizzle
// slot' = slot ^ ((row & 3) << 2) keeps the 32 lanes of a read group on 64 distinct banks.
// One workgroup owns one (k tile, c tile, tap) and a contiguous range of pixel chunks (split-K); fp32 partial
// tiles go to a workspace and are summed by a second kernel in fixed order (deterministic, no atomics).
namespace {

struct WgradArgs {
    const unsigned short* x;      // [N,H,W,C]
    const unsigned short* dy;     // [N,Ho,Wo,K]
    float* part;                  // [splits][K][R*S][C] fp32
    const unsigned short* zeros;
    int N, H, W, C, K, R, S, Ho, Wo, stride, pad, dil, splits, chunks_per_split;
    int strip_rows;               // > 0 (256-tile kernel, Wo % 64 == 0): a split = strip_rows whole output rows, K-tiles in
                                  // row-strip order (see wgrad256_tile); 0: a split = chunks_per_split consecutive 64-pixel chunks
    long P;
};

typedef short v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s* lds_v4s_ptr;

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* tile, int pix0, int chan0, int lane) {
    // 8 consecutive pixels (pix0 + 8*(lane>>5) ...) of channel chan0 + (lane & 31), from a [64][128] bf16 tile
    const int g = lane >> 4, s = lane & 15;
    const int chan = chan0 + (g & 1) * 16 + (s & 3) * 4;
    const int rowb = pix0 + (g >> 1) * 8 + (s >> 2);
    bf16x8 out;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int row = rowb + u * 4;
        const int slot = (chan >> 3) ^ ((row & 3) << 2);
        const unsigned char* p = tile + row * 256 + slot * 16 + (chan & 7) * 2;
        const v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_ptr)p);
        out[u * 4 + 0] = r[0]; out[u * 4 + 1] = r[1]; out[u * 4 + 2] = r[2]; out[u * 4 + 3] = r[3];
    }
    return out;
}

constexpr int WP = 64;                                  // pixels per chunk
constexpr int WSTAGE = 2 * WP * 128 * 2;                // dy tile + x tile = 32 KiB

// NST as in conv_igemm_kernel: 2 LDS stages at 2 workgroups per CU, or 1 stage at 4 workgroups per CU.
template <int NST>
__global__ __launch_bounds__(256, (NST == 1 ? 4 : 2)) void conv_wgrad_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int kt_n = a.K / 128, ct_n = a.C / 128, RS = a.R * a.S;
    // Work order: all (tap, k tile, c tile) workgroups of ONE pixel range run back to back on ONE XCD (block b
    // is dispatched to XCD b % 8), so the dy / x slices of that range are read from HBM once and then served by
    // that XCD's L2 to the other ~36 workgroups.  (Placement is a speed assumption only.)
    const int combos = kt_n * ct_n * RS;
    long bid = blockIdx.x;
    int split, combo;
    if (a.splits % 8 == 0) {
        const long xcd = bid & 7, j = bid >> 3;
        combo = (int)(j % combos);
        split = (int)((j / combos) * 8 + xcd);
    } else {
        combo = (int)(bid % combos);
        split = (int)(bid / combos);
    }
    const int ct = combo % ct_n;
    const int kt = (combo / ct_n) % kt_n;
    const int rs = combo / (ct_n * kt_n);
    const int r = rs / a.S, s = rs - r * a.S;
    const int k0 = kt * 128, c0 = ct * 128;
    const long nchunks = (a.P + WP - 1) / WP;
    const long ch0 = (long)split * a.chunks_per_split;
    const long ch1 = ch0 + a.chunks_per_split < nchunks ? ch0 + a.chunks_per_split : nchunks;

    // per-thread loader state: the 4 pieces a thread fetches per tile sit on rows (i*256 + tid) >> 4.  Their pixel is
    // decomposed once; per chunk the input offset and the tap coordinates advance by additions only (round 2: the
    // per-chunk 64-bit multiplies - 48 multiply instructions among 200 VALU per 16 MFMAs - made these kernels
    // VALU-bound at 430-550 TFLOP/s), and the dy offset is a uniform scalar.
    // State per piece: tap coordinates (hi, wi) and the input offset.  Row i*16 + tid/16 and the swizzled slot
    // (tid & 15) ^ ((row & 3) << 2) - the same for the four pieces - give one x base and one dy base per thread.
    int lhi[4], lwi[4];
    unsigned lxoff[4];          // element offsets into x, modulo 2^32 (the host checks N*H*W*C < 2^32)
    const int lrow0 = tid >> 4;
    const long dW = (long)a.stride * a.C, dH = (long)a.stride * a.W * a.C, dN = (long)a.H * a.W * a.C;
    uintptr_t xb, dyb0;
    {
        const long pbase = ch0 * WP;
        const long toff = ((long)(r * a.dil - a.pad) * a.W + (s * a.dil - a.pad)) * a.C;
        const int lslot = (tid & 15) ^ ((lrow0 & 3) << 2);
        xb = (uintptr_t)a.x + (uintptr_t)((toff + c0 + lslot * 8) * 2);
        dyb0 = (uintptr_t)(a.dy + (size_t)lrow0 * a.K + k0 + lslot * 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long p = pbase + lrow0 + i * 16;
            long n;
            int ho, wo;
            if (p <= 0x7fffffffL) {                    // 32-bit divisions (a fraction of the 64-bit sequences)
                const unsigned t = (unsigned)p / (unsigned)a.Wo;
                wo = (int)((unsigned)p - t * (unsigned)a.Wo);
                n = t / (unsigned)a.Ho;
                ho = (int)(t - (unsigned)n * (unsigned)a.Ho);
            } else {
                wo = (int)(p % a.Wo);
                const long t = p / a.Wo;
                ho = (int)(t % a.Ho);
                n = t / a.Ho;
            }
            lhi[i] = ho * a.stride - a.pad + r * a.dil;
            lwi[i] = wo * a.stride - a.pad + s * a.dil;
            lxoff[i] = (unsigned)(n * dN + ho * dH + wo * dW);
        }
    }
    const int adv_h = WP / a.Wo, adv_w = WP - adv_h * a.Wo;     // 64 pixels = adv_h rows + adv_w columns
    const unsigned add_chunk = (unsigned)(adv_w * dW + adv_h * dH), add_wrap_w = (unsigned)(dH - a.Wo * dW),
                   add_wrap_h = (unsigned)(dN - a.Ho * dH);
    const int hi_chunk = adv_h * a.stride, wi_chunk = adv_w * a.stride, wi_wrap = a.Wo * a.stride, hi_wrap = a.Ho * a.stride;
    // wo >= Wo <=> wi >= wi_lim, ho >= Ho <=> hi >= hi_lim (the tap shift is the same on both sides)
    const int wi_lim = wi_wrap - a.pad + s * a.dil, hi_lim = hi_wrap - a.pad + r * a.dil;
    const long chunk_dy = (long)WP * a.K, row16_dy = 16L * a.K;

    auto stage = [&](long ch, int buf) {
        unsigned char* sa = smem + buf * WSTAGE;        // dy tile [64][128]
        unsigned char* sb = sa + WP * 256;              // x tile  [64][128]
        const long left = a.P - ch * WP;
        const int rows = left < WP ? (int)left : WP;    // (uniform) valid rows of this chunk
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool in = lrow0 + i * 16 < rows;
            const bool ok = in && (unsigned)lhi[i] < (unsigned)a.H && (unsigned)lwi[i] < (unsigned)a.W;
            const uintptr_t sdy = in ? dyb0 + (uintptr_t)((ch * chunk_dy + i * row16_dy) * 2) : (uintptr_t)a.zeros;
            const uintptr_t sx = ok ? xb + ((uintptr_t)lxoff[i] << 1) : (uintptr_t)a.zeros;
            glds16((const void*)sdy, sa + i * 4096 + wave * 1024);
            glds16((const void*)sx, sb + i * 4096 + wave * 1024);
            // advance this row's pixel by one chunk
            lwi[i] += wi_chunk;
            lhi[i] += hi_chunk;
            lxoff[i] += add_chunk;
            if (lwi[i] >= wi_lim) {
                lwi[i] -= wi_wrap;
                lhi[i] += a.stride;
                lxoff[i] += add_wrap_w;
            }
            while (lhi[i] >= hi_lim) {
                lhi[i] -= hi_wrap;
                lxoff[i] += add_wrap_h;
            }
        }
    };
    // fragment addresses: row & 3 and the channel term do not depend on the k-step or on which 4-pixel half is read, so
    // one byte offset per 32-channel fragment is precomputed and every ds_read_b64_tr_b16 is base + immediate
    int faoff[2], fboff[2];
    {
        const int g = lane >> 4, s16 = lane & 15;
        const int rowb = (g >> 1) * 8 + (s16 >> 2);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ca = wm * 64 + i * 32 + (g & 1) * 16 + (s16 & 3) * 4, cb = wn * 64 + i * 32 + (g & 1) * 16 + (s16 & 3) * 4;
            faoff[i] = rowb * 256 + (((ca >> 3) ^ ((rowb & 3) << 2)) << 4) + (ca & 7) * 2;
            fboff[i] = rowb * 256 + (((cb >> 3) ^ ((rowb & 3) << 2)) << 4) + (cb & 7) * 2;
        }
    }
    auto frag = [&](const unsigned char* tile, int off, int t) -> bf16x8 {
        bf16x8 out;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const v4s rr = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_ptr)(tile + off + (t * 16 + u * 4) * 256));
            out[u * 4 + 0] = rr[0]; out[u * 4 + 1] = rr[1]; out[u * 4 + 2] = rr[2]; out[u * 4 + 3] = rr[3];
        }
        return out;
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if (ch0 < ch1) {
        stage(ch0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (long ch = ch0; ch < ch1; ++ch) {
            const int cur = NST == 2 ? (int)((ch - ch0) & 1) : 0;
            if (NST == 2 && ch + 1 < ch1) stage(ch + 1, cur ^ 1);
            const unsigned char* sa = smem + cur * WSTAGE;
            const unsigned char* sb = sa + WP * 256;
            // (two-stage form: the fragments of step t + 1 are requested before the MFMAs of step t - see conv_igemm_body)
            constexpr bool PIPE = NST == 2;
            bf16x8 fa[PIPE ? 2 : 1][2], fb[PIPE ? 2 : 1][2];
            auto frags = [&](const int t, const int b) {
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[b][i] = frag(sa, faoff[i], t);
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[b][j] = frag(sb, fboff[j], t);
            };
            if (PIPE) frags(0, 0);
#pragma unroll
            for (int t = 0; t < WP / 16; ++t) {
                if (PIPE) {
                    if (t + 1 < WP / 16) frags(t + 1, (t + 1) & 1);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                    frags(t, 0);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PIPE ? (t & 1) : 0][i], fb[PIPE ? (t & 1) : 0][j],
                                                                            acc[i][j], 0, 0, 0);
            }
            if (NST == 1) {
                __syncthreads();
                if (ch + 1 < ch1) stage(ch + 1, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    const int l31 = lane & 31, lh = lane >> 5;
    float* out = a.part + (size_t)split * a.K * RS * a.C;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int k = k0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                const int c = c0 + wn * 64 + j * 32 + l31;
                out[((size_t)k * RS + rs) * a.C + c] = acc[i][j][e];
            }
}

// ---- 256 x 256 tile weight gradient on the phase pipeline of conv_igemm256_kernel -------------------------------
// dW tile [256 c][256 k] of one tap; reduction over 64-pixel K-tiles.  LDS: two buffers of four 16 KiB half-tiles
// D0/D1 = dy channels 0-127 / 128-255, X0/X1 = x channels likewise, each [64 pixels][128 channels] as it lies in
// HBM (filled by global_load_lds); MFMA fragments (8 consecutive pixels per lane) come out of the transposing
// ds_read_b64_tr_b16.  16-byte slot swizzle slot ^ (((row & 3) << 2) | (((row >> 3) & 1) << 1)): the 8 pixel rows a
// half-wave reads per instruction land on 8 different 32-byte bank groups.  Schedule, stagger and vmcnt
// accounting are those of conv_igemm256_kernel with (P, W) -> (D, X).
__device__ __forceinline__ bf16x8 tr_frag16(const unsigned char* half, int pix0, int chan0, int lane) {
    // channel chan0 + (lane & 15), pixels pix0 + 8*(lane >> 4) + 0..7 of a [64][128] bf16 half-tile
    const int s = lane & 15, q = lane >> 4;
    const int chan = chan0 + (s & 3) * 4;
    bf16x8 out;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int row = pix0 + q * 8 + u * 4 + (s >> 2);
        const int slot = (chan >> 3) ^ (((row & 3) << 2) | (((row >> 3) & 1) << 1));
        const unsigned char* p = half + row * 256 + slot * 16 + (chan & 7) * 2;
        const v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_ptr)p);
        out[u * 4 + 0] = r[0]; out[u * 4 + 1] = r[1]; out[u * 4 + 2] = r[2]; out[u * 4 + 3] = r[3];
    }
    return out;
}

// Loader state of one K-tile (64-pixel chunk).  Everything a stage needs per piece is carried incrementally - the
// input offset of the piece's pixel and its tap coordinates - so that a load phase issues adds and compares only: with
// the offsets recomputed from (n, ho, wo) every time (seven quarter-rate 32-bit multiplies and two 64-bit multiply-adds
// per x piece) the load phase of one wave group was longer than the 32-MFMA phase of its partner (round-2 ISA count:
// 84 + 61 + 30 VALU per K-tile against 30 + 28 in the forward kernel).
struct PixState {
    int t, k, sidx;       // ordinal of the K-tile in this workgroup's sweep; row / strip counters of the row-strip order
    long ch;              // K-tile (64-pixel chunk) index (uniform)
    int ho[2], wo[2];     // output pixel of the piece's row
    int hi[2], wi[2];     // its input coordinates for this workgroup's tap
    long xoff[2];         // element offset of input pixel (n, ho * stride, wo * stride)
};

// one workgroup's share of a weight gradient: tile `combo` = (tap, k tile, c tile) of pixel range `split`
__device__ __forceinline__ void wgrad256_tile(const WgradArgs& a, const int split, const int combo, unsigned char* smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int kt_n = a.K / 256, ct_n = a.C / 256, RS = a.R * a.S;
    const int ct = combo % ct_n;
    const int kt = (combo / ct_n) % kt_n;
    const int rs = combo / (ct_n * kt_n);
    const int r = rs / a.S, s = rs - r * a.S;
    const int k0 = kt * 256, c0 = ct * 256;
    // K-tile order.  Linear (1x1 layers, ragged widths): the split's 64-pixel chunks in memory order.  ROW-STRIP (round 4;
    // filters with taps, Wo % 64 == 0): the split owns whole output rows and sweeps them strip by strip - 64 columns, all
    // its rows top to bottom, then the next 64 columns.  The workgroups of the R*S taps of a split run side by side on one
    // XCD; tap (r, s) of K-tile t reads input row (row_t + r - 1): in row-strip order the row a tap needs was staged by
    // its neighbour tap ONE K-tile earlier (linear order: Wo / 64 = 8 K-tiles earlier on P2, i.e. 3.5 splits x 8 x 66 KB
    // apart in a 4 MiB L2 that also streams dy - rocprofv3 FETCH_SIZE of the P2 3x3 launch was 5.6x its operand bytes).
    const bool strip = a.strip_rows > 0;
    const int spr = a.Wo / WP;                                  // 64-column strips per output row (strip mode)
    const long nchunks = (a.P + WP - 1) / WP;
    long ch0, row0 = 0;
    int nk, nrows = 0;
    if (strip) {
        const long rows_total = (long)a.N * a.Ho;
        row0 = (long)split * a.strip_rows;
        const long row1 = row0 + a.strip_rows < rows_total ? row0 + a.strip_rows : rows_total;
        nrows = row1 > row0 ? (int)(row1 - row0) : 0;
        nk = nrows * spr;
        ch0 = row0 * spr;
    } else {
        ch0 = (long)split * a.chunks_per_split;
        const long ch1 = ch0 + a.chunks_per_split < nchunks ? ch0 + a.chunks_per_split : nchunks;
        nk = ch1 > ch0 ? (int)(ch1 - ch0) : 0;
    }

    // loader geometry: piece q = i*512 + tid -> pixel row q >> 4 (0..63) of the K-tile, 16-byte slot q & 15
    int lrow[2], lslot[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = i * 512 + tid;
        lrow[i] = q >> 4;
        lslot[i] = (q & 15) ^ (((lrow[i] & 3) << 2) | (((lrow[i] >> 3) & 1) << 1));
    }
    const int adv_h = WP / a.Wo, adv_w = WP - adv_h * a.Wo;     // 64 pixels = adv_h rows + adv_w columns
    // element strides of x per output column / output row / image, and what one chunk / one wrap adds (uniform)
    const long dW = (long)a.stride * a.C, dH = (long)a.stride * a.W * a.C, dN = (long)a.H * a.W * a.C;
    const long add_chunk = adv_w * dW + adv_h * dH, add_wrap_w = dH - a.Wo * dW, add_wrap_h = dN - a.Ho * dH;
    const int hi_chunk = adv_h * a.stride, wi_chunk = adv_w * a.stride, wi_wrap = a.Wo * a.stride, hi_wrap = a.Ho * a.stride;
    const long chunk_dy = (long)WP * a.K;
    // per-piece bases: the tap offset and the channel slot are folded into the pointers (integer arithmetic: a tap
    // offset may point before x for pieces whose bounds test fails - those read the zero line)
    uintptr_t xb[2], dyb[2];
    {
        const long toff = ((long)(r * a.dil - a.pad) * a.W + (s * a.dil - a.pad)) * a.C;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            xb[i] = (uintptr_t)a.x + (uintptr_t)((toff + c0 + lslot[i] * 8) * 2);
            dyb[i] = (uintptr_t)(a.dy + (size_t)lrow[i] * a.K + k0 + lslot[i] * 8);
        }
    }
    auto init_state = [&](PixState& st, long ch) {
        st.ch = ch;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long p = ch * WP + lrow[i];
            st.wo[i] = (int)(p % a.Wo);
            const long t = p / a.Wo;
            st.ho[i] = (int)(t % a.Ho);
            const long n = t / a.Ho;
            st.hi[i] = st.ho[i] * a.stride - a.pad + r * a.dil;
            st.wi[i] = st.wo[i] * a.stride - a.pad + s * a.dil;
            st.xoff[i] = n * dN + st.ho[i] * dH + st.wo[i] * dW;
        }
    };
    auto advance = [&](PixState& st) {
        st.t++;
        if (strip) {
            if (++st.k >= nrows) {                 // next strip of this split's rows (uniform, once per nrows K-tiles)
                st.k = 0;
                ++st.sidx;
                if (st.t < nk) init_state(st, row0 * spr + st.sidx);
            } else {                               // same columns, next output row
                st.ch += spr;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    ++st.ho[i]; st.hi[i] += a.stride;
                    st.xoff[i] += dH;
                    if (st.ho[i] >= a.Ho) {        // first row of the next image
                        st.ho[i] -= a.Ho; st.hi[i] -= hi_wrap;
                        st.xoff[i] += add_wrap_h;
                    }
                }
            }
            return;
        }
        st.ch++;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            st.wo[i] += adv_w; st.wi[i] += wi_chunk;
            st.ho[i] += adv_h; st.hi[i] += hi_chunk;
            st.xoff[i] += add_chunk;
            if (st.wo[i] >= a.Wo) {
                st.wo[i] -= a.Wo; st.wi[i] -= wi_wrap;
                ++st.ho[i]; st.hi[i] += a.stride;
                st.xoff[i] += add_wrap_w;
            }
            while (st.ho[i] >= a.Ho) {          // next image (several for maps smaller than a chunk)
                st.ho[i] -= a.Ho; st.hi[i] -= hi_wrap;
                st.xoff[i] += add_wrap_h;
            }
        }
    };
    auto stage_dy = [&](int h, const PixState& st, int buf) {
        unsigned char* dst = smem + buf * BUF_BYTES + h * HALF_BYTES + wave * 1024;
        const long left = a.P - st.ch * WP;                       // pixels from this chunk's first to the end (uniform)
        const int rows = st.t < nk ? (left < WP ? (int)left : WP) : 0;
        const uintptr_t off = (uintptr_t)(st.ch * chunk_dy + h * 128) * 2;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uintptr_t src = lrow[i] < rows ? dyb[i] + off : (uintptr_t)a.zeros;
            glds16((const void*)src, dst + i * 8192);
        }
    };
    auto stage_x = [&](int h, const PixState& st, int buf) {
        unsigned char* dst = smem + buf * BUF_BYTES + (2 + h) * HALF_BYTES + wave * 1024;
        const long left = a.P - st.ch * WP;
        const int rows = st.t < nk ? (left < WP ? (int)left : WP) : 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool ok = lrow[i] < rows && (unsigned)st.hi[i] < (unsigned)a.H && (unsigned)st.wi[i] < (unsigned)a.W;
            const uintptr_t src = ok ? xb[i] + (uintptr_t)((st.xoff[i] + h * 128) * 2) : (uintptr_t)a.zeros;
            glds16((const void*)src, dst + i * 8192);
        }
    };

    f32x4v acc[2][2][2][4];        // [x half][x tile][dy half][dy tile]
#pragma unroll
    for (int x0 = 0; x0 < 2; ++x0)
#pragma unroll
        for (int x1 = 0; x1 < 2; ++x1)
#pragma unroll
            for (int x2 = 0; x2 < 2; ++x2)
#pragma unroll
                for (int x3 = 0; x3 < 4; ++x3) acc[x0][x1][x2][x3] = f32x4v{0.f, 0.f, 0.f, 0.f};

    // fragment addresses: for a lane the swizzle term and the channel term do not depend on the k-step or on which
    // 4-pixel half of the fragment is read (those only add multiples of 4 rows), so one byte offset per 16-channel
    // tile is precomputed and every ds_read_b64_tr_b16 is base + immediate
    int dyoff[4], xoff[2];
    {
        const int s16 = lane & 15, q4 = lane >> 4;
        const int row = q4 * 8 + (s16 >> 2);
        const int sw = ((row & 3) << 2) | (((row >> 3) & 1) << 1);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int chan = wr * 64 + it * 16 + (s16 & 3) * 4;
            dyoff[it] = row * 256 + (((chan >> 3) ^ sw) << 4) + (chan & 7) * 2;
        }
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            const int chan = wc * 32 + jt * 16 + (s16 & 3) * 4;
            xoff[jt] = row * 256 + (((chan >> 3) ^ sw) << 4) + (chan & 7) * 2;
        }
    }
    auto frag = [&](const unsigned char* half, int off, int ks) {
        bf16x8 out;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s_ptr)(half + off + (ks * 32 + u * 4) * 256));
            out[u * 4 + 0] = r[0]; out[u * 4 + 1] = r[1]; out[u * 4 + 2] = r[2]; out[u * 4 + 3] = r[3];
        }
        return out;
    };
    bf16x8 df[4][2], xf[2][2][2];
    auto read_dy = [&](int h, int buf) {
        const unsigned char* base = smem + buf * BUF_BYTES + h * HALF_BYTES;
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) df[it][ks] = frag(base, dyoff[it], ks);
    };
    auto read_x = [&](int h, int buf) {
        const unsigned char* base = smem + buf * BUF_BYTES + (2 + h) * HALF_BYTES;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) xf[h][jt][ks] = frag(base, xoff[jt], ks);
    };
    // two 32-MFMA phases per pixel chunk, reads retired before the phase's first barrier (see conv_igemm256_kernel)
#define OADG_WMFMA32(X_FIRST, DH)                                                                            \
    do {                                                                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
        asm volatile("s_barrier" ::: "memory");                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        __builtin_amdgcn_s_setprio(1);                                                                       \
        _Pragma("unroll") for (int hh = 0; hh < 2; ++hh)                                                     \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                 \
                _Pragma("unroll") for (int jt = 0; jt < 2; ++jt)                                             \
                    _Pragma("unroll") for (int it = 0; it < 4; ++it)                                         \
                        acc[hh ^ X_FIRST][jt][DH][it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(             \
                            xf[hh ^ X_FIRST][jt][ks], df[it][ks], acc[hh ^ X_FIRST][jt][DH][it], 0, 0, 0);   \
        __builtin_amdgcn_s_setprio(0);                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        asm volatile("s_barrier" ::: "memory");                                                              \
    } while (0)

    PixState s1, s2;
    s1.t = s1.k = s1.sidx = 0;
    init_state(s1, ch0);
    stage_dy(0, s1, 0);
    stage_x(0, s1, 0);
    stage_x(1, s1, 0);
    stage_dy(1, s1, 0);
    advance(s1);
    stage_dy(0, s1, 1);
    stage_x(0, s1, 1);
    s2 = s1;
    advance(s2);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    if (wr == 1) asm volatile("s_barrier" ::: "memory");

    for (int t = 0; t < nk; ++t) {
        const int buf = t & 1;
        // phase A: both x halves + dy half 0; stages x1, dy1 of chunk t+1
        read_x(0, buf);
        read_x(1, buf);
        read_dy(0, buf);
        stage_x(1, s1, buf ^ 1);
        stage_dy(1, s1, buf ^ 1);
        OADG_WMFMA32(0, 0);
        // phase B: dy half 1; stages dy0, x0 of chunk t+2 (this buffer: read in phase A, retired before its barrier)
        read_dy(1, buf);
        stage_dy(0, s2, buf);
        stage_x(0, s2, buf);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        OADG_WMFMA32(1, 1);
        s1 = s2;
        advance(s2);
    }
#undef OADG_WMFMA32
    if (wr == 0) asm volatile("s_barrier" ::: "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // partial tile: D rows = x channels (4 consecutive per lane), columns = dy channels
    float* out = a.part + (size_t)split * a.K * RS * a.C;
    const int fr = lane & 15, cq = lane >> 4;
#pragma unroll
    for (int xh = 0; xh < 2; ++xh)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            const int c = c0 + xh * 128 + wc * 32 + jt * 16 + 4 * cq;
#pragma unroll
            for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int k = k0 + dh * 128 + wr * 64 + it * 16 + fr;
                    *reinterpret_cast<f32x4v*>(out + ((size_t)k * RS + rs) * a.C + c) = acc[xh][jt][dh][it];
                }
        }
}

__global__ __launch_bounds__(512) void conv_wgrad256_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int combos = (a.K / 256) * (a.C / 256) * a.R * a.S;
    const long bid = blockIdx.x;
    int split, combo;
    if (a.splits % 8 == 0) {
        const long xcd = bid & 7, j = bid >> 3;
        combo = (int)(j % combos);
        split = (int)((j / combos) * 8 + xcd);
    } else {
        // workgroup b runs on XCD b % 8: XCD x takes the CONTIGUOUS slice [x * per, (x + 1) * per) of the split-major
        // workgroup list, so the taps / tiles of a pixel range share an XCD's L2 except where a slice boundary cuts a
        // split (28 splits x 9 taps = 252 workgroups use 252 CUs; 24 aligned splits only 216)
        const long total = (long)a.splits * combos, per = (total + 7) >> 3;
        const long xcd = bid & 7, slot = bid >> 3, w = xcd * per + slot;
        if (slot >= per || w >= total) return;
        combo = (int)(w % combos);
        split = (int)(w / combos);
    }
    wgrad256_tile(a, split, combo, smem);
}

// ---- several layers' weight gradients in ONE launch (round 4) -----------------------------------------------------
// The small maps of the backbone / neck (layer3, layer4, P3-P5 laterals and output convolutions: 1024 or 256 K-tiles of
// 64 pixels each) cannot fill 256 compute units with long workgroups on their own: one round of 256 workgroups gives each
// 8-37 K-tiles between a ~4 us prologue and a 16 us partial-tile epilogue, and leaves 28-64 fp32 partial tiles per
// weight tile to be written, re-read and summed.  Their launches are independent, so a trainer defers them and issues a
// GROUP as one launch: job j owns the workgroups [first_block_j, first_block_j + tiles_j * splits_j) of one list,
// splits_j chosen by oadg_conv2d_wgrad_multi_plan so that the list has <= 256 entries of about equal length (>= 64
// K-tiles each when the group is large enough) - fewer, longer splits: 4-8 partials per weight tile instead of 28-64.
// XCD x takes a contiguous slice of the list (as in the single-layer kernel), so the tiles of a pixel range meet in
// one L2; the slices hold equal WORK, not equal counts (round 5: a list of several rounds whose long entries sat in one
// slice kept that XCD busy 40 % longer than the others).  The job table lives in device memory; a workgroup finds its job by binary search on first_block (uniform).
struct XcdSlices {
    int first[9];        // XCD x runs the list entries [first[x], first[x + 1])
};

__global__ __launch_bounds__(512) void conv_wgrad256_multi_kernel(const oadg_wgrad_job* __restrict__ jobs, int n_jobs,
                                                                  XcdSlices xs, const unsigned short* zeros) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, w = xs.first[xcd] + slot;
    if (w >= xs.first[xcd + 1]) return;
    int lo = 0, hi = n_jobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= w) lo = mid; else hi = mid - 1;
    }
    const oadg_wgrad_job& jb = jobs[__builtin_amdgcn_readfirstlane(lo)];
    WgradArgs a;
    a.x = (const unsigned short*)jb.x; a.dy = (const unsigned short*)jb.dy; a.part = (float*)jb.part; a.zeros = zeros;
    a.N = jb.N; a.H = jb.H; a.W = jb.W; a.C = jb.C; a.K = jb.K; a.R = jb.R; a.S = jb.S; a.Ho = jb.Ho; a.Wo = jb.Wo;
    a.stride = jb.stride; a.pad = jb.pad; a.dil = jb.dil; a.splits = jb.splits; a.chunks_per_split = jb.chunks_per_split;
    a.strip_rows = jb.strip_rows;
    a.P = jb.P;
    const int local = w - jb.first_block;
    const int combos = (a.K / 256) * (a.C / 256) * a.R * a.S;
    wgrad256_tile(a, local / combos, local % combos, smem);
}

__global__ void wgrad_reduce_kernel(const float* __restrict__ part, int splits, long n, float* __restrict__ dw) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += part[(size_t)k * n + i];
    dw[i] = s;
}

// Kernel / split choice (tools/bench_conv.py --wgrad sweeps it; times vs MIOpen's igemm_wrw on MI355X):
//  - K and C multiples of 256 and >= 8 K-tiles per workgroup at one round of workgroups: the 256-tile phase pipeline
//    (round 2, with the incremental loader: 1000-1115 TFLOP/s on the 3x3 layers from P2 to layer4, 45 us against 63 on
//    the 1x1 layers of layer3 / layer4);
//  - other 3x3: 128-tile, ONE LDS stage at 4 workgroups per CU, ~1024 workgroups (1.06-1.2x);
//  - 1x1: 128-tile, two stages, ~512 workgroups - the fp32 partial tiles (splits x K x C x 4 bytes, written and read
//    back by the reduction) are the cost that matters there (1.15x on layer2, 1.6x on layer3 / layer4).
// One workgroup of the 256-tile kernel per CU, ONE round of <= 256 workgroups: half the partial tiles of two rounds to
// write and read back, and since the loader carries its addresses incrementally (round 2) one round is as fast as two
// on every shape (P2 3x3: 1076 us with 28 splits against 1125 with 56; P3: 273 / 294; layer3 / P4: 76 / 90).  The
// workgroup list is split-major and every XCD runs a contiguous slice of it (see the kernel), so the taps of a pixel
// range share an XCD's L2 for any split count - 28 splits x 9 taps use 252 CUs where 24 XCD-aligned splits use 216.
long wgrad256_splits(long P, int K, int C, int RS) {
    const long tiles = (long)(K / 256) * (C / 256) * RS;
    long s = 256 / tiles;
    if (s < 1) s = 1;
    return s;
}
bool wgrad_use256(long P, int K, int C, int RS) {
    if (K % 256 != 0 || C % 256 != 0) return false;
    const long nchunks = (P + WP - 1) / WP;
    return nchunks / wgrad256_splits(P, K, C, RS) >= 8;      // at least 8 K-tiles per workgroup
}
int wgrad_stages(int RS) { return RS > 1 ? 1 : 2; }

// row-strip K-tile order of the 256-tile kernel (wgrad256_tile): filters with taps on maps whose width is a multiple of
// the 64-pixel K-tile.  A split becomes a number of whole output rows; returns that number (0: linear order) and
// replaces splits / chunks_per_split by the equivalent row-aligned values (never more splits than before).
int wgrad_strip(int N, int Ho, int Wo, int RS, int& splits, int& chunks_per_split) {
    const long rows = (long)N * Ho;
    if (RS <= 1 || Wo % WP != 0 || rows < splits || splits < 1) return 0;
    const long per = (rows + splits - 1) / splits;
    splits = (int)((rows + per - 1) / per);
    chunks_per_split = (int)(per * (Wo / WP));
    return (int)per;
}

int wgrad_splits(long P, int K, int C, int RS) {
    if (wgrad_use256(P, K, C, RS)) return (int)wgrad256_splits(P, K, C, RS);
    const long tiles = (long)(K / 128) * (C / 128) * RS;
    const long nchunks = (P + WP - 1) / WP;
    // (round 4, operands from HBM: 512 workgroups beat 1024 on the 3x3 layers too - layer2 3x3 164 / 178 us, its stride-2
    //  first block 178 / 210 us incl. the reduction - and leave half the partial tiles to the consumer)
    const long target = 512;
    long s = (target + tiles - 1) / tiles;
    if (s > nchunks / 4) s = nchunks / 4;
    if (s >= 8) s = (s + 7) / 8 * 8;       // multiples of 8: one pixel range per XCD at a time
    if (s < 1) s = 1;
    if (s > 256) s = 256;
    return (int)s;
}

}  // namespace

extern "C" size_t oadg_conv2d_wgrad_workspace_bytes(int N, int Ho, int Wo, int C, int K, int R, int S) {
    if (C % 128 || K % 128) return 0;
    const int sp = wgrad_splits((long)N * Ho * Wo, K, C, R * S);
    return (size_t)sp * K * R * S * C * sizeof(float);
}

// which kernel a weight-gradient problem runs on: 256 = conv_wgrad256_kernel, 2 / 1 = conv_wgrad_kernel<2> / <1>
// (LDS stages), 0 = shape not covered.  (bench.py names its live event pairs with this.)
extern "C" int oadg_conv2d_wgrad_variant(int N, int Ho, int Wo, int C, int K, int R, int S) {
    if (C % 128 || K % 128) return 0;
    if (wgrad_use256((long)N * Ho * Wo, K, C, R * S)) return 256;
    return wgrad_stages(R * S);
}

namespace {
int wgrad_launch(const void* x, const void* dy, float* dw, const void* zeros16, void* workspace,
                 size_t workspace_bytes, int N, int H, int W, int C, int K, int R, int S, int stride, int pad, int dil,
                 int* splits_out, void* stream) {
    if (!x || !dy || !zeros16 || !workspace) return OADG_EARG;
    if (C % 128 != 0 || K % 128 != 0 || N < 1 || R < 1 || S < 1) return OADG_EARG;
    if ((double)N * H * W * C >= 4294967296.0) return OADG_EARG;      // the 128-tile loader keeps 32-bit element offsets
    WgradArgs a;
    a.x = (const unsigned short*)x; a.dy = (const unsigned short*)dy; a.part = (float*)workspace;
    a.zeros = (const unsigned short*)zeros16;
    a.N = N; a.H = H; a.W = W; a.C = C; a.K = K; a.R = R; a.S = S; a.stride = stride; a.pad = pad; a.dil = dil;
    a.Ho = (H + 2 * pad - dil * (R - 1) - 1) / stride + 1;
    a.Wo = (W + 2 * pad - dil * (S - 1) - 1) / stride + 1;
    a.P = (long)N * a.Ho * a.Wo;
    a.splits = wgrad_splits(a.P, K, C, R * S);
    const long nchunks = (a.P + WP - 1) / WP;
    a.chunks_per_split = (int)((nchunks + a.splits - 1) / a.splits);
    const size_t need = (size_t)a.splits * K * R * S * C * sizeof(float);
    if (workspace_bytes < need) return OADG_ESIZE;
    hipStream_t st = (hipStream_t)stream;
    a.strip_rows = 0;
    if (wgrad_use256(a.P, K, C, R * S)) {
        a.strip_rows = wgrad_strip(N, a.Ho, a.Wo, R * S, a.splits, a.chunks_per_split);
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute((const void*)conv_wgrad256_kernel,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF_BYTES);
            if (e != hipSuccess) return (int)e;
            attr_set = true;
        }
        long blocks = (long)a.splits * (K / 256) * (C / 256) * R * S;
        if (a.splits % 8 != 0) blocks = ((blocks + 7) / 8) * 8;          // eight equal XCD slices (the kernel drops the padding)
        hipLaunchKernelGGL(conv_wgrad256_kernel, dim3((unsigned)blocks), dim3(512), 2 * BUF_BYTES, st, a);
    } else {
        const long blocks = (long)a.splits * (K / 128) * (C / 128) * R * S;
        if (wgrad_stages(R * S) == 2)
            hipLaunchKernelGGL(conv_wgrad_kernel<2>, dim3((unsigned)blocks), dim3(256), 2 * WSTAGE, st, a);
        else
            hipLaunchKernelGGL(conv_wgrad_kernel<1>, dim3((unsigned)blocks), dim3(256), WSTAGE, st, a);
    }
    OADG_LAUNCH_CHECK();
    if (splits_out) {                 // the caller consumes the partial tiles itself (oadg_prep_conv_weights_bwd_parts)
        *splits_out = a.splits;
        return OADG_OK;
    }
    const long n = (long)K * R * S * C;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                       (const float*)workspace, a.splits, n, dw);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}
}  // namespace

// dw [K,R,S,C] fp32 (overwritten).  x [N,H,W,C] bf16, dy [N,Ho,Wo,K] bf16.  Requires C % 128 == 0, K % 128 == 0.
extern "C" int oadg_conv2d_wgrad_nhwc_bf16(const void* x, const void* dy, float* dw, const void* zeros16,
                                           void* workspace, size_t workspace_bytes, int N, int H, int W, int C,
                                           int K, int R, int S, int stride, int pad, int dil, void* stream) {
    if (!dw) return OADG_EARG;
    return wgrad_launch(x, dy, dw, zeros16, workspace, workspace_bytes, N, H, W, C, K, R, S, stride, pad, dil, nullptr,
                        stream);
}

// Partial tiles only: workspace = [*splits][K][R*S][C] fp32, to be summed by oadg_prep_conv_weights_bwd_parts (which
// also applies the BN-fold chain rule and the layout change) - no separate reduction, no bf16 round trip.
extern "C" int oadg_conv2d_wgrad_parts_nhwc_bf16(const void* x, const void* dy, const void* zeros16, void* workspace,
                                                 size_t workspace_bytes, int N, int H, int W, int C, int K, int R,
                                                 int S, int stride, int pad, int dil, int* splits, void* stream) {
    if (!splits) return OADG_EARG;
    return wgrad_launch(x, dy, nullptr, zeros16, workspace, workspace_bytes, N, H, W, C, K, R, S, stride, pad, dil,
                        splits, stream);
}

// ---- grouped launch (see conv_wgrad256_multi_kernel).  Plan on the host: jobs_host[i] carries the problem (x, dy, N, H,
// W, C, K, R, S, stride, pad, dil); this fills Ho, Wo, P, splits, chunks_per_split, first_block, blocks and returns the
// length of the workgroup list (<= target_blocks whenever the group's weight tiles fit, else one workgroup per tile), or
// a negative OADG_E* code.  The caller then points part at splits * K * R * S * C floats per job, copies the table to the
// device and launches.  Every job must be a shape the 256-tile kernel covers (K % 256 == 0, C % 256 == 0).
extern "C" long oadg_conv2d_wgrad_multi_plan(oadg_wgrad_job* jobs, int n, int target_blocks, int* xcd_first) {
    if (!jobs || n < 1 || target_blocks < 1) return -(long)OADG_EARG;
    double work = 0.0;
    for (int i = 0; i < n; ++i) {
        oadg_wgrad_job& j = jobs[i];
        if (j.N < 1 || j.H < 1 || j.W < 1 || j.R < 1 || j.S < 1 || j.stride < 1 || j.dil < 1 || j.pad < 0) return -(long)OADG_EARG;
        if (j.K % 256 != 0 || j.C % 256 != 0 || j.K < 256 || j.C < 256) return -(long)OADG_EARG;
        if ((double)j.N * j.H * j.W * j.C >= 4294967296.0) return -(long)OADG_EARG;
        j.Ho = (j.H + 2 * j.pad - j.dil * (j.R - 1) - 1) / j.stride + 1;
        j.Wo = (j.W + 2 * j.pad - j.dil * (j.S - 1) - 1) / j.stride + 1;
        if (j.Ho < 1 || j.Wo < 1) return -(long)OADG_EARG;
        j.P = (long)j.N * j.Ho * j.Wo;
        const long tiles = (long)(j.K / 256) * (j.C / 256) * j.R * j.S, nchunks = (j.P + WP - 1) / WP;
        work += (double)tiles * (double)nchunks;
    }
    // Split counts for a list of (about) `target` entries: K-tiles per workgroup if the list had exactly `target` entries;
    // every job starts with the split count that stays at or above it, then the job with the longest workgroups takes one
    // more split while the list has room.
    auto assign = [&](long target, int* sp_out) -> long {
        const double per = work / (double)target;
        long total = 0;
        for (int i = 0; i < n; ++i) {
            const oadg_wgrad_job& j = jobs[i];
            const long nchunks = (j.P + WP - 1) / WP;
            long sp = per > 0.0 ? (long)((double)nchunks / per) : 1;
            if (sp < 1) sp = 1;
            if (sp > nchunks) sp = nchunks;
            sp_out[i] = (int)sp;
            total += (long)(j.K / 256) * (j.C / 256) * j.R * j.S * sp;
        }
        while (total > target) {              // (jobs rounded up to one split pushed the list over: take splits back where
            int best = -1;                    //  the workgroups are shortest)
            double shortest = 1e300;
            for (int i = 0; i < n; ++i) {
                const oadg_wgrad_job& j = jobs[i];
                if (sp_out[i] <= 1) continue;
                const double len = (double)((j.P + WP - 1) / WP) / sp_out[i];
                if (len < shortest) { shortest = len; best = i; }
            }
            if (best < 0) break;
            sp_out[best] -= 1;
            total -= (long)(jobs[best].K / 256) * (jobs[best].C / 256) * jobs[best].R * jobs[best].S;
        }
        for (;;) {
            int best = -1;
            double longest = 0.0;
            for (int i = 0; i < n; ++i) {
                const oadg_wgrad_job& j = jobs[i];
                const long tiles = (long)(j.K / 256) * (j.C / 256) * j.R * j.S, nchunks = (j.P + WP - 1) / WP;
                if (sp_out[i] >= nchunks || total + tiles > target) continue;
                const double len = (double)nchunks / sp_out[i];
                if (len > longest) { longest = len; best = i; }
            }
            if (best < 0) break;
            sp_out[best] += 1;
            total += (long)(jobs[best].K / 256) * (jobs[best].C / 256) * jobs[best].R * jobs[best].S;
        }
        return total;
    };
    // Round 5: ONE round of <= target_blocks workgroups is not always the shortest launch.  A group with many weight tiles
    // of few K-tiles each (layer4: 228 tiles of 256 K-tiles + two layers of 1024) has no room to split its long jobs - the
    // list was 228 entries of 256 K-tiles, 12 of 341 and 16 of 512: the launch lasted 512 K-tiles at an average of 276
    // (0.95 ms at 622 TFLOP/s where the kernel runs 1100).  So the planner now also tries lists of 1.5 - 3 rounds and keeps
    // the one whose SIMULATED launch is shortest: the kernel's dispatch replayed on the host - eight XCD slices of the list,
    // 32 compute units each, a free unit takes the slice's next entry - with an entry costing its K-tiles + 11 (prologue +
    // partial-tile epilogue, ~20 us at 1.86 us per K-tile) and 0.1 K-tile per entry for the consumer's extra partial tile.
    // OADG_WGRAD_PLAN_ROUNDS=1: one round only (A/B probes).
    // Eight contiguous slices of the list, one per XCD (its compute units take the slice's entries in order, a free unit
    // the next one): the partition with the shortest simulated launch - greedy filling is optimal for a contiguous
    // partition under a time limit, the limit is found by bisection.  Returns that launch length (in K-tiles).
    const int cus_per_xcd = target_blocks >= 8 ? target_blocks / 8 : 1;
    auto slices = [cus_per_xcd](const std::vector<double>& len, long* fx) -> double {
        const long cnt = (long)len.size();
        // even = true: an XCD also stops at its share of the work that is left (same limit, later XCDs not left idle)
        auto fill = [&](double limit, long* out, bool even) -> bool {
            long w = 0;
            double left = 0.0;
            for (double v : len) left += v;
            std::vector<double> cu((size_t)cus_per_xcd);
            for (int x = 0; x < 8; ++x) {
                out[x] = w;
                std::fill(cu.begin(), cu.end(), 0.0);
                const double share = left / (8 - x);
                double mine = 0.0;
                while (w < cnt) {
                    if (even && mine >= share) break;
                    size_t m = 0;
                    for (size_t u = 1; u < cu.size(); ++u)
                        if (cu[u] < cu[m]) m = u;
                    if (cu[m] + len[(size_t)w] > limit) break;
                    cu[m] += len[(size_t)w];
                    mine += len[(size_t)w++];
                }
                left -= mine;
            }
            out[8] = w;
            return w == cnt;
        };
        double lo = 0.0, hi = 0.0, tot = 0.0;
        for (double v : len) { tot += v; lo = v > lo ? v : lo; }
        hi = tot + 1.0;                                   // (one unit running everything: always feasible)
        lo = lo > tot / (8.0 * cus_per_xcd) ? lo : tot / (8.0 * cus_per_xcd);
        long tmp[9];
        if (fill(lo, tmp, false)) hi = lo;
        for (int it = 0; it < 40 && hi - lo > 0.5; ++it) {
            const double mid = 0.5 * (lo + hi);
            if (fill(mid, tmp, false)) hi = mid; else lo = mid;
        }
        if (!fill(hi, fx, true)) fill(hi, fx, false);
        return hi;
    };
    std::vector<int> sp_best(n), sp_try(n);
    double best_cost = 1e300;
    static const bool one_round = getenv("OADG_WGRAD_PLAN_ROUNDS") && atoi(getenv("OADG_WGRAD_PLAN_ROUNDS")) == 1;
    const double halves[5] = {2, 3, 4, 5, 6};                  // list lengths in half rounds
    for (int c = 0; c < (one_round ? 1 : 5); ++c) {
        const long target = (long)(target_blocks * halves[c] / 2);
        const long total = assign(target, sp_try.data());
        std::vector<double> len;
        len.reserve((size_t)total);
        for (int i = 0; i < n; ++i) {
            const oadg_wgrad_job& j = jobs[i];
            const long tiles = (long)(j.K / 256) * (j.C / 256) * j.R * j.S, nchunks = (j.P + WP - 1) / WP;
            const long cps = (nchunks + sp_try[i] - 1) / sp_try[i], eff = (nchunks + cps - 1) / cps;
            for (long sidx = 0; sidx < eff; ++sidx) {
                const long mine = nchunks - sidx * cps < cps ? nchunks - sidx * cps : cps;
                for (long t = 0; t < tiles; ++t) len.push_back((double)mine + 11.0);
            }
        }
        const long cnt = (long)len.size();
        long fx[9];
        const double makespan = slices(len, fx);
        const double cost = makespan + 0.1 * (double)cnt;
        if (getenv("OADG_WGRAD_PLAN_DEBUG")) fprintf(stderr, "plan: target %ld entries %ld makespan %.0f cost %.0f\n", target, cnt, makespan, cost);
        if (cost < best_cost * 0.97) {                          // (a longer list has to pay: 3 % at least)
            best_cost = cost;
            sp_best = sp_try;
        }
        bool saturated = true;                                  // every job is split as far as it goes already
        for (int i = 0; i < n; ++i) saturated = saturated && sp_try[i] >= (jobs[i].P + WP - 1) / WP;
        if (saturated) break;
    }
    for (int i = 0; i < n; ++i) jobs[i].splits = sp_best[i];
    long first = 0;
    for (int i = 0; i < n; ++i) {
        oadg_wgrad_job& j = jobs[i];
        const long tiles = (long)(j.K / 256) * (j.C / 256) * j.R * j.S, nchunks = (j.P + WP - 1) / WP;
        j.chunks_per_split = (int)((nchunks + j.splits - 1) / j.splits);
        j.splits = (int)((nchunks + j.chunks_per_split - 1) / j.chunks_per_split);      // no empty pixel range
        j.strip_rows = wgrad_strip(j.N, j.Ho, j.Wo, j.R * j.S, j.splits, j.chunks_per_split);
        j.pad_ = 0;
        j.first_block = (int)first;
        j.blocks = (int)(tiles * j.splits);
        first += j.blocks;
        if (first > 0x3fffffffL) return -(long)OADG_EARG;
    }
    if (xcd_first) {
        std::vector<double> len;
        len.reserve((size_t)first);
        for (int i = 0; i < n; ++i) {
            const oadg_wgrad_job& j = jobs[i];
            const long tiles = (long)(j.K / 256) * (j.C / 256) * j.R * j.S, nchunks = (j.P + WP - 1) / WP;
            for (long sidx = 0; sidx < j.splits; ++sidx) {
                const long mine = nchunks - sidx * j.chunks_per_split < j.chunks_per_split ? nchunks - sidx * j.chunks_per_split
                                                                                             : j.chunks_per_split;
                for (long t = 0; t < tiles; ++t) len.push_back((double)mine + 11.0);
            }
        }
        long fx[9];
        slices(len, fx);
        for (int x = 0; x < 9; ++x) xcd_first[x] = (int)fx[x];
    }
    return first;
}

extern "C" int oadg_conv2d_wgrad_multi(const oadg_wgrad_job* jobs_dev, int n, int total_blocks, const int* xcd_first,
                                       const void* zeros16, void* stream) {
    if (!jobs_dev || n < 1 || total_blocks < 1 || !zeros16) return OADG_EARG;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)conv_wgrad256_multi_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    XcdSlices xs;
    int longest = 0;
    for (int x = 0; x <= 8; ++x) {
        // (host array from the plan; NULL: eight slices of equal length)
        xs.first[x] = xcd_first ? xcd_first[x] : (int)((long)((total_blocks + 7) / 8) * x < total_blocks ? ((total_blocks + 7) / 8) * x : total_blocks);
        if (x > 0) {
            if (xs.first[x] < xs.first[x - 1]) return OADG_EARG;
            longest = xs.first[x] - xs.first[x - 1] > longest ? xs.first[x] - xs.first[x - 1] : longest;
        }
    }
    if (xs.first[0] != 0 || xs.first[8] != total_blocks) return OADG_EARG;
    hipLaunchKernelGGL(conv_wgrad256_multi_kernel, dim3((unsigned)longest * 8), dim3(512), 2 * BUF_BYTES, (hipStream_t)stream,
                       jobs_dev, n, xs, (const unsigned short*)zeros16);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

// ================================================================================================ weight preparation
// One launch per layer and step instead of ~10 element-wise launches: fold the eval-mode BatchNorm of
// mmdet/models/backbones/resnet.py:648-657 into the convolution weight (w * gamma / sqrt(var + eps)) and bias
// (beta - mean * scale), cast to bf16, lay out as KRSC for the forward kernel and as the flipped/transposed
// C(R)(S)K copy the stride-1 data gradient consumes.  Its backward: d w = d wf * scale,
// d gamma = (sum d wf * w - d bias * mean) / sqrt(var + eps), d beta = d bias.
namespace {

__device__ __forceinline__ void prep_weights_channel(int k, const float* __restrict__ w, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, const float* __restrict__ mean,
                                                     const float* __restrict__ var, float eps,
                                                     const float* __restrict__ bias_in, int K, int C, int R, int S,
                                                     unsigned short* __restrict__ wf, unsigned short* __restrict__ wt,
                                                     float* __restrict__ bias, float* __restrict__ scale_out, int w_krsc,
                                                     int wt_mode) {
    float scale = 1.f, b = bias_in ? bias_in[k] : 0.f;
    if (gamma) {
        scale = gamma[k] * rsqrtf(var[k] + eps);
        b = beta[k] - mean[k] * scale;
    }
    if (threadIdx.x == 0) {
        if (bias) bias[k] = b;
        if (scale_out) scale_out[k] = scale;
    }
    const int RS = R * S, n = C * RS;
    const float* wk = w + (size_t)k * n;
    for (int i = threadIdx.x; i < n; i += 256) {      // i runs over the OUTPUT order (r, s, c): coalesced writes
        const int c = i % C, rs = i / C;
        const unsigned short v = f32_to_bf16(wk[w_krsc ? (size_t)i : (size_t)c * RS + rs] * scale);
        wf[(size_t)k * n + i] = v;
        if (wt && wt_mode != 2) {
            wt[((size_t)c * RS + (RS - 1 - rs)) * K + k] = v;
        } else if (wt) {
            // stride-2 data gradient as four stride-1 convolutions over dy, one per output parity class (ph, pw), class
            // blocks [C][taps][K] back to back in the order (0,0) (0,1) (1,0) (1,1).  3x3 / pad 1: parity 0 uses the
            // centre tap, parity 1 the taps 2 (dy[a]) and 0 (dy[a+1]) in that order; 1x1: a single class.
            if (RS == 1) {
                wt[(size_t)c * K + k] = v;
            } else {
                const int r = rs / S, q = rs - r * S;
                const int ph = r == 1 ? 0 : 1, pw = q == 1 ? 0 : 1;
                const int tr = r == 0 ? 1 : 0, tq = q == 0 ? 1 : 0;          // r = 1 -> 0, r = 2 -> 0, r = 0 -> 1
                const int Sc = pw ? 2 : 1, T = (ph ? 2 : 1) * Sc;
                const int cls_off = ph == 0 ? (pw == 0 ? 0 : 1) : (pw == 0 ? 3 : 5);
                wt[(size_t)cls_off * C * K + ((size_t)c * T + tr * Sc + tq) * K + k] = v;
            }
        }
    }
}

__global__ __launch_bounds__(256) void prep_weights_kernel(const float* __restrict__ w, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ mean,
                                                           const float* __restrict__ var, float eps,
                                                           const float* __restrict__ bias_in, int K, int C, int R, int S,
                                                           unsigned short* __restrict__ wf, unsigned short* __restrict__ wt,
                                                           float* __restrict__ bias, float* __restrict__ scale_out,
                                                           int w_krsc, int wt_mode) {
    prep_weights_channel(blockIdx.x, w, gamma, beta, mean, var, eps, bias_in, K, C, R, S, wf, wt, bias, scale_out, w_krsc,
                         wt_mode);
}

// every prepared layer of a model in ONE launch (weights change only in optimizer.step(): hip_conv.refresh_prepared
// re-prepares all of them right after it instead of one launch per layer inside the next forward pass): workgroup b
// belongs to the layer with first_block <= b < first_block + K (binary search over the descriptor table)
// Tiled form for layers with K % 32 == 0: a workgroup prepares 32 output channels x 256 consecutive (r, s, c) positions.
// The per-channel form writes the data-gradient copy wt [C][taps][K] with ONE 2-byte store per element, K * 2 bytes apart
// (24 M scattered stores per step: the multi-layer launch ran at 0.6 TB/s); here the tile goes through LDS and leaves as
// 16-byte pieces of 8 consecutive k.  Same values (same fp32 product, same rounding).
constexpr int PT_K = 32, PT_I = 256, PT_LD = PT_I + 8;
__device__ __forceinline__ void prep_weights_tile(int tile, const oadg_prep_desc& d) {
    __shared__ unsigned short T[PT_K][PT_LD];
    __shared__ float sc[PT_K];
    const int K = d.K, C = d.C, RS = d.R * d.S, S = d.S, n = C * RS;
    const int itiles = (n + PT_I - 1) / PT_I;
    const int k0 = (tile / itiles) * PT_K, i0 = (tile % itiles) * PT_I;
    const int tid = threadIdx.x;
    if (tid < PT_K) {
        const int k = k0 + tid;
        float scale = 1.f, b = d.bias_in ? d.bias_in[k] : 0.f;
        if (d.gamma) {
            scale = d.gamma[k] * rsqrtf(d.var[k] + d.eps);
            b = d.beta[k] - d.mean[k] * scale;
        }
        sc[tid] = scale;
        if (i0 == 0) {
            if (d.bias) d.bias[k] = b;
            if (d.scale) d.scale[k] = scale;
        }
    }
    __syncthreads();
    const int i = i0 + tid;
    unsigned short* wf = (unsigned short*)d.wf;
    if (i < n) {
        const int c = i % C, rs = i / C;
        const size_t src = d.w_krsc ? (size_t)i : (size_t)c * RS + rs;
#pragma unroll 4
        for (int kk = 0; kk < PT_K; ++kk) {
            const unsigned short v = f32_to_bf16(d.w[(size_t)(k0 + kk) * n + src] * sc[kk]);
            T[kk][tid] = v;
            wf[(size_t)(k0 + kk) * n + i] = v;
        }
    }
    if (!d.wt) return;
    __syncthreads();
    unsigned short* wt = (unsigned short*)d.wt;
    for (int j = tid; j < PT_I * (PT_K / 8); j += 256) {
        const int il = j >> 2, kq = (j & 3) * 8;
        const int ii = i0 + il;
        if (ii >= n) continue;
        const int c = ii % C, rs = ii / C;
        size_t dst;
        if (d.wt_mode != 2) {
            dst = ((size_t)c * RS + (RS - 1 - rs)) * K;
        } else if (RS == 1) {
            dst = (size_t)c * K;
        } else {
            const int r = rs / S, q = rs - r * S;
            const int ph = r == 1 ? 0 : 1, pw = q == 1 ? 0 : 1;
            const int tr = r == 0 ? 1 : 0, tq = q == 0 ? 1 : 0;
            const int Sc = pw ? 2 : 1, Tt = (ph ? 2 : 1) * Sc;
            const int cls_off = ph == 0 ? (pw == 0 ? 0 : 1) : (pw == 0 ? 3 : 5);
            dst = (size_t)cls_off * C * K + ((size_t)c * Tt + tr * Sc + tq) * K;
        }
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (short)T[kq + e][il];
        *reinterpret_cast<bf16x8*>(wt + dst + k0 + kq) = v;
    }
}

// every prepared layer of a model in ONE launch (weights change only in optimizer.step(): hip_conv.refresh_prepared
// re-prepares all of them right after it instead of one launch per layer inside the next forward pass): workgroup b
// belongs to the layer with first_block <= b < first_block + blocks (binary search over the descriptor table); a layer
// has (K / 32) * ceil(C R S / 256) tile workgroups when K % 32 == 0, else one workgroup per output channel
__global__ __launch_bounds__(256) void prep_weights_multi_kernel(const oadg_prep_desc* __restrict__ descs, int n_layers) {
    int lo = 0, hi = n_layers - 1;
    const int b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].first_block <= b) lo = mid; else hi = mid - 1;
    }
    const oadg_prep_desc d = descs[lo];
    if (d.K % PT_K == 0) {
        prep_weights_tile(b - d.first_block, d);
        return;
    }
    prep_weights_channel(b - d.first_block, d.w, d.gamma, d.beta, d.mean, d.var, d.eps, d.bias_in, d.K, d.C, d.R, d.S,
                         (unsigned short*)d.wf, (unsigned short*)d.wt, d.bias, d.scale, d.w_krsc, d.wt_mode);
}

__global__ __launch_bounds__(256) void prep_weights_bwd_kernel(const unsigned short* __restrict__ gwf,
                                                               const float* __restrict__ gbias,
                                                               const float* __restrict__ w,
                                                               const float* __restrict__ scale,
                                                               const float* __restrict__ mean,
                                                               const float* __restrict__ var, float eps, int K, int C,
                                                               int R, int S, float* __restrict__ dw,
                                                               float* __restrict__ dgamma, int w_flags) {
    __shared__ float red[16];
    const int w_krsc = w_flags & 1;
    const int k = blockIdx.x;
    const int RS = R * S, n = C * RS;
    const float sc = scale ? scale[k] : 1.f;
    const float* wk = w + (size_t)k * n;
    float dot = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {      // i over w's memory order: coalesced reads of w / writes of dw
        const int rs = i % RS, c = i / RS;
        const float g = bf16_to_f32(gwf[(size_t)k * n + (w_krsc ? (size_t)i : (size_t)rs * C + c)]);
        dw[(size_t)k * n + i] = g * sc;
        dot += g * wk[i];
    }
    if (dgamma) {
        const float tot = block_sum(dot, red);
        // (w_krsc bit 1: the bias gradient is not reduced yet - store the raw dot product; oadg_colsum_reduce_multi
        // finishes dgamma with the same expression once the column sums exist)
        if (threadIdx.x == 0)
            dgamma[k] = (w_flags & 2) ? tot : (tot - (gbias ? gbias[k] : 0.f) * mean[k]) * rsqrtf(var[k] + eps);
    }
}

}  // namespace

extern "C" int oadg_prep_conv_weights(const float* w, const float* gamma, const float* beta, const float* mean,
                                      const float* var, float eps, const float* bias_in, int K, int C, int R, int S,
                                      void* wf, void* wt, float* bias, float* scale, int w_krsc, int wt_mode,
                                      void* stream) {
    if (!w || !wf || K < 1 || C < 1 || R < 1 || S < 1) return OADG_EARG;
    if (wt && wt_mode == 2 && !((R == 3 && S == 3) || (R == 1 && S == 1))) return OADG_EARG;
    if (gamma && (!beta || !mean || !var)) return OADG_EARG;
    hipLaunchKernelGGL(prep_weights_kernel, dim3(K), dim3(256), 0, (hipStream_t)stream, w, gamma, beta, mean, var, eps,
                       bias_in, K, C, R, S, (unsigned short*)wf, (unsigned short*)wt, bias, scale, w_krsc, wt_mode);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

// descs (device memory): n_layers descriptors sorted by first_block, first_block = sum of K over the layers before;
// total_blocks = sum of oadg_prep_conv_weights_multi_blocks over the layers.  Same arithmetic per layer as
// oadg_prep_conv_weights.
extern "C" int oadg_prep_conv_weights_multi_blocks(int K, int C, int R, int S) {
    if (K % PT_K == 0) return (K / PT_K) * ((C * R * S + PT_I - 1) / PT_I);
    return K;
}
extern "C" int oadg_prep_conv_weights_multi(const oadg_prep_desc* descs, int n_layers, int total_blocks, void* stream) {
    if (!descs || n_layers < 1 || total_blocks < 1) return OADG_EARG;
    hipLaunchKernelGGL(prep_weights_multi_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, descs, n_layers);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

namespace {
// same as prep_weights_bwd_kernel with the weight gradient given as fp32 split partials [splits][K][R*S][C]: phase 1
// sums the splits in (rs, c) order (coalesced reads) into LDS, phase 2 walks (c, rs) order for the coalesced dw write.
__global__ __launch_bounds__(1024) void prep_weights_bwd_parts_kernel(const float* __restrict__ part, int splits,
                                                                     const float* __restrict__ gbias,
                                                                     const float* __restrict__ w,
                                                                     const float* __restrict__ scale,
                                                                     const float* __restrict__ mean,
                                                                     const float* __restrict__ var, float eps, int K,
                                                                     int C, int R, int S, float* __restrict__ dw,
                                                                     float* __restrict__ dgamma, int w_flags) {
    extern __shared__ float gsum[];            // [R*S][C]
    __shared__ float red[16];
    const int w_krsc = w_flags & 1;
    const int k = blockIdx.x;
    const int RS = R * S, n = C * RS;
    const size_t stride = (size_t)K * n;
    const float* p0 = part + (size_t)k * n;
    // 1024 threads: thread group q = tid / 256 takes every 4th split, 8 loads in flight per thread (the sum order is
    // fixed: deterministic)
    const int q4 = threadIdx.x >> 8, t256 = threadIdx.x & 255;
    float* gq = gsum + n;                        // [4][n] partial sums of the four split groups
    for (int j = t256; j < n; j += 256) {
        float acc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int sp = q4;
        for (; sp + 28 < splits; sp += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc8[u] += p0[(size_t)(sp + 4 * u) * stride + j];
        }
        for (int u = 0; sp < splits; sp += 4, ++u) acc8[u] += p0[(size_t)sp * stride + j];
        gq[q4 * n + j] = ((acc8[0] + acc8[1]) + (acc8[2] + acc8[3])) + ((acc8[4] + acc8[5]) + (acc8[6] + acc8[7]));
    }
    __syncthreads();
    for (int j = threadIdx.x; j < n; j += 1024) gsum[j] = (gq[j] + gq[n + j]) + (gq[2 * n + j] + gq[3 * n + j]);
    __syncthreads();
    const float sc = scale ? scale[k] : 1.f;
    const float* wk = w + (size_t)k * n;
    float dot = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const int rs = i % RS, c = i / RS;
        const float g = gsum[w_krsc ? i : rs * C + c];
        dw[(size_t)k * n + i] = g * sc;
        dot += g * wk[i];
    }
    if (dgamma) {
        const float tot = block_sum(dot, red);
        // (w_krsc bit 1: the bias gradient is not reduced yet - store the raw dot product; oadg_colsum_reduce_multi
        // finishes dgamma with the same expression once the column sums exist)
        if (threadIdx.x == 0)
            dgamma[k] = (w_flags & 2) ? tot : (tot - (gbias ? gbias[k] : 0.f) * mean[k]) * rsqrtf(var[k] + eps);
    }
}
}  // namespace

extern "C" int oadg_prep_conv_weights_bwd_parts(const float* part, int splits, const float* gbias, const float* w,
                                                const float* scale, const float* mean, const float* var, float eps,
                                                int K, int C, int R, int S, float* dw, float* dgamma, int w_krsc,
                                                void* stream) {
    if (!part || splits < 1 || !w || !dw) return OADG_EARG;
    if (dgamma && (!mean || !var)) return OADG_EARG;
    const size_t lds = 5 * (size_t)C * R * S * sizeof(float);
    if (lds > 5 * 12000) return OADG_EARG;    // callers fall back to the reduced form
    hipLaunchKernelGGL(prep_weights_bwd_parts_kernel, dim3(K), dim3(1024), lds, (hipStream_t)stream, part, splits, gbias,
                       w, scale, mean, var, eps, K, C, R, S, dw, dgamma, w_krsc);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

namespace {
// prep_weights_bwd_parts_kernel for the jobs of a grouped weight-gradient launch: few splits (1-8), so one pass of 512
// threads with eight independent accumulators per element (fixed order: deterministic) and ONE fp32 copy of the filter
// in LDS ([R*S][C], <= 18 KiB for 512 x 3 x 3) instead of five; block b belongs to the job with first_block <= b <
// first_block + K (binary search over the device table), one output channel per block.
__global__ __launch_bounds__(512) void prep_weights_bwd_parts_multi_kernel(const oadg_prep_bwd_job* __restrict__ jobs,
                                                                           int n_jobs) {
    extern __shared__ float gsum[];            // [R*S][C] of the widest job
    __shared__ float red[16];
    int lo = 0, hi = n_jobs - 1;
    const int b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= b) lo = mid; else hi = mid - 1;
    }
    const oadg_prep_bwd_job& d = jobs[__builtin_amdgcn_readfirstlane(lo)];
    const int k = b - d.first_block;
    const int K = d.K, C = d.C, RS = d.R * d.S, n = C * RS, splits = d.splits;
    const int w_krsc = d.w_krsc & 1;
    const size_t stride = (size_t)K * n;
    const float* p0 = d.part + (size_t)k * n;
    // (round 5: four elements per thread as one 16-byte load per split - the 4-byte form kept at most 8 x 256 bytes per
    //  wave in flight on a kernel whose workgroups live for a few microseconds: 1.85 TB/s; n = C * R * S is a multiple of
    //  64 and the partial tiles are 16-byte aligned.  Per element the same eight accumulators in the same order.)
    for (int j = threadIdx.x * 4; j < n; j += 2048) {
        f32x4v acc8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc8[u] = f32x4v{0.f, 0.f, 0.f, 0.f};
        int sp = 0;
        for (; sp + 8 <= splits; sp += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc8[u] += *reinterpret_cast<const f32x4v*>(p0 + (size_t)(sp + u) * stride + j);
        }
        for (int u = 0; sp < splits; ++sp, ++u) acc8[u] += *reinterpret_cast<const f32x4v*>(p0 + (size_t)sp * stride + j);
        *reinterpret_cast<f32x4v*>(gsum + j) =
            ((acc8[0] + acc8[1]) + (acc8[2] + acc8[3])) + ((acc8[4] + acc8[5]) + (acc8[6] + acc8[7]));
    }
    __syncthreads();
    const float sc = d.scale ? d.scale[k] : 1.f;
    const float* wk = d.w + (size_t)k * n;
    float* dw = d.dw + (size_t)k * n;
    float dot = 0.f;
    for (int i = threadIdx.x; i < n; i += 512) {
        const int rs = i % RS, c = i / RS;
        const float g = gsum[w_krsc ? i : rs * C + c];
        dw[i] = g * sc;
        dot += g * wk[i];
    }
    if (d.dgamma) {
        const float tot = block_sum(dot, red);
        if (threadIdx.x == 0)
            d.dgamma[k] = (d.w_krsc & 2) ? tot : (tot - (d.gbias ? d.gbias[k] : 0.f) * d.mean[k]) * rsqrtf(d.var[k] + d.eps);
    }
}
}  // namespace

// jobs_dev [n] on the DEVICE in ascending first_block order (first_block = sum of K over the jobs before); total_blocks =
// sum of K; max_crs = the largest C * R * S of the group (<= 36000: LDS).  Per job the arithmetic of
// oadg_prep_conv_weights_bwd_parts with the splits summed in one fixed order.
extern "C" int oadg_prep_conv_weights_bwd_parts_multi(const oadg_prep_bwd_job* jobs_dev, int n, int total_blocks,
                                                      int max_crs, void* stream) {
    if (!jobs_dev || n < 1 || total_blocks < 1 || max_crs < 1 || max_crs > 36000) return OADG_EARG;
    const size_t lds = (size_t)max_crs * sizeof(float);
    if (lds > 48 * 1024) {
        static bool attr = false;
        if (!attr) {
            hipError_t e = hipFuncSetAttribute((const void*)prep_weights_bwd_parts_multi_kernel,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 36000 * sizeof(float));
            if (e != hipSuccess) return (int)e;
            attr = true;
        }
    }
    hipLaunchKernelGGL(prep_weights_bwd_parts_multi_kernel, dim3((unsigned)total_blocks), dim3(512), lds,
                       (hipStream_t)stream, jobs_dev, n);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

extern "C" int oadg_prep_conv_weights_bwd(const void* gwf, const float* gbias, const float* w, const float* scale,
                                          const float* mean, const float* var, float eps, int K, int C, int R, int S,
                                          float* dw, float* dgamma, int w_krsc, void* stream) {
    if (!gwf || !w || !dw) return OADG_EARG;
    if (dgamma && (!mean || !var)) return OADG_EARG;
    hipLaunchKernelGGL(prep_weights_bwd_kernel, dim3(K), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)gwf, gbias, w, scale, mean, var, eps, K, C, R, S, dw, dgamma, w_krsc);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}
