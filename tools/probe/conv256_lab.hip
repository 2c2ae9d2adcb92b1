// Experimental bench of the 256-tile convolution kernel (not part of the product): ablation modes of a copy of
// conv_igemm256_kernel, timed on the FPN / RPN 3x3 P2 shape with random operands.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I oa-dg_amd/csrc tools/probe/conv256_lab.hip -o tools/probe/conv256_lab_bin
#include <string.h>
#include "../../oa-dg_amd/csrc/conv_mfma.hip"
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>

namespace {
static unsigned short lab_f2b(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
template <bool POST, int MODE>
__global__ __launch_bounds__(512) void lab256_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int n_tiles = a.K / TN;
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
    const long m0 = (MODE & 32) ? (mt % 32) * TM : mt * TM;
    const int k0 = nt * TN;
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
            wb[h * 2 + i] = a.w + (size_t)(k0 + h * 128 + row) * a.R * a.S * a.C + lslot * 8;
        }

    auto advance = [&](TapState& st) {
        st.t++;
        st.c0 += BK;
        if (st.c0 == a.C) {
            st.c0 = 0;
            st.rs++;
            if (++st.s == a.S) { st.s = 0; st.r++; }
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

    f32x4v acc[2][2][2][4];        // [w half][w tile][p half][p tile]
#pragma unroll
    for (int x0 = 0; x0 < 2; ++x0)
#pragma unroll
        for (int x1 = 0; x1 < 2; ++x1)
#pragma unroll
            for (int x2 = 0; x2 < 2; ++x2)
#pragma unroll
                for (int x3 = 0; x3 < 4; ++x3) acc[x0][x1][x2][x3] = f32x4v{0.f, 0.f, 0.f, 0.f};

    bf16x8 pf[4][2], wf[2][2];
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
            for (int ks = 0; ks < 2; ++ks) wf[jt][ks] = *reinterpret_cast<const bf16x8*>(base + woff[jt][ks]);
    };
#define OADG_QUADRANT(WH, PH_)                                                                               \
    do {                                                                                                     \
        asm volatile("s_barrier" ::: "memory");                                                              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        __builtin_amdgcn_s_setprio(1);                                                                       \
        if (!(MODE & 4))                                                                                     \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                     \
            _Pragma("unroll") for (int jt = 0; jt < 2; ++jt)                                                 \
                _Pragma("unroll") for (int it = 0; it < 4; ++it)                                             \
                    acc[WH][jt][PH_][it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[jt][ks], pf[it][ks],   \
                                                                                   acc[WH][jt][PH_][it], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        asm volatile("s_barrier" ::: "memory");                                                              \
    } while (0)

    // ---- prologue: tile 0 complete, plus the two half-tiles of tile 1 that phases 3/4 of "tile -1" would stage
    TapState s1{0, 0, 0, 0, 0};
    stage_pix(0, s1, 0);
    stage_wgt(0, s1, 0);
    stage_wgt(1, s1, 0);
    stage_pix(1, s1, 0);
    advance(s1);                     // s1 = tile 1
    stage_pix(0, s1, 1);
    stage_wgt(1, s1, 1);
    TapState s2 = s1;
    advance(s2);                     // s2 = tile 2
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    if (!(MODE & 16) && wr == 1) asm volatile("s_barrier" ::: "memory");     // stagger: group 1 runs one barrier behind
    if (MODE & 2) { read_wgt(0, 0); read_pix(0, 0); }

    for (int t = 0; t < nk; ++t) {
        const int buf = t & 1;
        // phase 1
        if (!(MODE & 2)) { read_wgt(0, buf); read_pix(0, buf); }
        if (!(MODE & 1) && !(MODE & 64)) stage_pix(1, s1, buf ^ 1);
        OADG_QUADRANT(0, 0);
        // phase 2
        if (!(MODE & 2)) read_wgt(1, buf);
        if (!(MODE & 1) && !(MODE & 128)) stage_wgt(0, s1, buf ^ 1);
        OADG_QUADRANT(1, 0);
        // phase 3
        if (!(MODE & 2)) read_pix(1, buf);
        if (!(MODE & 1) && !(MODE & 64)) stage_pix(0, s2, buf);
        OADG_QUADRANT(1, 1);
        // phase 4
        if (!(MODE & 2)) read_wgt(0, buf);
        if (!(MODE & 1) && !(MODE & 128)) stage_wgt(1, s2, buf);
        if (!(MODE & 1)) { if (MODE & 192) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        OADG_QUADRANT(0, 1);
        s1 = s2;
        advance(s2);
    }
#undef OADG_QUADRANT
    if (!(MODE & 16) && wr == 0) asm volatile("s_barrier" ::: "memory");     // balance the stagger
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the zero-line stages of the tail have landed
    asm volatile("s_barrier" ::: "memory");

    // ---- epilogue: bf16 C image [256 pixels][256 channels] in LDS (16-byte slot ^ (pixel & 15): the 16 pixels of a
    // ds_write_b64 lane group land on 16 different bank groups), then 16-byte row-contiguous stores
    {
        const int cq = lane >> 4;
#pragma unroll
        for (int wh = 0; wh < 2; ++wh)
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
                        *reinterpret_cast<uint2*>(smem + p * 512 + ((((ch >> 3) ^ (p & 15))) << 4) + ((ch >> 2) & 1) * 8) = pk;
                    }
            }
    }
    // residual / mask pieces requested before the barrier, all 16 (x2) loads of the thread in flight together: the
    // 128 accumulator registers are dead once the C image is written
    constexpr int NPIECE = (TM * TN / 8) / 512;
    bf16x8 rv[POST ? NPIECE : 1], mv[POST ? NPIECE : 1];
    if (POST) {
#pragma unroll
        for (int it = 0; it < NPIECE; ++it) {
            const int q = it * 512 + tid;
            const long m = m0 + (q >> 5);
            const size_t off = (size_t)m * a.K + k0 + (q & 31) * 8;
            rv[it] = (a.res && m < a.M) ? *reinterpret_cast<const bf16x8*>(a.res + off) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            mv[it] = (a.mask && m < a.M) ? *reinterpret_cast<const bf16x8*>(a.mask + off) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    } else {
        rv[0] = mv[0] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    __syncthreads();
    float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < NPIECE; ++it) {
        const int q = it * 512 + tid;
        const int p = q >> 5, sg = q & 31;             // sg = tid & 31 for every piece of this thread
        const long m = m0 + p;
        if (m >= a.M) continue;
        bf16x8 v = *reinterpret_cast<const bf16x8*>(smem + p * 512 + ((sg ^ (p & 15)) << 4));
        const size_t off = (size_t)m * a.K + k0 + sg * 8;
        *reinterpret_cast<bf16x8*>(a.y + off) = finish_piece<POST>(a, v, rv[POST ? it : 0], mv[POST ? it : 0], csum);
    }
    if (a.colsum) {       // 16 threads share a channel slot: combine through the 16 KiB behind the C image
        float* red = reinterpret_cast<float*>(smem + 2 * BUF_BYTES);    // [16][256]
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(tid >> 5) * TN + (tid & 31) * 8 + e] = csum[e];
        __syncthreads();
        if (tid < TN) {
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) t += red[g * TN + tid];
            a.colsum[(size_t)mt * a.K + k0 + tid] = t;
        }
    }
}


template <bool POST, int MODE>
__global__ __launch_bounds__(512) void lab256b_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int n_tiles = a.K / TN;
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
    const long m0 = (MODE & 32) ? (mt % 32) * TM : mt * TM;
    const int k0 = nt * TN;
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
            wb[h * 2 + i] = a.w + (size_t)(k0 + h * 128 + row) * a.R * a.S * a.C + lslot * 8;
        }

    auto advance = [&](TapState& st) {
        st.t++;
        st.c0 += BK;
        if (st.c0 == a.C) {
            st.c0 = 0;
            st.rs++;
            if (++st.s == a.S) { st.s = 0; st.r++; }
        }
    };
    auto stage_pix = [&](int h, const TapState& st, int buf) {
        unsigned char* dst = smem + buf * BUF_BYTES + h * HALF_BYTES + wave * 1024;
        const bool live = st.t < nk;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (MODE & 512) { glds16(pb[h * 2 + i] + st.c0, dst + i * 8192); continue; }
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

    f32x4v acc[2][2][2][4];        // [w half][w tile][p half][p tile]
#pragma unroll
    for (int x0 = 0; x0 < 2; ++x0)
#pragma unroll
        for (int x1 = 0; x1 < 2; ++x1)
#pragma unroll
            for (int x2 = 0; x2 < 2; ++x2)
#pragma unroll
                for (int x3 = 0; x3 < 4; ++x3) acc[x0][x1][x2][x3] = f32x4v{0.f, 0.f, 0.f, 0.f};

    bf16x8 pf[4][2], wf2[2][2][2];
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
            for (int ks = 0; ks < 2; ++ks) wf2[h][jt][ks] = *reinterpret_cast<const bf16x8*>(base + woff[jt][ks]);
    };
#define OADG_MFMA32(W_FIRST, PH_)                                                                            \
    do {                                                                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
        asm volatile("s_barrier" ::: "memory");                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        __builtin_amdgcn_s_setprio(1);                                                                       \
        if (!(MODE & 4))                                                                                     \
        _Pragma("unroll") for (int hh = 0; hh < 2; ++hh)                                                     \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                     \
            _Pragma("unroll") for (int jt = 0; jt < 2; ++jt)                                                 \
                _Pragma("unroll") for (int it = 0; it < 4; ++it)                                             \
                    acc[hh ^ W_FIRST][jt][PH_][it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                \
                        wf2[hh ^ W_FIRST][jt][ks], pf[it][ks], acc[hh ^ W_FIRST][jt][PH_][it], 0, 0, 0);     \
        __builtin_amdgcn_s_setprio(0);                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        asm volatile("s_barrier" ::: "memory");                                                              \
    } while (0)

    // ---- prologue: tile 0 complete in buffer 0, P0 + W0 of tile 1 in buffer 1
    TapState s1{0, 0, 0, 0, 0};
    stage_pix(0, s1, 0);
    stage_wgt(0, s1, 0);
    stage_wgt(1, s1, 0);
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
        // phase A: both weight halves + pixel half 0 -> 32 MFMAs; stages W1 + P1 of tile t+1
        if (!(MODE & 2)) { read_wgt(0, buf); read_wgt(1, buf); read_pix(0, buf); }
        if (!(MODE & 1)) { stage_wgt(1, s1, buf ^ 1); stage_pix(1, s1, buf ^ 1); }
        if ((MODE & 256) && !(MODE & 1)) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        OADG_MFMA32(0, 0);
        // phase B: pixel half 1 -> 32 MFMAs; stages P0 + W0 of tile t+2 (this buffer: read in phase A, retired before
        // its first barrier)
        if (!(MODE & 2)) read_pix(1, buf);
        if (!(MODE & 1)) { stage_pix(0, s2, buf); stage_wgt(0, s2, buf); }
        if (!(MODE & 1)) { if (MODE & 256) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        OADG_MFMA32(1, 1);
        s1 = s2;
        advance(s2);
    }
#undef OADG_MFMA32
    if (!(MODE & 16) && wr == 0) asm volatile("s_barrier" ::: "memory");     // balance the stagger
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the zero-line stages of the tail have landed
    asm volatile("s_barrier" ::: "memory");

    // ---- epilogue: bf16 C image [256 pixels][256 channels] in LDS (16-byte slot ^ (pixel & 15): the 16 pixels of a
    // ds_write_b64 lane group land on 16 different bank groups), then 16-byte row-contiguous stores
    {
        const int cq = lane >> 4;
#pragma unroll
        for (int wh = 0; wh < 2; ++wh)
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
                        *reinterpret_cast<uint2*>(smem + p * 512 + ((((ch >> 3) ^ (p & 15))) << 4) + ((ch >> 2) & 1) * 8) = pk;
                    }
            }
    }
    // residual / mask pieces requested before the barrier, all 16 (x2) loads of the thread in flight together: the
    // 128 accumulator registers are dead once the C image is written
    constexpr int NPIECE = (TM * TN / 8) / 512;
    bf16x8 rv[POST ? NPIECE : 1], mv[POST ? NPIECE : 1];
    if (POST) {
#pragma unroll
        for (int it = 0; it < NPIECE; ++it) {
            const int q = it * 512 + tid;
            const long m = m0 + (q >> 5);
            const size_t off = (size_t)m * a.K + k0 + (q & 31) * 8;
            rv[it] = (a.res && m < a.M) ? *reinterpret_cast<const bf16x8*>(a.res + off) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            mv[it] = (a.mask && m < a.M) ? *reinterpret_cast<const bf16x8*>(a.mask + off) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    } else {
        rv[0] = mv[0] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    __syncthreads();
    float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < NPIECE; ++it) {
        const int q = it * 512 + tid;
        const int p = q >> 5, sg = q & 31;             // sg = tid & 31 for every piece of this thread
        const long m = m0 + p;
        if (m >= a.M) continue;
        bf16x8 v = *reinterpret_cast<const bf16x8*>(smem + p * 512 + ((sg ^ (p & 15)) << 4));
        const size_t off = (size_t)m * a.K + k0 + sg * 8;
        *reinterpret_cast<bf16x8*>(a.y + off) = finish_piece<POST>(a, v, rv[POST ? it : 0], mv[POST ? it : 0], csum);
    }
    if (a.colsum) {       // 16 threads share a channel slot: combine through the 16 KiB behind the C image
        float* red = reinterpret_cast<float*>(smem + 2 * BUF_BYTES);    // [16][256]
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(tid >> 5) * TN + (tid & 31) * 8 + e] = csum[e];
        __syncthreads();
        if (tid < TN) {
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) t += red[g * TN + tid];
            a.colsum[(size_t)mt * a.K + k0 + tid] = t;
        }
    }
}


template <bool POST, int MODE>
__global__ __launch_bounds__(512) void lab256c_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int n_tiles = a.K / TN;
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
    const long m0 = (MODE & 32) ? (mt % 32) * TM : mt * TM;
    const int k0 = nt * TN;
    const int nk = a.R * a.S * (a.C / BK);

    // ---- loader geometry: piece q = i*512 + tid of a half-tile -> row = q >> 3 (0..127), 16-byte slot q & 7.
    // Buffer-resource loads: a lane's source is a 32-bit byte offset into x (or w); taps outside the image and tiles
    // past the end use an offset beyond num_records, which the hardware turns into zeros - no zero line, no per-load
    // bounds arithmetic: one validity bit per filter tap, computed once.
    const unsigned x_bytes = (unsigned)((size_t)a.N * a.H * a.W * a.C * 2);
    const unsigned w_bytes = (unsigned)((size_t)a.K * a.R * a.S * a.C * 2);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, w_bytes, 0x00020000);
    constexpr unsigned OOB = 0xFFFFF000u;
    int poff0[4];                     // byte offset of (n, hi0, wi0, slot) - may be "negative" before the tap is added
    unsigned pmask[4];                // bit rs: tap (r, s) of this pixel lies inside the image
    int woff0[4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = i * 512 + tid;
            const int row = q >> 3, lslot = (q & 7) ^ ((row >> 1) & 7);
            const long m = m0 + h * 128 + row;
            unsigned mask = 0;
            int off = 0;
            if (m < a.M) {
                const unsigned mu = (unsigned)m;                 // M < 2^31 (checked on the host): 32-bit divisions
                const unsigned tq = mu / (unsigned)a.Wo;
                const int wo = (int)(mu - tq * (unsigned)a.Wo);
                const int n = (int)(tq / (unsigned)a.Ho);
                const int ho = (int)(tq - (unsigned)n * (unsigned)a.Ho);
                const int hi0 = ho * a.stride - a.pad, wi0 = wo * a.stride - a.pad;
                off = (int)((((unsigned)n * a.H + hi0) * a.W + wi0) * a.C + lslot * 8) * 2;
                for (int r = 0; r < a.R; ++r)
                    for (int s2 = 0; s2 < a.S; ++s2)
                        if ((unsigned)(hi0 + r * a.dil) < (unsigned)a.H && (unsigned)(wi0 + s2 * a.dil) < (unsigned)a.W)
                            mask |= 1u << (r * a.S + s2);
            }
            poff0[h * 2 + i] = off;
            pmask[h * 2 + i] = mask;
            woff0[h * 2 + i] = (int)(((unsigned)(k0 + h * 128 + row) * a.R * a.S * a.C + lslot * 8) * 2);
        }

    auto advance = [&](TapState& st) {
        st.t++;
        st.c0 += BK;
        if (st.c0 == a.C) {
            st.c0 = 0;
            st.rs++;
            if (++st.s == a.S) { st.s = 0; st.r++; }
        }
    };
    auto stage_pix = [&](int h, const TapState& st, int buf) {
        unsigned char* dst = smem + buf * BUF_BYTES + h * HALF_BYTES + wave * 1024;
        const unsigned bit = st.t < nk ? (1u << st.rs) : 0u;                       // wave-uniform
        const int tap = (((st.r * a.dil) * a.W + st.s * a.dil) * a.C + st.c0) * 2;   // wave-uniform
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned vo = (pmask[h * 2 + i] & bit) ? (unsigned)(poff0[h * 2 + i] + tap) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_ptr_t)(dst + i * 8192), 16, (int)vo, 0, 0, 0);
        }
    };
    auto stage_wgt = [&](int h, const TapState& st, int buf) {
        unsigned char* dst = smem + buf * BUF_BYTES + (2 + h) * HALF_BYTES + wave * 1024;
        const bool live = st.t < nk;
        const int tap = (st.rs * a.C + st.c0) * 2;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned vo = live ? (unsigned)(woff0[h * 2 + i] + tap) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (lds_ptr_t)(dst + i * 8192), 16, (int)vo, 0, 0, 0);
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

    f32x4v acc[2][2][2][4];        // [w half][w tile][p half][p tile]
#pragma unroll
    for (int x0 = 0; x0 < 2; ++x0)
#pragma unroll
        for (int x1 = 0; x1 < 2; ++x1)
#pragma unroll
            for (int x2 = 0; x2 < 2; ++x2)
#pragma unroll
                for (int x3 = 0; x3 < 4; ++x3) acc[x0][x1][x2][x3] = f32x4v{0.f, 0.f, 0.f, 0.f};

    bf16x8 pf[4][2], wf2[2][2][2];
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
            for (int ks = 0; ks < 2; ++ks) wf2[h][jt][ks] = *reinterpret_cast<const bf16x8*>(base + woff[jt][ks]);
    };
#define OADG_MFMA32(W_FIRST, PH_)                                                                            \
    do {                                                                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                   \
        asm volatile("s_barrier" ::: "memory");                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        __builtin_amdgcn_s_setprio(1);                                                                       \
        if (!(MODE & 4))                                                                                     \
        _Pragma("unroll") for (int hh = 0; hh < 2; ++hh)                                                     \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                     \
            _Pragma("unroll") for (int jt = 0; jt < 2; ++jt)                                                 \
                _Pragma("unroll") for (int it = 0; it < 4; ++it)                                             \
                    acc[hh ^ W_FIRST][jt][PH_][it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                \
                        wf2[hh ^ W_FIRST][jt][ks], pf[it][ks], acc[hh ^ W_FIRST][jt][PH_][it], 0, 0, 0);     \
        __builtin_amdgcn_s_setprio(0);                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        asm volatile("s_barrier" ::: "memory");                                                              \
    } while (0)

    // ---- prologue: tile 0 complete in buffer 0, P0 + W0 of tile 1 in buffer 1
    TapState s1{0, 0, 0, 0, 0};
    stage_pix(0, s1, 0);
    stage_wgt(0, s1, 0);
    stage_wgt(1, s1, 0);
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
        // phase A: both weight halves + pixel half 0 -> 32 MFMAs; stages W1 + P1 of tile t+1
        if (!(MODE & 2)) { read_wgt(0, buf); read_wgt(1, buf); read_pix(0, buf); }
        if (!(MODE & 1)) { stage_wgt(1, s1, buf ^ 1); stage_pix(1, s1, buf ^ 1); }
        if ((MODE & 256) && !(MODE & 1)) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        OADG_MFMA32(0, 0);
        // phase B: pixel half 1 -> 32 MFMAs; stages P0 + W0 of tile t+2 (this buffer: read in phase A, retired before
        // its first barrier)
        if (!(MODE & 2)) read_pix(1, buf);
        if (!(MODE & 1)) { stage_pix(0, s2, buf); stage_wgt(0, s2, buf); }
        if (!(MODE & 1)) { if (MODE & 256) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        OADG_MFMA32(1, 1);
        s1 = s2;
        advance(s2);
    }
#undef OADG_MFMA32
    if (!(MODE & 16) && wr == 0) asm volatile("s_barrier" ::: "memory");     // balance the stagger
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the zero-line stages of the tail have landed
    asm volatile("s_barrier" ::: "memory");

    // ---- epilogue: bf16 C image [256 pixels][256 channels] in LDS (16-byte slot ^ (pixel & 15): the 16 pixels of a
    // ds_write_b64 lane group land on 16 different bank groups), then 16-byte row-contiguous stores
    {
        const int cq = lane >> 4;
#pragma unroll
        for (int wh = 0; wh < 2; ++wh)
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
                        *reinterpret_cast<uint2*>(smem + p * 512 + ((((ch >> 3) ^ (p & 15))) << 4) + ((ch >> 2) & 1) * 8) = pk;
                    }
            }
    }
    // residual / mask pieces requested before the barrier, all 16 (x2) loads of the thread in flight together: the
    // 128 accumulator registers are dead once the C image is written
    constexpr int NPIECE = (TM * TN / 8) / 512;
    bf16x8 rv[POST ? NPIECE : 1], mv[POST ? NPIECE : 1];
    if (POST) {
#pragma unroll
        for (int it = 0; it < NPIECE; ++it) {
            const int q = it * 512 + tid;
            const long m = m0 + (q >> 5);
            const size_t off = (size_t)m * a.K + k0 + (q & 31) * 8;
            rv[it] = (a.res && m < a.M) ? *reinterpret_cast<const bf16x8*>(a.res + off) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            mv[it] = (a.mask && m < a.M) ? *reinterpret_cast<const bf16x8*>(a.mask + off) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    } else {
        rv[0] = mv[0] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
    __syncthreads();
    float csum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int it = 0; it < NPIECE; ++it) {
        const int q = it * 512 + tid;
        const int p = q >> 5, sg = q & 31;             // sg = tid & 31 for every piece of this thread
        const long m = m0 + p;
        if (m >= a.M) continue;
        bf16x8 v = *reinterpret_cast<const bf16x8*>(smem + p * 512 + ((sg ^ (p & 15)) << 4));
        const size_t off = (size_t)m * a.K + k0 + sg * 8;
        *reinterpret_cast<bf16x8*>(a.y + off) = finish_piece<POST>(a, v, rv[POST ? it : 0], mv[POST ? it : 0], csum);
    }
    if (a.colsum) {       // 16 threads share a channel slot: combine through the 16 KiB behind the C image
        float* red = reinterpret_cast<float*>(smem + 2 * BUF_BYTES);    // [16][256]
#pragma unroll
        for (int e = 0; e < 8; ++e) red[(tid >> 5) * TN + (tid & 31) * 8 + e] = csum[e];
        __syncthreads();
        if (tid < TN) {
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) t += red[g * TN + tid];
            a.colsum[(size_t)mt * a.K + k0 + tid] = t;
        }
    }
}


template <int MODE, int KERN = 0>
float run(const ConvArgs& a, int iters) {
    auto kern = KERN == 0 ? lab256_kernel<false, MODE> : (KERN == 1 ? lab256b_kernel<false, MODE> : lab256c_kernel<false, MODE>);
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF_BYTES + 16384);
    const long m_tiles = (a.M + TM - 1) / TM;
    const long blocks = ((m_tiles + 7) / 8) * 8 * (a.K / TN);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), 2 * BUF_BYTES + 16384, 0, a);
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i)
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), 2 * BUF_BYTES + 16384, 0, a);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / iters;
}
}  // namespace

int main(int argc, char** argv) {
    const int N = 8, H = 256, W = 512, C = 256, K = 256, R = 3;
    const size_t nx = (size_t)N * H * W * C, nw = (size_t)K * R * R * C, ny = (size_t)N * H * W * K;
    std::vector<unsigned short> hx(nx), hw(nw);
    srand(1);
    for (auto& v : hx) v = lab_f2b((rand() / (float)RAND_MAX) * 2.f - 1.f);
    for (auto& v : hw) v = lab_f2b(((rand() / (float)RAND_MAX) * 2.f - 1.f) / 48.f);
    unsigned short *x, *w, *y, *z;
    hipMalloc(&x, nx * 2); hipMalloc(&w, nw * 2); hipMalloc(&y, ny * 2); hipMalloc(&z, 256);
    hipMemcpy(x, hx.data(), nx * 2, hipMemcpyHostToDevice);
    hipMemcpy(w, hw.data(), nw * 2, hipMemcpyHostToDevice);
    hipMemset(z, 0, 256);
    ConvArgs a{};
    a.x = x; a.w = w; a.bias = nullptr; a.res = nullptr; a.y = y; a.zeros = z; a.mask = nullptr; a.colsum = nullptr;
    a.N = N; a.H = H; a.W = W; a.C = C; a.K = K; a.R = R; a.S = R; a.stride = 1; a.pad = 1; a.dil = 1; a.relu = 0;
    a.Ho = H; a.Wo = W; a.M = (long)N * H * W; a.scatter = 0; a.OH = H; a.OW = W; a.osh = a.osw = 1; a.oph = a.opw = 0;
    const double gf = 2.0 * a.M * K * C * R * R / 1e9;
    const int it = 20;
#define RUN(M_, what) { float ms = run<M_>(a, it); printf("mode %2d %-44s %7.3f ms %7.1f TF/s\n", M_, what, ms, gf / ms); }
    RUN(0, "baseline");
    RUN(1, "no global_load_lds in the loop");
    RUN(2, "no ds_read in the loop");
    RUN(4, "no MFMA");
    RUN(3, "no glds, no ds_read (MFMA + barriers)");
    RUN(5, "no glds, no MFMA (ds_read + barriers)");
    RUN(6, "no ds_read, no MFMA (glds + barriers)");
    RUN(7, "barriers only");
    RUN(70, "glds pixels only + barriers");
    RUN(134, "glds weights only + barriers");
    RUN(64, "all but pixel loads");
    RUN(128, "all but weight loads");
    RUN(38, "glds + barriers, 32 pixel tiles only (L2 hits)");
    RUN(32, "everything, 32 pixel tiles only (L2 hits)");
    RUN(0, "baseline again");
#define RUNB(M_, what) { float ms = run<M_, 1>(a, it); printf("B mode %2d %-42s %7.3f ms %7.1f TF/s\n", M_, what, ms, gf / ms); }
    RUNB(0, "32 MFMAs per phase");
    RUNB(3, "32/phase: MFMA + barriers");
    RUNB(6, "32/phase: glds + barriers");
    RUNB(256, "32/phase, waits vmcnt(8)/(6)");
    RUNB(512, "32/phase, trivial pixel addresses (wrong data)");
    RUNB(0, "32 MFMAs per phase again");
    RUNB(256, "32/phase, waits vmcnt(8)/(6) again");
#define RUNC(M_, what) { float ms = run<M_, 2>(a, it); printf("C mode %2d %-42s %7.3f ms %7.1f TF/s\n", M_, what, ms, gf / ms); }
    RUNC(0, "32/phase + buffer loads, tap masks");
    RUNB(0, "32 MFMAs per phase again");
    RUNC(0, "32/phase + buffer loads, tap masks again");
    {
        std::vector<unsigned short> y0(ny), y1(ny);
        run<0, 0>(a, 1); hipMemcpy(y0.data(), y, ny * 2, hipMemcpyDeviceToHost);
        hipMemset(y, 0, ny * 2);
        run<0, 2>(a, 1); hipMemcpy(y1.data(), y, ny * 2, hipMemcpyDeviceToHost);
        size_t bad = 0; for (size_t i = 0; i < ny; ++i) bad += y0[i] != y1[i];
        printf("C vs baseline: %zu of %zu outputs differ\n", bad, ny);
    }
    {   // correctness of B against the baseline kernel: identical accumulation order per output -> identical bytes
        std::vector<unsigned short> y0(ny), y1(ny);
        run<0, 0>(a, 1); hipMemcpy(y0.data(), y, ny * 2, hipMemcpyDeviceToHost);
        hipMemset(y, 0, ny * 2);
        run<256, 1>(a, 1); hipMemcpy(y1.data(), y, ny * 2, hipMemcpyDeviceToHost);
        size_t bad = 0; for (size_t i = 0; i < ny; ++i) bad += y0[i] != y1[i];
        printf("B vs baseline: %zu of %zu outputs differ\n", bad, ny);
    }
    return 0;
}
