// Lab of the HBM-bound 1x1 convolutions (not part of the product): what bounds conv_igemm_kernel<128, POST, 1> on the
// bottleneck shapes?  Ablations of a copy of that kernel (drop the A / B staging, the MFMAs, the epilogue loads, the
// stores) beside a pure streaming kernel moving the same bytes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I oa-dg_amd/csrc tools/probe/conv1x1_lab.hip -o tools/probe/conv1x1_lab_bin
#include <string.h>
#include "../../oa-dg_amd/csrc/conv_mfma.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <type_traits>

namespace {
static unsigned short lab_f2b(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }

// ABL bits: 1 = no A staging after the first chunk's addresses (reads the zero line), 2 = no MFMA, 4 = no epilogue
// loads, 8 = no stores (one guarded store keeps the arithmetic alive), 16 = no B staging
template <bool POST, int ABL>
__global__ __launch_bounds__(256, 4) void k128_abl(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int TBN = 128, WN = 2, WM = 2, AF = 2, NBP = 4;
    constexpr int TSTAGE = (BM + TBN) * BK * 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int n_tiles = a.K / TBN;
    const long m_tiles = (a.M + BM - 1) / BM;
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
    const long m0 = mt * BM;
    const int k0 = nt * TBN;
    const int cpc = a.C / BK;
    const int nchunks = cpc;
    const unsigned short* a_base[4];
    int seg[4];
    const unsigned short* b_base[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = i * 256 + tid;
        const int row = q >> 3, slot = q & 7;
        seg[i] = slot ^ ((row >> 1) & 7);
        const long m = m0 + row;
        a_base[i] = m < a.M ? a.x + (size_t)m * a.C : nullptr;
        b_base[i] = a.w + (size_t)(k0 + row) * a.C;
    }
    auto stage = [&](int kc) {
        const int c0 = kc * BK;
        unsigned char* sa = smem;
        unsigned char* sb = sa + BM * BK * 2;
        if (!(ABL & 1)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned short* src = a_base[i] ? a_base[i] + c0 + seg[i] * 8 : a.zeros;
                glds16(src, sa + i * 4096 + wave * 1024);
            }
        }
        if (!(ABL & 16)) {
#pragma unroll
            for (int i = 0; i < NBP; ++i) glds16(b_base[i] + c0 + seg[i] * 8, sb + i * 4096 + wave * 1024);
        }
    };
    f32x16 acc[AF][2];
#pragma unroll
    for (int i = 0; i < AF; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int l31 = lane & 31, lh = lane >> 5;
    for (int kc = 0; kc < nchunks; ++kc) {
        const unsigned char* sa = smem;
        const unsigned char* sb = sa + BM * BK * 2;
        if (!(ABL & 2)) {
#pragma unroll
            for (int kk = 0; kk < BK / 16; ++kk) {
                const int sg = kk * 2 + lh;
                bf16x8 fa[AF], fb[2];
#pragma unroll
                for (int i = 0; i < AF; ++i) {
                    const int row = wm * (AF * 32) + i * 32 + l31;
                    fa[i] = *reinterpret_cast<const bf16x8*>(sa + row * 128 + ((sg ^ ((row >> 1) & 7)) << 4));
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int row = wn * 64 + j * 32 + l31;
                    fb[j] = *reinterpret_cast<const bf16x8*>(sb + row * 128 + ((sg ^ ((row >> 1) & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < AF; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
        if (kc + 1 < nchunks) stage(kc + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    unsigned short* tile = reinterpret_cast<unsigned short*>(smem);
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
    constexpr int SPR = TBN / 8;
    constexpr int NPIECE = (BM * TBN / 8) / 256;
    bf16x8 rv[POST ? NPIECE : 1], mv[POST ? NPIECE : 1];
    unsigned mb[POST ? NPIECE : 1];
    if (POST) {
#pragma unroll
        for (int it = 0; it < NPIECE; ++it) {
            const int q = it * 256 + tid;
            const long m = m0 + q / SPR;
            const size_t off = (size_t)(m < a.M ? m : 0) * a.K + k0 + (q % SPR) * 8;
            const bool ld = !(ABL & 4) && m < a.M;
            rv[it] = (a.res && ld) ? *reinterpret_cast<const bf16x8*>(a.res + off) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            mv[it] = (a.mask && ld) ? *reinterpret_cast<const bf16x8*>(a.mask + off) : bf16x8{1, 1, 1, 1, 1, 1, 1, 1};
            mb[it] = (a.bits_in && ld) ? a.bits_in[off >> 3] : 0xffu;
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
        const int row = q / SPR, sg = q % SPR;
        const long m = m0 + row;
        if (m >= a.M) continue;
        bf16x8 v = *reinterpret_cast<const bf16x8*>(tile + row * TBN + sg * 8);
        const size_t off = (size_t)m * a.K + k0 + sg * 8;
        v = finish_piece<POST>(a, v, rv[POST ? it : 0], mv[POST ? it : 0], csum, mb[POST ? it : 0], off);
        if (!(ABL & 8) || (unsigned short)v[0] == 0x7fc1u) *reinterpret_cast<bf16x8*>(a.y + off) = v;
    }
}

// the same bytes without the convolution: y[m][k] = relu(x[m][k % C] + res[m][k]) * bits
__global__ __launch_bounds__(256) void stream_kernel(ConvArgs a) {
    const long pieces = a.M * (a.K / 8);
    const int kp = a.K / 8, cp = a.C / 8;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < pieces; q += (long)gridDim.x * 256) {
        const long m = q / kp;
        const int j = (int)(q - m * kp);
        bf16x8 v = *reinterpret_cast<const bf16x8*>(a.x + (size_t)m * a.C + (size_t)(j % cp) * 8);
        bf16x8 r = a.res ? *reinterpret_cast<const bf16x8*>(a.res + (size_t)q * 8) : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        unsigned b = a.bits_in ? a.bits_in[q] : 0xffu;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float f = bf16_to_f32((unsigned short)v[e]) + bf16_to_f32((unsigned short)r[e]);
            if (!((b >> e) & 1u)) f = 0.f;
            v[e] = (short)f32_to_bf16(fmaxf(f, 0.f));
        }
        *reinterpret_cast<bf16x8*>(a.y + (size_t)q * 8) = v;
    }
}


// (the weights-stationary persistent kernel developed here is now csrc conv_pw_stream_kernel, variant 4)

template <typename F>
float time_it(F f, int it) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(e0, 0);
    for (int i = 0; i < it; ++i) f();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms / it;
}

template <bool POST, int ABL>
float run_abl(const ConvArgs& a, int it) {
    const long m_tiles = (a.M + BM - 1) / BM;
    const long blocks = ((m_tiles + 7) / 8) * 8 * (a.K / 128);
    return time_it([&] { hipLaunchKernelGGL((k128_abl<POST, ABL>), dim3((unsigned)blocks), dim3(256), (BM + 128) * BK * 2 + 8192, 0, a); }, it);
}
}  // namespace

int main(int argc, char** argv) {
    struct Shape { const char* name; int N, H, W, C, K, res, bits; };
    const Shape shapes[] = {{"l1 conv3 64->256 +res", 8, 256, 512, 64, 256, 1, 0},
                            {"l2 conv3 128->512 +res", 8, 128, 256, 128, 512, 1, 0},
                            {"l2 dgrad1 128->512 +res+bits", 8, 128, 256, 128, 512, 1, 1},
                            {"l3 conv3 256->1024 +res", 8, 64, 128, 256, 1024, 1, 0},
                            {"l3 dgrad1 256->1024 +res+bits", 8, 64, 128, 256, 1024, 1, 1},
                            {"l2 conv1 512->128", 8, 128, 256, 512, 128, 0, 0},
                            {"l2 dgrad3 512->128 +bits", 8, 128, 256, 512, 128, 0, 1},
                            {"lat 256->256 P2", 8, 256, 512, 256, 256, 0, 0},
                            {"l3 conv1 1024->256", 8, 64, 128, 1024, 256, 0, 0}, {"l3 dgrad3 1024->256 +bits", 8, 64, 128, 1024, 256, 0, 1},
                            {"l4 conv1 2048->512", 8, 32, 64, 2048, 512, 0, 0}, {"l3 down 512->1024 (s1 form)", 8, 64, 128, 512, 1024, 0, 0},
                            {"l4 conv3 512->2048 +res", 8, 32, 64, 512, 2048, 1, 0}, {"lat P3 512->256", 8, 128, 256, 512, 256, 0, 0}};
    for (const Shape& sh : shapes) {
        const int N = sh.N, H = sh.H, W = sh.W, C = sh.C, K = sh.K;
        const size_t M = (size_t)N * H * W, nx = M * C, nw = (size_t)K * C, ny = M * K;
        std::vector<unsigned short> hx(nx), hw(nw), hr(ny);
        srand(1);
        for (auto& v : hx) v = lab_f2b((rand() / (float)RAND_MAX) * 2.f - 1.f);
        for (auto& v : hw) v = lab_f2b(((rand() / (float)RAND_MAX) * 2.f - 1.f) / 16.f);
        for (size_t i = 0; i < ny; i += 7) hr[i] = lab_f2b((rand() / (float)RAND_MAX) * 2.f - 1.f);
        unsigned short *x, *w, *y, *z, *r;
        unsigned char* bits;
        hipMalloc(&x, nx * 2); hipMalloc(&w, nw * 2); hipMalloc(&y, ny * 2); hipMalloc(&z, 256); hipMalloc(&r, ny * 2);
        hipMalloc(&bits, ny / 8);
        hipMemcpy(x, hx.data(), nx * 2, hipMemcpyHostToDevice);
        hipMemcpy(w, hw.data(), nw * 2, hipMemcpyHostToDevice);
        hipMemcpy(r, hr.data(), ny * 2, hipMemcpyHostToDevice);
        hipMemset(z, 0, 256);
        hipMemset(bits, 0xb7, ny / 8);
        ConvArgs a{};
        a.x = x; a.w = w; a.y = y; a.zeros = z; a.res = sh.res ? r : nullptr; a.bits_in = sh.bits ? bits : nullptr;
        a.N = N; a.H = H; a.W = W; a.C = C; a.K = K; a.R = 1; a.S = 1; a.stride = 1; a.pad = 0; a.dil = 1; a.relu = sh.res && !sh.bits;
        a.Ho = H; a.Wo = W; a.M = (long)M; a.scatter = 0; a.OH = H; a.OW = W; a.osh = a.osw = 1; a.oph = a.opw = 0;
        const double gb = (2.0 * nx + 2.0 * nw + 2.0 * ny * (1 + sh.res) + (sh.bits ? ny / 8.0 : 0)) / 1e9;
        const int it = 20;
        const bool post = sh.res || sh.bits;
        float t[8];
        const float tp = time_it([&] { conv_launch(x, w, nullptr, a.res, y, z, N, H, W, C, K, 1, 1, 1, 0, 1, a.relu, 3, nullptr, nullptr, nullptr, nullptr, a.bits_in, nullptr); }, it);
        const float t256 = K % 256 == 0 ? time_it([&] { conv_launch(x, w, nullptr, a.res, y, z, N, H, W, C, K, 1, 1, 1, 0, 1, a.relu, 2, nullptr, nullptr, nullptr, nullptr, a.bits_in, nullptr); }, it) : 0.f;
        if (post) {
            t[0] = run_abl<true, 0>(a, it); t[1] = run_abl<true, 1>(a, it); t[2] = run_abl<true, 2>(a, it); t[3] = run_abl<true, 4>(a, it);
            t[4] = run_abl<true, 8>(a, it); t[5] = run_abl<true, 16>(a, it); t[6] = run_abl<true, 17>(a, it); t[7] = run_abl<true, 19>(a, it);
        } else {
            t[0] = run_abl<false, 0>(a, it); t[1] = run_abl<false, 1>(a, it); t[2] = run_abl<false, 2>(a, it); t[3] = run_abl<false, 4>(a, it);
            t[4] = run_abl<false, 8>(a, it); t[5] = run_abl<false, 16>(a, it); t[6] = run_abl<false, 17>(a, it); t[7] = run_abl<false, 19>(a, it);
        }
        float tw = 0.f; size_t bad = 0, worse = 0;
        if (pw_stream_ranges((long)M, C, K, 1, 1, 1, 0) > 0) {
            auto lw = [&] { conv_launch(x, w, nullptr, a.res, y, z, N, H, W, C, K, 1, 1, 1, 0, 1, a.relu, 4, nullptr, nullptr, nullptr, nullptr, a.bits_in, nullptr); };
            std::vector<unsigned short> y0(ny), y1(ny);
            conv_launch(x, w, nullptr, a.res, y, z, N, H, W, C, K, 1, 1, 1, 0, 1, a.relu, 3, nullptr, nullptr, nullptr, nullptr, a.bits_in, nullptr);
            hipMemcpy(y0.data(), y, ny * 2, hipMemcpyDeviceToHost);
            hipMemset(y, 0x55, ny * 2);
            lw();
            hipDeviceSynchronize();
            hipMemcpy(y1.data(), y, ny * 2, hipMemcpyDeviceToHost);
            for (size_t i = 0; i < ny; ++i) {
                if (y0[i] != y1[i]) { ++bad; int d = (int)y0[i] - (int)y1[i]; if (d < -1 || d > 1) ++worse; }
            }
            tw = time_it(lw, it);
            printf("    streaming kernel (variant 4) %6.1f us %5.2f TB/s  (%zu of %zu outputs differ from the 128-tile kernel, %zu by more than one bf16 step)\n",
                   tw * 1e3, gb / tw, bad, ny, worse);
        }
        const float ts = time_it([&] { hipLaunchKernelGGL(stream_kernel, dim3(256 * 8), dim3(256), 0, 0, a); }, it);
        printf("%-30s %6.1f MB | prod(v3) %6.1f us %5.2f TB/s | v2 %6.1f us | copy %6.1f us | -A %6.1f | -mfma %6.1f | -eload %6.1f | "
               "-store %6.1f | -B %6.1f | -A-B %6.1f | -A-B-mfma %6.1f | stream %6.1f us %5.2f TB/s\n",
               sh.name, gb * 1e3, tp * 1e3, gb / tp, t256 * 1e3, t[0] * 1e3, t[1] * 1e3, t[2] * 1e3, t[3] * 1e3, t[4] * 1e3, t[5] * 1e3,
               t[6] * 1e3, t[7] * 1e3, ts * 1e3, gb / ts);
        hipFree(x); hipFree(w); hipFree(y); hipFree(z); hipFree(r); hipFree(bits);
    }
    return 0;
}
