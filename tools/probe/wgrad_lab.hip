// Lab of the 128-tile weight-gradient kernel (not part of the product): ablations of a copy of conv_wgrad_kernel
// (drop the dy / x staging, the MFMAs + transposing LDS reads, the partial-tile stores) on the backbone shapes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I oa-dg_amd/csrc tools/probe/wgrad_lab.hip -o tools/probe/wgrad_lab_bin
#include <string.h>
#include "../../oa-dg_amd/csrc/conv_mfma.hip"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

namespace {
static unsigned short lab_f2b(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }

// ABL: 1 = no dy staging, 2 = no x staging, 4 = no MFMA / LDS reads, 8 = no partial stores
template <int NST, int ABL>
__global__ __launch_bounds__(256, (NST == 1 ? 4 : 2)) void wgrad_abl(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int kt_n = a.K / 128, ct_n = a.C / 128, RS = a.R * a.S;
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
    int ln[4], lho[4], lwo[4], lslot[4], lrow[4];
    {
        const long pbase = ch0 * WP;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = i * 256 + tid;
            lrow[i] = q >> 4;
            lslot[i] = (q & 15) ^ ((lrow[i] & 3) << 2);
            const unsigned p = (unsigned)(pbase + lrow[i]);
            const unsigned t = p / (unsigned)a.Wo;
            lwo[i] = (int)(p - t * (unsigned)a.Wo);
            ln[i] = (int)(t / (unsigned)a.Ho);
            lho[i] = (int)(t - (unsigned)ln[i] * (unsigned)a.Ho);
        }
    }
    const int adv_h = WP / a.Wo, adv_w = WP - adv_h * a.Wo;
    auto stage = [&](long ch, int buf) {
        unsigned char* sa = smem + buf * WSTAGE;
        unsigned char* sb = sa + WP * 256;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long p = ch * WP + lrow[i];
            const unsigned short* sdy = a.zeros;
            const unsigned short* sx = a.zeros;
            if (p < a.P) {
                sdy = a.dy + (size_t)p * a.K + k0 + lslot[i] * 8;
                const int hi = lho[i] * a.stride - a.pad + r * a.dil, wi = lwo[i] * a.stride - a.pad + s * a.dil;
                if (hi >= 0 && hi < a.H && wi >= 0 && wi < a.W)
                    sx = a.x + (((size_t)ln[i] * a.H + hi) * a.W + wi) * a.C + c0 + lslot[i] * 8;
            }
            if (!(ABL & 1)) glds16(sdy, sa + i * 4096 + wave * 1024);
            if (!(ABL & 2)) glds16(sx, sb + i * 4096 + wave * 1024);
            lwo[i] += adv_w;
            lho[i] += adv_h;
            if (lwo[i] >= a.Wo) { lwo[i] -= a.Wo; ++lho[i]; }
            while (lho[i] >= a.Ho) { lho[i] -= a.Ho; ++ln[i]; }
        }
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
            if (!(ABL & 4)) {
#pragma unroll
                for (int t = 0; t < WP / 16; ++t) {
                    bf16x8 fa[2], fb[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) fa[i] = tr_frag(sa, t * 16, wm * 64 + i * 32, lane);
#pragma unroll
                    for (int j = 0; j < 2; ++j) fb[j] = tr_frag(sb, t * 16, wn * 64 + j * 32, lane);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
                }
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
                if (!(ABL & 8) || acc[i][j][e] == 12345.f) out[((size_t)k * RS + rs) * a.C + c] = acc[i][j][e];
            }
}

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
template <int NST, int ABL>
float run_abl(const WgradArgs& a, int it) {
    const long blocks = (long)a.splits * (a.K / 128) * (a.C / 128) * a.R * a.S;
    return time_it([&] { hipLaunchKernelGGL((wgrad_abl<NST, ABL>), dim3((unsigned)blocks), dim3(256), NST * WSTAGE, 0, a); }, it);
}
}  // namespace

int main(int argc, char** argv) {
    struct Shape { const char* name; int N, H, W, C, K, R; };
    const Shape shapes[] = {{"l2 conv3 128->512", 8, 128, 256, 128, 512, 1}, {"l2 conv1 512->128", 8, 128, 256, 512, 128, 1},
                            {"l3 conv3 256->1024", 8, 64, 128, 256, 1024, 1}, {"l3 conv1 1024->256", 8, 64, 128, 1024, 256, 1},
                            {"l4 conv3 512->2048", 8, 32, 64, 512, 2048, 1}, {"l4 conv1 2048->512", 8, 32, 64, 2048, 512, 1},
                            {"lateral P2 256->256", 8, 256, 512, 256, 256, 1}, {"lateral P3 512->256", 8, 128, 256, 512, 256, 1},
                            {"l2 conv2 3x3 128", 8, 128, 256, 128, 128, 3}, {"l3 conv2 3x3 256", 8, 64, 128, 256, 256, 3},
                            {"l4 conv2 3x3 512", 8, 32, 64, 512, 512, 3}, {"FPN P4 3x3 256", 8, 64, 128, 256, 256, 3},
                            {"FPN P5 3x3 256", 8, 32, 64, 256, 256, 3}, {"FPN P3 3x3 256", 8, 128, 256, 256, 256, 3},
                            {"FPN P2 3x3 256", 8, 256, 512, 256, 256, 3}, {"l2 down 256->512 (s1 form)", 8, 128, 256, 256, 512, 1}};
    for (const Shape& sh : shapes) {
        const int N = sh.N, H = sh.H, W = sh.W, C = sh.C, K = sh.K, R = sh.R;
        const size_t P = (size_t)N * H * W, nx = P * C, ny = P * K;
        std::vector<unsigned short> hx(nx), hy(ny);
        srand(1);
        for (auto& v : hx) v = lab_f2b((rand() / (float)RAND_MAX) * 2.f - 1.f);
        for (auto& v : hy) v = lab_f2b((rand() / (float)RAND_MAX) * 2.f - 1.f);
        unsigned short *x, *dy, *z;
        float* ws;
        hipMalloc(&x, nx * 2); hipMalloc(&dy, ny * 2); hipMalloc(&z, 256);
        hipMemcpy(x, hx.data(), nx * 2, hipMemcpyHostToDevice);
        hipMemcpy(dy, hy.data(), ny * 2, hipMemcpyHostToDevice);
        hipMemset(z, 0, 256);
        WgradArgs a{};
        a.x = x; a.dy = dy; a.zeros = z;
        a.N = N; a.H = H; a.W = W; a.C = C; a.K = K; a.R = R; a.S = R; a.stride = 1; a.pad = R / 2; a.dil = 1;
        a.Ho = H; a.Wo = W; a.P = (long)P;
        a.splits = wgrad_splits(a.P, K, C, R * R);
        const long nchunks = (a.P + WP - 1) / WP;
        a.chunks_per_split = (int)((nchunks + a.splits - 1) / a.splits);
        const size_t wsb = (size_t)a.splits * K * R * R * C * 4;
        hipMalloc(&ws, wsb);
        a.part = ws;
        const double gb = (2.0 * nx + 2.0 * ny + 2.0 * wsb) / 1e9, gf = 2.0 * P * K * C * R * R / 1e9;
        const int it = 20;
        float t[8];
        const float tp = time_it([&] { int sp; wgrad_launch(x, dy, nullptr, z, ws, wsb, N, H, W, C, K, R, R, 1, R / 2, 1, &sp, nullptr); }, it);
        if (R == 1) {
            t[0] = run_abl<2, 0>(a, it); t[1] = run_abl<2, 1>(a, it); t[2] = run_abl<2, 2>(a, it); t[3] = run_abl<2, 4>(a, it);
            t[4] = run_abl<2, 8>(a, it); t[5] = run_abl<2, 3>(a, it); t[6] = run_abl<2, 7>(a, it); t[7] = run_abl<1, 0>(a, it);
        } else {
            t[0] = run_abl<1, 0>(a, it); t[1] = run_abl<1, 1>(a, it); t[2] = run_abl<1, 2>(a, it); t[3] = run_abl<1, 4>(a, it);
            t[4] = run_abl<1, 8>(a, it); t[5] = run_abl<1, 3>(a, it); t[6] = run_abl<1, 7>(a, it); t[7] = run_abl<2, 0>(a, it);
        }
        if (K % 256 == 0 && C % 256 == 0) {
            hipFuncSetAttribute((const void*)conv_wgrad256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF_BYTES);
            const long tiles = (long)(K / 256) * (C / 256) * R * R;
            for (int target : {216, 256, 288, 512}) {
                WgradArgs b = a;
                long sp = target / tiles; if (sp < 1) sp = 1;
                if (sp > nchunks / 4) sp = nchunks / 4;
                b.splits = (int)sp;
                b.chunks_per_split = (int)((nchunks + b.splits - 1) / b.splits);
                const size_t wsb2 = (size_t)b.splits * K * R * R * C * 4;
                float* ws2; hipMalloc(&ws2, wsb2); b.part = ws2;
                long blocks = (long)b.splits * tiles;
                if (b.splits % 8 != 0) blocks = ((blocks + 7) / 8) * 8;
                const float t2 = time_it([&] { hipLaunchKernelGGL(conv_wgrad256_kernel, dim3((unsigned)blocks), dim3(512), 2 * BUF_BYTES, 0, b); }, it);
                printf("    256-tile: splits %3d (%ld workgroups, %d chunks each, ws %5.1f MB) %6.1f us %6.1f TF/s\n", b.splits, blocks,
                       b.chunks_per_split, wsb2 / 1e6, t2 * 1e3, gf / t2);
                hipFree(ws2);
            }
        }
        printf("%-22s splits %3d %6.1f MB (ws %5.1f) %6.1f GF | prod %6.1f us %5.2f TB/s %6.1f TF/s | copy %6.1f | -dy %6.1f | -x %6.1f | -mfma %6.1f | "
               "-store %6.1f | -dy-x %6.1f | -dy-x-mfma %6.1f | other NST %6.1f\n",
               sh.name, a.splits, gb * 1e3, 2.0 * wsb / 1e6, gf, tp * 1e3, gb / tp, gf / tp, t[0] * 1e3, t[1] * 1e3, t[2] * 1e3, t[3] * 1e3, t[4] * 1e3,
               t[5] * 1e3, t[6] * 1e3, t[7] * 1e3);
        hipFree(x); hipFree(dy); hipFree(z); hipFree(ws);
    }
    return 0;
}
