// Micro-victim 9 (see micro_victims.hip): C++ source, packed fp32 allowed, SLP on.  Weights come from LDS as float4, four reads
// in flight per iteration; the accumulation is the RoIAlign kernels': T[x] += w[x] * v (v one value per lane, w[x] from LDS),
// acc[y][x] += wy[y] * T[x].
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void victim9(const float* __restrict__ in, float* __restrict__ out, int iters) {
    __shared__ f32x4 tab[64][8];                       // a table per wave-quarter... indexed uniformly per wave below
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int wave = threadIdx.x >> 6;
    float v0 = in[i], v1 = in[i] * 0.5f + 0.25f;
    float acc[4][4];
    for (int y = 0; y < 4; ++y) for (int x = 0; x < 4; ++x) acc[y][x] = 0.f;
    for (int k = 0; k < iters; ++k) {
        if ((threadIdx.x & 63) < 8)                   // lanes 0-7 of each wave refresh that wave's 8 table rows
            tab[wave * 16 + (k & 1) * 8][threadIdx.x & 7] = f32x4{1.0f + k * 1e-6f, 0.5f, 0.25f + (threadIdx.x & 7) * 1e-3f, 0.125f};
        __syncthreads();
        const f32x4* row = tab[wave * 16 + (k & 1) * 8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 wx = row[j], wy = row[4 + j];   // uniform addresses: LDS broadcasts, as in the kernels
            const float c = j & 1 ? v1 : v0;
            float T[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) T[x] = wx[x] * c;
#pragma unroll
            for (int y = 0; y < 4; ++y)
#pragma unroll
                for (int x = 0; x < 4; ++x) acc[y][x] += wy[y] * T[x];
        }
        v0 = v0 * 0.9999f + 1e-4f; v1 = v1 * 0.9998f + 2e-4f;
        __syncthreads();
    }
    float s0 = 0.f, s1 = 0.f;
    for (int y = 0; y < 4; ++y) for (int x = 0; x < 4; ++x) { s0 += acc[y][x] * (1 + x); s1 += acc[y][x] * (1 + y); }
    out[2 * i] = s0; out[2 * i + 1] = s1;
}

extern "C" int launch_victim9(const float* in, float* out, int blocks, int iters, void* stream) {
    hipLaunchKernelGGL(victim9, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, iters);
    return (int)hipGetLastError();
}
