// Shared device/host helpers for the OA-DG gfx950 kernels.
// Everything here is CDNA4-only (wave64, 256 CUs, 8 XCDs); no other target is supported.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define OADG_OK 0
#define OADG_EARG (-1)     // bad argument (null pointer, negative size, unsupported shape)
#define OADG_ESIZE (-2)    // workspace too small
#define OADG_EIO (-3)      // host file could not be read (png_decode.hip)
#define OADG_EUNSUPPORTED (-4)   // a file format variant the native decoder does not cover (the caller falls back)

#define OADG_WAVE 64

static inline int oadg_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Return code convention of the C ABI: 0 ok, <0 argument error, >0 hipError_t.
#define OADG_LAUNCH_CHECK()                         \
    do {                                            \
        hipError_t e__ = hipGetLastError();         \
        if (e__ != hipSuccess) return (int)e__;     \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block-wide sum for blocks of up to 1024 threads; result valid in every thread.
// `red` must hold >= 16 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < nw; ++i) s += red[i];
    return s;
}
__device__ __forceinline__ double block_sum_d(double v, double* red) {
    v = wave_sum_d(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < nw; ++i) s += red[i];
    return s;
}

__device__ __forceinline__ float bf16_to_f32(unsigned short h) {
    return __uint_as_float(((unsigned)h) << 16);
}
// round-to-nearest-even f32 -> bf16 (NaN -> quiet NaN) on gfx950's v_cvt_pk_bf16_f32: neighbouring conversions are
// paired by the compiler, 1/2 instruction per element where the integer rounding sequence took 6 - the epilogues of
// the HBM-bound convolution launches were VALU-bound on it (tools/probe/conv1x1_lab.hip)
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
    return __builtin_bit_cast(unsigned short, (__bf16)f);
}
