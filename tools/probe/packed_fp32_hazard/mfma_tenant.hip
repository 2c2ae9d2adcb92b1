// co-tenant micro-kernels: nothing but matrix instructions (or plain VALU) in a loop, one wave per SIMD x `waves`
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ __launch_bounds__(256) void k_mfma32(float* out, int iters) {
    bf16x8 a, b; for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    f32x16 c = {0};
    for (int i = 0; i < iters; ++i) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    out[blockIdx.x * 256 + threadIdx.x] = c[0] + c[15];
}
__global__ __launch_bounds__(256) void k_mfma16(float* out, int iters) {
    bf16x8 a, b; for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
    f32x4 c = {0};
    for (int i = 0; i < iters; ++i) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    out[blockIdx.x * 256 + threadIdx.x] = c[0] + c[3];
}
__global__ __launch_bounds__(256) void k_valu(float* out, int iters) {
    float x = threadIdx.x * 0.001f, y = 1.0001f;
    for (int i = 0; i < iters; ++i) { x = x * y + 0.5f; y = y * 0.99999f + x * 1e-9f; }
    out[blockIdx.x * 256 + threadIdx.x] = x + y;
}
extern "C" int launch(int kind, float* out, int blocks, int iters, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (kind == 0) hipLaunchKernelGGL(k_mfma32, dim3(blocks), dim3(256), 0, st, out, iters);
    else if (kind == 1) hipLaunchKernelGGL(k_mfma16, dim3(blocks), dim3(256), 0, st, out, iters);
    else hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), 0, st, out, iters);
    return (int)hipGetLastError();
}
