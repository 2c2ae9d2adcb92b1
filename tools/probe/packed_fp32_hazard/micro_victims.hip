// Micro-victims for the packed-fp32 hazard: each kernel runs ONE instruction pattern in a dependent loop on per-lane data and
// writes the result; the host compares launches bitwise while a co-tenant issues matrix instructions on another stream.
//   0  scalar reference      v_fma_f32 chains (no packed instruction)
//   1  pk, plain operands    v_pk_fma_f32 v[a:b], v[c:d], v[e:f], v[a:b]        (two independent lanes of data per instruction)
//   2  pk, broadcast operand v_pk_mul_f32 ..., w op_sel_hi:[1,0] + v_pk_add_f32  (one weight for both halves, as the RoIAlign kernel)
//   3  v_mov then pk         v_mov_b32 v40, w ; v_pk_mul_f32 ..., v[40:41] op_sel_hi:[1,0]   (the pair found in the failing ISA)
//   4  the same with s_nop 0 between the two
//   5 / 6  weight read back from LDS right before a packed / a scalar consumer;  7 / 8  the same through global memory
// (build with -fno-slp-vectorize: the vectorizer would pack kind 0 too)
#include <hip/hip_runtime.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void victim(int kind, const float* __restrict__ in, float* __restrict__ out, int iters) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    float a0 = in[i], a1 = in[i] * 0.5f + 1.0f, w = 1.0f + (threadIdx.x & 7) * 0.001f, b0 = 0.25f, b1 = -0.125f;
    if (kind == 0) {
        for (int k = 0; k < iters; ++k) {
            a0 = __builtin_fmaf(a0, w, b0); a1 = __builtin_fmaf(a1, w, b1);
            a0 = a0 * 0.999f; a1 = a1 * 0.999f;
        }
    } else if (kind == 1) {
        f32x2 a = {a0, a1}, ww = {w, w * 1.0001f}, bb = {b0, b1}, s = {0.999f, 0.999f};
        for (int k = 0; k < iters; ++k) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(ww), "v"(bb));
            asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a) : "v"(s));
        }
        a0 = a[0]; a1 = a[1];
    } else if (kind == 3 || kind == 4) {
        // the weight reaches the packed multiply through a 32-bit v_mov into the low half of the pair, issued IMMEDIATELY
        // before it (kind 3: the pair found in the failing ISA) or one wait state earlier (kind 4)
        f32x2 a = {a0, a1}, bb = {b0, b1}, s = {0.999f, 55.0f}, t;
        float w2 = w;
        for (int k = 0; k < iters; ++k) {
            if (kind == 3)
                asm volatile("v_mov_b32 v40, %2\n\tv_pk_mul_f32 %0, %1, v[40:41] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w2) : "v40", "v41");
            else
                asm volatile("v_mov_b32 v40, %2\n\ts_nop 0\n\tv_pk_mul_f32 %0, %1, v[40:41] op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w2) : "v40", "v41");
            asm volatile("s_nop 0\n\tv_pk_add_f32 %0, %1, %2" : "=v"(a) : "v"(t), "v"(bb));
            asm volatile("s_nop 0\n\tv_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(a) : "v"(s));
            w2 = w2 * 1.0000001f;                          // (a new value every iteration: a stale read is a different value)
        }
        a0 = a[0]; a1 = a[1];
    } else if (kind >= 5 && kind <= 8) {
        // the weight comes back from MEMORY right before its consumer: LDS (5: packed consumer, 6: scalar consumer) or global
        // memory (7: packed, 8: scalar); the thread writes a new weight to its own slot every iteration
        __shared__ float slot[256 * 2];
        float* gslot = out + 2 * (size_t)i;                 // (this thread's output slot doubles as its global scratch)
        f32x2 a = {a0, a1}, bb = {b0, b1}, t;
        float w2 = w;
        for (int k = 0; k < iters; ++k) {
            f32x2 wv;
            if (kind <= 6) {
                slot[2 * threadIdx.x] = w2; slot[2 * threadIdx.x + 1] = 77.0f;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(wv) : "v"((unsigned)(2 * threadIdx.x * 4)) : "memory");
            } else {
                gslot[0] = w2; gslot[1] = 77.0f;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(wv) : "v"(gslot) : "memory");
            }
            if (kind == 5 || kind == 7) {
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(wv));
                asm volatile("s_nop 0\n\tv_pk_add_f32 %0, %1, %2" : "=v"(a) : "v"(t), "v"(bb));
            } else {
                a[0] = a[0] * wv[0] + bb[0]; a[1] = a[1] * wv[0] + bb[1];
            }
            a[0] *= 0.999f; a[1] *= 0.999f;
            w2 = w2 * 1.0000001f;
        }
        a0 = a[0]; a1 = a[1];
    } else {
        f32x2 a = {a0, a1}, ww = {w, 123.0f}, bb = {b0, b1}, s = {0.999f, 55.0f}, t;
        for (int k = 0; k < iters; ++k) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(ww));
            asm volatile("s_nop 0\n\tv_pk_add_f32 %0, %1, %2" : "=v"(a) : "v"(t), "v"(bb));
            asm volatile("s_nop 0\n\tv_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(a) : "v"(s));
        }
        a0 = a[0]; a1 = a[1];
    }
    out[2 * i] = a0; out[2 * i + 1] = a1;
}

// 9: written in C++, compiled WITH packed fp32 and the SLP vectorizer (this file is otherwise built with -fno-slp-vectorize:
// see run.sh, which builds this kernel from micro_victim9.hip): weights read from LDS as float4, several reads in flight while
// packed multiplies / adds consume the ones that have arrived - the shape of the RoIAlign kernels' inner loops
extern "C" int launch_victim(int kind, const float* in, float* out, int blocks, int iters, void* stream) {
    hipLaunchKernelGGL(victim, dim3(blocks), dim3(256), 0, (hipStream_t)stream, kind, in, out, iters);
    return (int)hipGetLastError();
}
