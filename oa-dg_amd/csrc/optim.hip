// SGD with momentum and weight decay for ALL parameters of the detector in one launch
//   serves torch.optim.SGD.step (the reference's optimizer: configs/_base_/schedules/schedule_1x.py
//   `optimizer = dict(type='SGD', lr=0.02, momentum=0.9, weight_decay=0.0001)`, stepped by mmcv's OptimizerHook
//   at mmdet/apis/train.py:150-161)
// torch's multi-tensor path runs four passes in ~13 launches over 166 MB of parameters (0.49 ms per step on the
// R50-FPN detector); one pass needs p, g, m read and p, m written once: 830 MB, HBM-bound.  The arithmetic is torch's,
// operation for operation (torch/optim/sgd.py _multi_tensor_sgd; ATen's `a + alpha * b` is one fused multiply-add):
//     g' = fma(wd, p, g)            (weight_decay != 0)
//     m  = g'                       (first step of a parameter)       m = m * momentum + g'   (later steps)
//     p  = fma(-lr, m, p)
#include "common.h"
#include "../../include/oadg_hip.h"

namespace {

constexpr int SGD_BLOCK_ELEMS = 4096;     // 256 threads x 4 floats x 4 rounds

__device__ __forceinline__ void sgd_one(float& p, const float g, float& m, float lr, float momentum, float wd, bool first) {
    const float gp = wd != 0.f ? __fmaf_rn(wd, p, g) : g;
    const float mn = first ? gp : (m * momentum + gp);
    m = mn;
    p = __fmaf_rn(-lr, mn, p);
}

__global__ __launch_bounds__(256) void sgd_multi_kernel(const oadg_sgd_tensor* __restrict__ tab, int n, float lr,
                                                        float momentum, float wd) {
    // the tensor this block works on: last entry whose first_block <= blockIdx.x
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (tab[mid].first_block <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const oadg_sgd_tensor t = tab[lo];
    const long long base = ((long long)blockIdx.x - t.first_block) * SGD_BLOCK_ELEMS;
    float* __restrict__ p = reinterpret_cast<float*>(t.param);
    const float* __restrict__ g = reinterpret_cast<const float*>(t.grad);
    float* __restrict__ m = reinterpret_cast<float*>(t.momentum);
    const bool first = t.first_step != 0;
    const bool vec = (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m) & 15) == 0;
    if (vec) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long long i = base + (long long)(r * 256 + threadIdx.x) * 4;
            if (i + 3 < t.numel) {
                float4 pv = *reinterpret_cast<float4*>(p + i);
                const float4 gv = *reinterpret_cast<const float4*>(g + i);
                float4 mv = first ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<float4*>(m + i);
                sgd_one(pv.x, gv.x, mv.x, lr, momentum, wd, first);
                sgd_one(pv.y, gv.y, mv.y, lr, momentum, wd, first);
                sgd_one(pv.z, gv.z, mv.z, lr, momentum, wd, first);
                sgd_one(pv.w, gv.w, mv.w, lr, momentum, wd, first);
                *reinterpret_cast<float4*>(p + i) = pv;
                *reinterpret_cast<float4*>(m + i) = mv;
            } else {
                for (long long j = i; j < t.numel && j < i + 4; ++j) {
                    float pv = p[j], mv = first ? 0.f : m[j];
                    sgd_one(pv, g[j], mv, lr, momentum, wd, first);
                    p[j] = pv; m[j] = mv;
                }
            }
        }
    } else {
        for (int r = 0; r < 16; ++r) {
            const long long j = base + r * 256 + threadIdx.x;
            if (j < t.numel) {
                float pv = p[j], mv = first ? 0.f : m[j];
                sgd_one(pv, g[j], mv, lr, momentum, wd, first);
                p[j] = pv; m[j] = mv;
            }
        }
    }
}

}  // namespace

extern "C" long long oadg_sgd_blocks(long long numel) {
    return numel <= 0 ? 0 : (numel + SGD_BLOCK_ELEMS - 1) / SGD_BLOCK_ELEMS;
}

extern "C" int oadg_sgd_step_multi(const oadg_sgd_tensor* table_dev, int n, long long total_blocks, float lr,
                                   float momentum, float weight_decay, void* stream) {
    if (!table_dev || n < 1 || total_blocks < 1 || total_blocks > 0x7fffffffLL) return OADG_EARG;
    hipLaunchKernelGGL(sgd_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, table_dev, n, lr,
                       momentum, weight_decay);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}
