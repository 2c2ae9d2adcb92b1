// MaxIoUAssigner for a whole batch in two launches (mmdet/core/bbox/assigners/max_iou_assigner.py:61-213 with
// BboxOverlaps2D, iou_calculators/iou2d_calculator.py:78-...), instead of ~25 tensor ops and a [G, N] IoU
// matrix per image.  Integer / compare work on fp32 IoUs: the IoU is evaluated with exactly the operation order
// of the reference's tensor expression (no contraction: the library is built with -ffp-contract=off), so the
// assignment is bit-identical to the tensor path.
//   pass 1: gt_max[b][g] = max over the valid boxes of IoU(gt g, box)         (ordered-int atomic max)
//   pass 2: per box: max / first argmax over the gts, the negative / positive thresholds, then low-quality
//           matching "every box that attains a gt's maximum (>= min_pos_iou) is assigned to it; later gts
//           overwrite earlier ones" (:195-201); optional labels; candidate counts for the sampler.
// The IoU of pass 2 is recomputed by the same code as pass 1, so `iou == gt_max` is an exact comparison.
#include "common.h"
#include "../../include/oadg_hip.h"

namespace {

constexpr int AS_THREADS = 256;
constexpr int AS_MAXG = 1024;          // gts per image kept in LDS (16 KiB)

struct AssignArgs {
    const float* boxes;        // [B or 1][N][4]
    long box_stride;           // floats between images (0: shared)
    const unsigned char* valid;  // [B][N] or null
    const float* gts;          // [B][Gmax][4]
    const int* gt_counts;      // [B]
    const int64_t* gt_labels;  // [B][Gmax] or null
    int B, N, Gmax;
    float pos_thr, neg_lo, neg_hi, min_pos;
    int match_low_quality;
    unsigned* gt_max;          // [B][Gmax] ordered-int keys (workspace)
    int64_t* gt_inds;          // [B][N]
    float* max_overlaps;       // [B][N]
    int64_t* labels;           // [B][N] or null
    int* counts;               // [B][2] (#gt_inds > 0, #gt_inds == 0), zeroed by the caller-side memset
};

__device__ __forceinline__ unsigned fkey(float f) {      // order-preserving float -> unsigned
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float funkey(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// bbox_overlaps(gt, box, mode='iou', eps=1e-6): same fp32 operation order as the tensor expression
__device__ __forceinline__ float iou_of(const float4 g, const float ga, const float4 b, const float ba) {
    const float ltx = fmaxf(g.x, b.x), lty = fmaxf(g.y, b.y);
    const float rbx = fminf(g.z, b.z), rby = fminf(g.w, b.w);
    const float w = fmaxf(rbx - ltx, 0.f), h = fmaxf(rby - lty, 0.f);
    const float overlap = w * h;
    float uni = (ga + ba) - overlap;
    uni = fmaxf(uni, 1e-6f);
    return overlap / uni;
}

__global__ __launch_bounds__(AS_THREADS) void assign_gtmax_kernel(AssignArgs a) {
    __shared__ float4 sg[AS_MAXG];
    __shared__ float sga[AS_MAXG];
    __shared__ unsigned smax[AS_MAXG];
    const int b = blockIdx.y;
    const int G = a.gt_counts[b];
    for (int g = threadIdx.x; g < G; g += AS_THREADS) {
        const float4 v = reinterpret_cast<const float4*>(a.gts)[(long)b * a.Gmax + g];
        sg[g] = v;
        sga[g] = (v.z - v.x) * (v.w - v.y);
        smax[g] = fkey(-1.f);
    }
    __syncthreads();
    const float* boxes = a.boxes + (long)b * a.box_stride;
    const int lane = threadIdx.x & 63;
    const long chunk = (long)gridDim.x * AS_THREADS;
    for (long base = (long)blockIdx.x * AS_THREADS; base < a.N; base += chunk) {
        const long n = base + threadIdx.x;
        const bool ok = n < a.N && (!a.valid || a.valid[(long)b * a.N + n]);
        float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < a.N) bx = reinterpret_cast<const float4*>(boxes)[n];
        const float ba = (bx.z - bx.x) * (bx.w - bx.y);
        for (int g = 0; g < G; ++g) {
            float v = ok ? iou_of(sg[g], sga[g], bx, ba) : -1.f;
            v = wave_max(v);
            if (lane == 0) atomicMax(&smax[g], fkey(v));
        }
    }
    __syncthreads();
    for (int g = threadIdx.x; g < G; g += AS_THREADS) atomicMax(&a.gt_max[(long)b * a.Gmax + g], smax[g]);
}

__global__ __launch_bounds__(AS_THREADS) void assign_kernel(AssignArgs a) {
    __shared__ float4 sg[AS_MAXG];
    __shared__ float sga[AS_MAXG];
    __shared__ float sgm[AS_MAXG];
    __shared__ int red[2][AS_THREADS / 64];
    const int b = blockIdx.y;
    const int G = a.gt_counts[b];
    for (int g = threadIdx.x; g < G; g += AS_THREADS) {
        const float4 v = reinterpret_cast<const float4*>(a.gts)[(long)b * a.Gmax + g];
        sg[g] = v;
        sga[g] = (v.z - v.x) * (v.w - v.y);
        sgm[g] = funkey(a.gt_max[(long)b * a.Gmax + g]);
    }
    __syncthreads();
    const float* boxes = a.boxes + (long)b * a.box_stride;
    int npos = 0, nneg = 0;
    const long chunk = (long)gridDim.x * AS_THREADS;
    for (long n = (long)blockIdx.x * AS_THREADS + threadIdx.x; n < a.N; n += chunk) {
        const bool ok = !a.valid || a.valid[(long)b * a.N + n];
        const float4 bx = reinterpret_cast<const float4*>(boxes)[n];
        const float ba = (bx.z - bx.x) * (bx.w - bx.y);
        int64_t ind = -1;
        float mo = 0.f;
        if (G == 0) {
            ind = 0;                                   // no gt: everything is background (:131-134)
        } else {
            float best = -2.f;
            int arg = 0, last = 0;
            for (int g = 0; g < G; ++g) {
                const float v = ok ? iou_of(sg[g], sga[g], bx, ba) : -1.f;
                if (v > best) { best = v; arg = g; }
                if (a.match_low_quality && sgm[g] >= a.min_pos && v == sgm[g]) last = g + 1;
            }
            mo = best;
            if (best >= a.neg_lo && best < a.neg_hi) ind = 0;
            if (best >= a.pos_thr) ind = arg + 1;
            if (last > 0) ind = last;
        }
        if (!ok) ind = -1;                             // filtered-out box (anchor_inside_flags / padding row)
        a.gt_inds[(long)b * a.N + n] = ind;
        a.max_overlaps[(long)b * a.N + n] = mo;
        if (a.labels) a.labels[(long)b * a.N + n] = ind > 0 ? a.gt_labels[(long)b * a.Gmax + (ind - 1)] : -1;
        npos += ind > 0;
        nneg += ind == 0;
    }
    npos = wave_sum_i(npos);
    nneg = wave_sum_i(nneg);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { red[0][w] = npos; red[1][w] = nneg; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int p = 0, q = 0;
        for (int i = 0; i < AS_THREADS / 64; ++i) { p += red[0][i]; q += red[1][i]; }
        if (p) atomicAdd(&a.counts[b * 2], p);
        if (q) atomicAdd(&a.counts[b * 2 + 1], q);
    }
}

__global__ void assign_init_kernel(unsigned* gt_max, int n, int* counts, int m) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) gt_max[i] = fkey(-1.f);
    if (i < m) counts[i] = 0;
}

}  // namespace

extern "C" size_t oadg_max_iou_assign_workspace_bytes(int B, int Gmax) {
    return B > 0 && Gmax >= 0 ? (size_t)B * (Gmax > 0 ? Gmax : 1) * sizeof(unsigned) : 0;
}

extern "C" int oadg_max_iou_assign(const float* boxes, long box_stride, const unsigned char* valid, const float* gts,
                                   const int* gt_counts, const int64_t* gt_labels, int B, int N, int Gmax,
                                   float pos_iou_thr, float neg_iou_lo, float neg_iou_hi, float min_pos_iou,
                                   int match_low_quality, void* workspace, size_t workspace_bytes, int64_t* gt_inds,
                                   float* max_overlaps, int64_t* labels, int* counts, void* stream) {
    if (!boxes || !gt_counts || !gt_inds || !max_overlaps || !counts || !workspace) return OADG_EARG;
    if (B < 1 || N < 0 || Gmax < 0 || Gmax > AS_MAXG || (Gmax > 0 && !gts) || (labels && !gt_labels)) return OADG_EARG;
    if (workspace_bytes < oadg_max_iou_assign_workspace_bytes(B, Gmax)) return OADG_ESIZE;
    AssignArgs a;
    a.boxes = boxes; a.box_stride = box_stride; a.valid = valid; a.gts = gts; a.gt_counts = gt_counts;
    a.gt_labels = gt_labels; a.B = B; a.N = N; a.Gmax = Gmax;
    a.pos_thr = pos_iou_thr; a.neg_lo = neg_iou_lo; a.neg_hi = neg_iou_hi; a.min_pos = min_pos_iou;
    a.match_low_quality = match_low_quality;
    a.gt_max = (unsigned*)workspace; a.gt_inds = gt_inds; a.max_overlaps = max_overlaps; a.labels = labels;
    a.counts = counts;
    hipStream_t st = (hipStream_t)stream;
    const int ninit = B * (Gmax > 0 ? Gmax : 1) > 2 * B ? B * (Gmax > 0 ? Gmax : 1) : 2 * B;
    hipLaunchKernelGGL(assign_init_kernel, dim3((ninit + 255) / 256), dim3(256), 0, st, a.gt_max,
                       B * (Gmax > 0 ? Gmax : 1), counts, 2 * B);
    OADG_LAUNCH_CHECK();
    if (N == 0) return OADG_OK;
    int bx = (N + AS_THREADS - 1) / AS_THREADS;
    if (bx > 256) bx = 256;
    if (Gmax > 0) {
        hipLaunchKernelGGL(assign_gtmax_kernel, dim3(bx, B), dim3(AS_THREADS), 0, st, a);
        OADG_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(assign_kernel, dim3(bx, B), dim3(AS_THREADS), 0, st, a);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}
