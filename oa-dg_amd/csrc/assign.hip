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
#include <cstring>
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
    long out_stride;           // elements between the images' rows of gt_inds / max_overlaps / labels (N when dense)
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
        a.gt_inds[(long)b * a.out_stride + n] = ind;
        a.max_overlaps[(long)b * a.out_stride + n] = mo;
        if (a.labels) a.labels[(long)b * a.out_stride + n] = ind > 0 ? a.gt_labels[(long)b * a.Gmax + (ind - 1)] : -1;
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

// ---- RoI head: assignment of the proposals + "add the gts as proposals" for the whole batch --------------------------
// Row r of image b in the [Gmax + N]-row outputs: r >= Gmax = proposal r - Gmax; r in [Gmax - G_b, Gmax) = gt
// r - (Gmax - G_b), self-matched (gt_inds j + 1, its label, overlap 1: AssignResult.add_gt_); rows below are unused, so
// image b's tensors are the contiguous row ranges [Gmax - G_b, Gmax + N).
struct RoiPrepArgs {
    oadg_roi_assign_image img[OADG_ROI_ASSIGN_MAX_IMAGES];
    int B, N, Gmax;
    float* boxes_full;         // [B][Gmax + N][4]
    int64_t* gt_inds_full;     // [B][Gmax + N]
    int64_t* labels_full;      // [B][Gmax + N]
    float* max_ov_full;        // [B][Gmax + N]
    unsigned char* valid;      // [B][N]
    float* gts_pad;            // [B][Gmax][4]
    int64_t* gl_pad;           // [B][Gmax]
    int* gt_counts;            // [B]
    unsigned* gt_max;          // [B][max(Gmax, 1)]
    int* counts;               // [B][2]
};

__global__ __launch_bounds__(256) void roi_assign_prep_kernel(const RoiPrepArgs a) {
    const int b = blockIdx.y;
    const float* props = nullptr; const float* gts = nullptr; const int64_t* gl = nullptr;
    int stride = 4, G = 0;
#pragma unroll
    for (int i = 0; i < OADG_ROI_ASSIGN_MAX_IMAGES; ++i)          // compile-time indices into the argument block
        if (i == b) { props = a.img[i].proposals; gts = a.img[i].gt_bboxes; gl = a.img[i].gt_labels;
                      stride = a.img[i].stride; G = a.img[i].num_gts; }
    const int rows = a.Gmax + a.N;
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r == 0) {
        a.gt_counts[b] = G;
        a.counts[2 * b] = 0; a.counts[2 * b + 1] = 0;
        if (a.Gmax == 0) a.gt_max[b] = fkey(-1.f);
    }
    if (r >= rows) return;
    const long o = (long)b * rows + r;
    if (r < a.Gmax) {
        // padded per-image gt tables for the assignment kernels (+ their per-gt maxima reset)
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        int64_t l = 0;
        if (r < G) { g = reinterpret_cast<const float4*>(gts)[r]; l = gl[r]; }
        reinterpret_cast<float4*>(a.gts_pad)[(long)b * a.Gmax + r] = g;
        a.gl_pad[(long)b * a.Gmax + r] = l;
        a.gt_max[(long)b * a.Gmax + r] = fkey(-1.f);
        const int j = r - (a.Gmax - G);
        if (j >= 0) {
            reinterpret_cast<float4*>(a.boxes_full)[o] = reinterpret_cast<const float4*>(gts)[j];
            a.gt_inds_full[o] = j + 1;
            a.labels_full[o] = gl[j];
            a.max_ov_full[o] = 1.0f;
        }
        return;
    }
    const int n = r - a.Gmax;
    const float* p = props + (long)n * stride;
    reinterpret_cast<float4*>(a.boxes_full)[o] = make_float4(p[0], p[1], p[2], p[3]);
    a.valid[(long)b * a.N + n] = stride >= 5 ? (p[4] >= 0.f ? 1 : 0) : 1;
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
    a.counts = counts; a.out_stride = N;
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

extern "C" int oadg_roi_assign_add_gt(const oadg_roi_assign_image* images_host, int B, int N, int Gmax, float pos_iou_thr,
                                      float neg_iou_lo, float neg_iou_hi, float min_pos_iou, int match_low_quality,
                                      float* boxes_full, int64_t* gt_inds_full, int64_t* labels_full, float* max_ov_full,
                                      unsigned char* valid, float* gts_pad, int64_t* gl_pad, int* gt_counts,
                                      void* workspace, size_t workspace_bytes, int* counts, void* stream) {
    if (!images_host || B < 1 || B > OADG_ROI_ASSIGN_MAX_IMAGES || N < 1 || Gmax < 0 || Gmax > AS_MAXG || !boxes_full ||
        !gt_inds_full || !labels_full || !max_ov_full || !valid || !gt_counts || !workspace || !counts ||
        (Gmax > 0 && (!gts_pad || !gl_pad)))
        return OADG_EARG;
    if (workspace_bytes < oadg_max_iou_assign_workspace_bytes(B, Gmax)) return OADG_ESIZE;
    RoiPrepArgs p;
    std::memset(&p, 0, sizeof(p));
    for (int i = 0; i < B; ++i) {
        const oadg_roi_assign_image& im = images_host[i];
        if (!im.proposals || im.stride < 4 || im.num_gts < 0 || im.num_gts > Gmax ||
            (im.num_gts > 0 && (!im.gt_bboxes || !im.gt_labels)))
            return OADG_EARG;
        p.img[i] = im;
    }
    p.B = B; p.N = N; p.Gmax = Gmax; p.boxes_full = boxes_full; p.gt_inds_full = gt_inds_full;
    p.labels_full = labels_full; p.max_ov_full = max_ov_full; p.valid = valid; p.gts_pad = gts_pad; p.gl_pad = gl_pad;
    p.gt_counts = gt_counts; p.gt_max = (unsigned*)workspace; p.counts = counts;
    hipStream_t st = (hipStream_t)stream;
    const int rows = Gmax + N;
    hipLaunchKernelGGL(roi_assign_prep_kernel, dim3((rows + 255) / 256, B), dim3(256), 0, st, p);
    OADG_LAUNCH_CHECK();
    AssignArgs a;
    a.boxes = boxes_full + (size_t)Gmax * 4; a.box_stride = (long)rows * 4; a.valid = valid; a.gts = gts_pad;
    a.gt_counts = gt_counts; a.gt_labels = gl_pad; a.B = B; a.N = N; a.Gmax = Gmax;
    a.pos_thr = pos_iou_thr; a.neg_lo = neg_iou_lo; a.neg_hi = neg_iou_hi; a.min_pos = min_pos_iou;
    a.match_low_quality = match_low_quality; a.gt_max = (unsigned*)workspace;
    a.gt_inds = gt_inds_full + Gmax; a.max_overlaps = max_ov_full + Gmax; a.labels = labels_full + Gmax;
    a.counts = counts; a.out_stride = rows;
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
