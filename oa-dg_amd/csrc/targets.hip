// Sampler selection and anchor targets for a whole batch (integer / index work, bit-exact):
//   oadg_sample_select  - RandomSampler: locate the candidates with the ranks the host drew
//       (mmdet/core/bbox/samplers/random_sampler.py:32-82: candidates = nonzero(gt_inds > 0) resp. (== 0) in
//        ascending order, chosen = candidates[perm[:num]], then .unique() = sorted)
//   oadg_anchor_targets - AnchorHead._get_targets_single for all images (anchor_head.py:201-297 with
//        DeltaXYWHBBoxCoder.encode, delta_xywh_bbox_coder.py:119-180 incl. the fork's zero-size guard :152-160)
#include <cstring>
#include "common.h"
#include "../../include/oadg_hip.h"

namespace {

constexpr int SEL_CHUNK = 1024;        // candidates are counted per 1024-element chunk
constexpr int SEL_MAX_CHUNKS = 8192;   // 8.4 M boxes per job

__device__ __forceinline__ bool is_cand(long long v, int mode) { return mode == 0 ? v > 0 : v == 0; }

__global__ __launch_bounds__(256) void sel_count_kernel(const oadg_select_job* __restrict__ jobs, int* __restrict__ cnt,
                                                        int max_chunks) {
    const oadg_select_job jb = jobs[blockIdx.y];
    const long base = (long)blockIdx.x * SEL_CHUNK;
    if (base >= jb.n) return;
    const long long* v = (const long long*)jb.gt_inds;
    int c = 0;
    for (int i = threadIdx.x; i < SEL_CHUNK; i += 256) {
        const long n = base + i;
        c += (n < jb.n && is_cand(v[n], jb.mode)) ? 1 : 0;
    }
    __shared__ int red[4];
    c = wave_sum_i(c);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) cnt[(long)blockIdx.y * max_chunks + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// workgroup (x, job): exclusive scan of the job's chunk counts in LDS (every workgroup of the job repeats it: a few
// KiB), then its four waves locate ranks 16x .. 16x+15 of the job, one rank per wave at a time
constexpr int SEL_RANKS_PER_BLOCK = 16;

__global__ __launch_bounds__(256) void sel_locate_kernel(const oadg_select_job* __restrict__ jobs,
                                                         const int* __restrict__ cnt, const int* __restrict__ ranks,
                                                         long long* __restrict__ out, int max_chunks) {
    __shared__ int pre[SEL_MAX_CHUNKS + 1];
    __shared__ int wsum[4];
    const oadg_select_job jb = jobs[blockIdx.y];
    const int r0 = blockIdx.x * SEL_RANKS_PER_BLOCK;
    if (r0 >= jb.k) return;
    const int nch = (int)((jb.n + SEL_CHUNK - 1) / SEL_CHUNK);
    const int* c = cnt + (long)blockIdx.y * max_chunks;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // block-wide exclusive scan, 256 chunks per round
    int carry = 0;
    for (int base = 0; base < nch; base += 256) {
        const int i = base + threadIdx.x;
        const int v = i < nch ? c[i] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        if (i < nch) pre[i] = carry + woff + incl - v;
        carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) pre[nch] = carry;
    __syncthreads();
    const long long* v = (const long long*)jb.gt_inds;
    for (int i = r0 + wave; i < jb.k && i < r0 + SEL_RANKS_PER_BLOCK; i += 4) {
        const int r = jb.all ? i : ranks[jb.rank_off + i];
        if (r >= carry) continue;                      // cannot happen with consistent counts
        int lo = 0, hi = nch;                          // largest b with pre[b] <= r
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (pre[mid] <= r) lo = mid; else hi = mid;
        }
        int need = r - pre[lo];                        // rank inside the chunk
        const long base = (long)lo * SEL_CHUNK;
        for (int j = 0; j < SEL_CHUNK; j += 64) {
            const long n = base + j + lane;
            const bool m = n < jb.n && is_cand(v[n], jb.mode);
            const unsigned long long bal = __ballot(m);
            const int pc = __popcll(bal);
            if (need < pc) {
                const int before = __popcll(bal & ((1ull << lane) - 1ull));
                if (m && before == need) out[jb.out_off + i] = n;
                break;
            }
            need -= pc;
        }
    }
}

struct TargetArgs {
    const float* anchors;          // [A][4]
    const float* gts;              // [B][Gmax][4]
    const long long* gt_inds;      // [B][A]
    const long long* gt_labels;    // [B][Gmax] or null (RPN: positives get label 0)
    const oadg_select_job* jobs;   // [2B]: job 2b = positives of image b, 2b+1 = negatives
    const long long* sel;          // selected indices (oadg_sample_select output)
    long long* labels;             // [B][A]
    float* label_weights;          // [B][A]
    float* bbox_targets;           // [B][A][4]
    float* bbox_weights;           // [B][A][4]
    int B, A, Gmax;
    long long fill_label;
    float pos_weight;
    float mean[4], stdv[4];
    // device-side sample counts (oadg_roi_sample_device, csrc/roi_sampler.hip): target entry i holds `npos` = its CAPACITY
    // of rows; its first counts[(i % n_src) * 2] rows are positives, the next counts[.. + 1] negatives (their indices follow
    // the positives' in pos_inds), the rest - only when the image had fewer candidates than the capacity - is padding
    const int* counts;
    int n_src;
};

__global__ __launch_bounds__(256) void targets_fill_kernel(TargetArgs a) {
    const long total = (long)a.B * a.A;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        a.labels[i] = a.fill_label;
        a.label_weights[i] = 0.f;
        reinterpret_cast<float4*>(a.bbox_targets)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        reinterpret_cast<float4*>(a.bbox_weights)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// bbox2delta of one pair, operation for operation as the tensor expression in core/bbox.py.  The fork's guard for
// zero-size proposals pairs rows positionally (gy[nan_x] = py[nan_y], delta_xywh_bbox_coder.py:160); anchors from
// AnchorGenerator never have a zero side, so the row-wise form below (exact when a row is degenerate in both
// dimensions or in none) is what this kernel needs.
__device__ __forceinline__ float4 encode_delta(const float4 p, const float4 g, const float* mean, const float* stdv) {
    const float px = (p.x + p.z) * 0.5f, py = (p.y + p.w) * 0.5f;
    float pw = p.z - p.x, ph = p.w - p.y;
    float gx = (g.x + g.z) * 0.5f, gy = (g.y + g.w) * 0.5f;
    float gw = g.z - g.x, gh = g.w - g.y;
    const bool nx = pw == 0.f, ny = ph == 0.f;
    if (nx) { pw = 1e-6f; gw = 1e-6f; gx = px; }
    if (ny) { ph = 1e-6f; gh = 1e-6f; }
    if (nx && ny) gy = py;
    float4 d;
    d.x = ((gx - px) / pw - mean[0]) / stdv[0];
    d.y = ((gy - py) / ph - mean[1]) / stdv[1];
    d.z = (logf(gw / pw) - mean[2]) / stdv[2];
    d.w = (logf(gh / ph) - mean[3]) / stdv[3];
    return d;
}

__global__ __launch_bounds__(256) void targets_scatter_kernel(TargetArgs a) {
    const int job = blockIdx.y;
    const oadg_select_job jb = a.jobs[job];
    const int b = job >> 1;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < jb.k; i += gridDim.x * 256) {
        const long long n = a.sel[jb.out_off + i];
        const long o = (long)b * a.A + n;
        if ((job & 1) == 0) {
            const long long gi = a.gt_inds[o] - 1;
            const float4 p = reinterpret_cast<const float4*>(a.anchors)[n];
            const float4 g = reinterpret_cast<const float4*>(a.gts)[(long)b * a.Gmax + gi];
            reinterpret_cast<float4*>(a.bbox_targets)[o] = encode_delta(p, g, a.mean, a.stdv);
            reinterpret_cast<float4*>(a.bbox_weights)[o] = make_float4(1.f, 1.f, 1.f, 1.f);
            a.labels[o] = a.gt_labels ? a.gt_labels[(long)b * a.Gmax + gi] : 0;
            a.label_weights[o] = a.pos_weight <= 0.f ? 1.f : a.pos_weight;
        } else {
            a.label_weights[o] = 1.f;
        }
    }
}

// ---- RoI head: rois + targets of all sampled rows in one launch ---------------------------------------------------
struct RoiTargetArgs {
    oadg_roi_target_entry e[OADG_ROI_TARGET_MAX_ENTRIES];
    int row_off[OADG_ROI_TARGET_MAX_ENTRIES + 1];
    int n_entries, n_target;       // entries [n_target, n_entries) only produce rois
    int target_rows;               // row_off[n_target]
    int rows;                      // all rows
    float* rois;                   // [rows][5]
    long long* labels;             // [target_rows]
    float* label_weights;
    float* bbox_targets;           // [target_rows][4]
    float* bbox_weights;
    float* absolute;               // [target_rows][4] or null
    long long fill_label;
    float pos_weight;
    float mean[4], stdv[4];
    // device-side sample counts (oadg_roi_sample_device, csrc/roi_sampler.hip): target entry i holds `npos` = its CAPACITY
    // of rows; its first counts[(i % n_src) * 2] rows are positives, the next counts[.. + 1] negatives (their indices follow
    // the positives' in pos_inds), the rest - only when the image had fewer candidates than the capacity - is padding
    const int* counts;
    int n_src;
};

__global__ __launch_bounds__(256) void roi_targets_kernel(const RoiTargetArgs a) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= a.rows) return;
    // compile-time entry indices only: a dynamically indexed by-value argument block would be copied to scratch
    const float* bboxes = nullptr; const float* gtb = nullptr;
    const long long *gt_inds = nullptr, *lab = nullptr, *pos = nullptr, *neg = nullptr;
    int npos = 0, stride = 4, batch = 0, base = 0, ei = 0;
#pragma unroll
    for (int i = 0; i < OADG_ROI_TARGET_MAX_ENTRIES; ++i)
        if (i < a.n_entries && r >= a.row_off[i]) {
            bboxes = a.e[i].bboxes; gtb = a.e[i].gt_bboxes; gt_inds = (const long long*)a.e[i].gt_inds;
            lab = (const long long*)a.e[i].labels; pos = (const long long*)a.e[i].pos_inds;
            neg = (const long long*)a.e[i].neg_inds; npos = a.e[i].npos; stride = a.e[i].stride;
            batch = a.e[i].batch; base = a.row_off[i]; ei = i;
        }
    const int j = r - base;
    bool pad = false;
    if (a.counts && ei < a.n_target) {
        const int kp = a.counts[(ei % a.n_src) * 2], kn = a.counts[(ei % a.n_src) * 2 + 1];
        npos = kp;
        neg = pos + kp;
        pad = j >= kp + kn;         // (a short image: the trainer repeats the step on the host path; read nothing stale)
    }
    const bool is_pos = j < npos;
    const long long idx = pad ? 0 : (is_pos ? pos[j] : (neg ? neg[j - npos] : (long long)(j - npos)));
    const float* bp = bboxes + idx * stride;
    const float4 p = make_float4(bp[0], bp[1], bp[2], bp[3]);
    float* ro = a.rois + (long)r * 5;
    ro[0] = (float)batch; ro[1] = p.x; ro[2] = p.y; ro[3] = p.z; ro[4] = p.w;
    if (ei >= a.n_target) return;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f), w = t, ab = t;
    long long l = a.fill_label;
    float lw = 1.f;
    if (is_pos) {
        const long long gi = gt_inds[idx] - 1;
        const float4 g = reinterpret_cast<const float4*>(gtb)[gi];
        t = encode_delta(p, g, a.mean, a.stdv);
        w = make_float4(1.f, 1.f, 1.f, 1.f);
        ab = g;
        l = lab[idx];
        lw = a.pos_weight <= 0.f ? 1.f : a.pos_weight;
    }
    a.labels[r] = l;
    a.label_weights[r] = lw;
    reinterpret_cast<float4*>(a.bbox_targets)[r] = t;
    reinterpret_cast<float4*>(a.bbox_weights)[r] = w;
    if (a.absolute) reinterpret_cast<float4*>(a.absolute)[r] = ab;
}

}  // namespace

extern "C" size_t oadg_sample_select_workspace_bytes(int jobs, long max_n) {
    if (jobs < 1 || max_n < 0) return 0;
    const long ch = (max_n + SEL_CHUNK - 1) / SEL_CHUNK;
    return (size_t)jobs * (size_t)(ch > 0 ? ch : 1) * sizeof(int);
}

extern "C" int oadg_sample_select(const oadg_select_job* jobs_dev, int jobs, long max_n, int max_k,
                                  const int* ranks_dev, int64_t* out, void* workspace, size_t workspace_bytes,
                                  void* stream) {
    if (!jobs_dev || jobs < 1 || max_n < 0 || !out || !workspace) return OADG_EARG;
    const long ch = (max_n + SEL_CHUNK - 1) / SEL_CHUNK;
    if (ch > SEL_MAX_CHUNKS) return OADG_EARG;
    if (workspace_bytes < oadg_sample_select_workspace_bytes(jobs, max_n)) return OADG_ESIZE;
    if (max_n == 0 || max_k <= 0) return OADG_OK;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sel_count_kernel, dim3((unsigned)ch, jobs), dim3(256), 0, st, jobs_dev, (int*)workspace, (int)ch);
    OADG_LAUNCH_CHECK();
    hipLaunchKernelGGL(sel_locate_kernel, dim3((max_k + SEL_RANKS_PER_BLOCK - 1) / SEL_RANKS_PER_BLOCK, jobs), dim3(256), 0,
                       st, jobs_dev, (const int*)workspace, ranks_dev, (long long*)out, (int)ch);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

extern "C" int oadg_anchor_targets(const float* anchors, const float* gts, const int64_t* gt_inds,
                                   const int64_t* gt_labels, const oadg_select_job* jobs_dev, const int64_t* sel,
                                   int B, int A, int Gmax, int max_k, int64_t fill_label, float pos_weight,
                                   const float* means4, const float* stds4, int64_t* labels, float* label_weights,
                                   float* bbox_targets, float* bbox_weights, void* stream) {
    if (!anchors || !gt_inds || !jobs_dev || !sel || !labels || !label_weights || !bbox_targets || !bbox_weights ||
        !means4 || !stds4)
        return OADG_EARG;
    if (B < 1 || A < 1 || Gmax < 0 || max_k < 0 || (Gmax > 0 && !gts)) return OADG_EARG;
    TargetArgs a;
    a.anchors = anchors; a.gts = gts; a.gt_inds = (const long long*)gt_inds; a.gt_labels = (const long long*)gt_labels;
    a.jobs = jobs_dev; a.sel = (const long long*)sel; a.labels = (long long*)labels; a.label_weights = label_weights;
    a.bbox_targets = bbox_targets; a.bbox_weights = bbox_weights; a.B = B; a.A = A; a.Gmax = Gmax;
    a.fill_label = fill_label; a.pos_weight = pos_weight;
    for (int i = 0; i < 4; ++i) { a.mean[i] = means4[i]; a.stdv[i] = stds4[i]; }
    hipStream_t st = (hipStream_t)stream;
    const long total = (long)B * A;
    const int fb = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(targets_fill_kernel, dim3(fb), dim3(256), 0, st, a);
    OADG_LAUNCH_CHECK();
    if (max_k > 0) {
        hipLaunchKernelGGL(targets_scatter_kernel, dim3((max_k + 255) / 256, 2 * B), dim3(256), 0, st, a);
        OADG_LAUNCH_CHECK();
    }
    return OADG_OK;
}

namespace {
int roi_targets_launch(const oadg_roi_target_entry* entries_host, int n_entries, int n_target,
                       int64_t fill_label, float pos_weight, const float* means4, const float* stds4,
                       float* rois, int64_t* labels, float* label_weights, float* bbox_targets,
                       float* bbox_weights, float* absolute, const int* counts_dev, int n_src, void* stream) {
    if (!entries_host || n_entries < 1 || n_entries > OADG_ROI_TARGET_MAX_ENTRIES || n_target < 0 ||
        n_target > n_entries || !rois || !means4 || !stds4)
        return OADG_EARG;
    if (counts_dev && (n_src < 1 || n_src > n_target)) return OADG_EARG;
    if (n_target > 0 && (!labels || !label_weights || !bbox_targets || !bbox_weights)) return OADG_EARG;
    RoiTargetArgs a;
    std::memset(&a, 0, sizeof(a));
    long rows = 0;
    for (int i = 0; i < n_entries; ++i) {
        const oadg_roi_target_entry& e = entries_host[i];
        if (e.npos < 0 || e.nneg < 0 || e.stride < 4) return OADG_EARG;
        const bool raw = i >= n_target;
        if (e.npos + e.nneg > 0 && !e.bboxes) return OADG_EARG;
        if (raw && e.npos != 0) return OADG_EARG;
        if (!raw && ((e.npos > 0 && (!e.pos_inds || !e.gt_inds || !e.labels || !e.gt_bboxes)) ||
                     (e.nneg > 0 && !e.neg_inds)))
            return OADG_EARG;
        a.e[i] = e;
        a.row_off[i] = (int)rows;
        rows += (long)e.npos + e.nneg;
        if (rows > 0x7fffffffL) return OADG_ESIZE;
    }
    for (int i = n_entries; i <= OADG_ROI_TARGET_MAX_ENTRIES; ++i) a.row_off[i] = (int)rows;
    a.n_entries = n_entries; a.n_target = n_target; a.target_rows = a.row_off[n_target]; a.rows = (int)rows;
    a.rois = rois; a.labels = (long long*)labels; a.label_weights = label_weights; a.bbox_targets = bbox_targets;
    a.bbox_weights = bbox_weights; a.absolute = absolute; a.fill_label = fill_label; a.pos_weight = pos_weight;
    a.counts = counts_dev; a.n_src = counts_dev ? n_src : 1;
    for (int i = 0; i < 4; ++i) { a.mean[i] = means4[i]; a.stdv[i] = stds4[i]; }
    if (rows == 0) return OADG_OK;
    hipLaunchKernelGGL(roi_targets_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}
}  // namespace

extern "C" int oadg_roi_targets(const oadg_roi_target_entry* entries_host, int n_entries, int n_target,
                                int64_t fill_label, float pos_weight, const float* means4, const float* stds4,
                                float* rois, int64_t* labels, float* label_weights, float* bbox_targets,
                                float* bbox_weights, float* absolute, void* stream) {
    return roi_targets_launch(entries_host, n_entries, n_target, fill_label, pos_weight, means4, stds4, rois, labels,
                              label_weights, bbox_targets, bbox_weights, absolute, nullptr, 0, stream);
}

// the same with the positive / negative split of every sampled image read from DEVICE memory (counts_dev [n_src][2] of
// oadg_roi_sample_device; target entry i uses row i % n_src - the views of a batch share their sampling): entry.npos =
// the entry's row capacity (the sampler's `num`), entry.nneg = 0, entry.pos_inds = the image's block of `sel`
extern "C" int oadg_roi_targets_dev(const oadg_roi_target_entry* entries_host, int n_entries, int n_target, int n_src,
                                    const int* counts_dev, int64_t fill_label, float pos_weight, const float* means4,
                                    const float* stds4, float* rois, int64_t* labels, float* label_weights,
                                    float* bbox_targets, float* bbox_weights, float* absolute, void* stream) {
    if (!counts_dev) return OADG_EARG;
    return roi_targets_launch(entries_host, n_entries, n_target, fill_label, pos_weight, means4, stds4, rois, labels,
                              label_weights, bbox_targets, bbox_weights, absolute, counts_dev, n_src, stream);
}
