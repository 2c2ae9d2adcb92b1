// RPN proposal generation after the per-level top-k, fused for gfx950 (three launches + the NMS pair instead of ~120
// element-wise / gather / sort launches):
//   mmdet/models/dense_heads/rpn_head.py:103-235 (_get_bboxes_single + _bbox_post_process, batched over the images)
//   mmdet/core/bbox/coder/delta_xywh_bbox_coder.py:184-... (delta2bbox, same fp32 operation order: no fused multiply-adds)
//   mmcv.ops.batched_nms's class offset (boxes + level * (max coordinate + 1)) and its descending-score order
//
// oadg_rpn_decode  one thread per kept candidate (image, m): reads its 4 deltas straight from the RPN head's output
//                  (any strides / fp32 or bf16: the fused cls+reg head writes one 128-channel NHWC tensor) and its anchor
//                  through the top-k index, decodes, clips to the image, tests the minimum size.
// oadg_rpn_order   one workgroup per image.  The candidates of every level arrive sorted by descending score (stable),
//                  so the global stable descending order of key = valid ? score : -1 is a MERGE: the position of a valid
//                  candidate is the number of valid candidates that precede it in each level (a binary search in that
//                  level's sorted scores + a prefix count of valid flags), invalid ones follow in concatenation order -
//                  no sort.  Also the per-image maximum coordinate, the class(level)-offset boxes in that order for the
//                  NMS kernels (csrc/nms.hip) and the valid count.
// oadg_rpn_gather  the kept proposals as fixed-size lists [I, P, 5] (rows past the kept count: zero boxes, score -1).
// Integer / index work is exact; the float arithmetic repeats the tensor expressions of the torch path operation for
// operation (tests/test_hip_proposals.py compares both bit for bit).
#include "common.h"
#include "../../include/oadg_hip.h"

namespace {

struct RpnLevels {
    oadg_rpn_level l[OADG_RPN_MAX_LEVELS];
    int n;
};

__device__ __forceinline__ float load_delta(const void* p, long off, int dtype) {
    if (dtype == 0) return reinterpret_cast<const float*>(p)[off];
    const unsigned u = (unsigned)reinterpret_cast<const unsigned short*>(p)[off] << 16;
    return __builtin_bit_cast(float, u);
}

__global__ __launch_bounds__(256) void rpn_decode_kernel(RpnLevels lv, int n_img, int M, f32x4 means, f32x4 stds,
                                                         float max_ratio, const float* __restrict__ lim, int clip,
                                                         float min_size, float* __restrict__ props,
                                                         float* __restrict__ scores, unsigned char* __restrict__ valid) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    const int img = blockIdx.y;
    if (m >= M) return;
    int li = 0;
    while (li + 1 < lv.n && m >= lv.l[li + 1].first) ++li;
    const oadg_rpn_level L = lv.l[li];
    const int r = m - L.first;
    const long long i = L.index ? L.index[(size_t)img * L.k + r] : (long long)r;
    const int a = (int)(i % L.A);
    const long long pix = i / L.A;
    const int w = (int)(pix % L.W), h = (int)(pix / L.W);
    const long base = (long)img * L.sN + (long)h * L.sH + (long)w * L.sW;
    float d[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) d[j] = load_delta(L.deltas, base + (long)(a * 4 + j) * L.sC, L.dtype) * stds[j] + means[j];
    const float* an = L.anchors + (size_t)i * 4;
    const float ax1 = an[0], ay1 = an[1], ax2 = an[2], ay2 = an[3];
    const float px = (ax1 + ax2) * 0.5f, py = (ay1 + ay2) * 0.5f;
    const float pw = ax2 - ax1, ph = ay2 - ay1;
    const float dxw = pw * d[0], dyh = ph * d[1];
    const float dw = fminf(fmaxf(d[2], -max_ratio), max_ratio), dh = fminf(fmaxf(d[3], -max_ratio), max_ratio);
    const float gx = px + dxw, gy = py + dyh;
    const float gw = pw * expf(dw), gh = ph * expf(dh);
    float x1 = gx - gw * 0.5f, y1 = gy - gh * 0.5f, x2 = gx + gw * 0.5f, y2 = gy + gh * 0.5f;
    if (clip) {
        const float lw = lim[2 * img], lh = lim[2 * img + 1];
        x1 = fminf(fmaxf(x1, 0.f), lw); x2 = fminf(fmaxf(x2, 0.f), lw);
        y1 = fminf(fmaxf(y1, 0.f), lh); y2 = fminf(fmaxf(y2, 0.f), lh);
    }
    const size_t o = (size_t)img * M + m;
    reinterpret_cast<f32x4*>(props)[o] = f32x4{x1, y1, x2, y2};
    scores[o] = L.scores[(size_t)img * L.k + r];
    valid[o] = min_size >= 0.f ? (unsigned char)((x2 - x1 > min_size) && (y2 - y1 > min_size)) : (unsigned char)1;
}

// number of leading elements of the descending array s[0..n) that are >= v (ge) or > v
__device__ __forceinline__ int count_before(const float* __restrict__ s, int n, float v, bool ge) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const bool before = ge ? s[mid] >= v : s[mid] > v;
        if (before) lo = mid + 1; else hi = mid;
    }
    return lo;
}

constexpr int ORD_THREADS = 1024;

__global__ __launch_bounds__(ORD_THREADS) void rpn_order_kernel(RpnLevels lv, int M, const float* __restrict__ props,
                                                                const float* __restrict__ scores,
                                                                const unsigned char* __restrict__ valid,
                                                                float* __restrict__ boxes_sorted, int* __restrict__ order,
                                                                int* __restrict__ counts, float* __restrict__ mx_out) {
    extern __shared__ int G[];                 // exclusive prefix of the valid flags over the concatenation, [M + 1]
    float* SL = reinterpret_cast<float*>(G + M + 1);      // the image's scores (binary searches run on LDS)
    __shared__ float redf[ORD_THREADS / 64];
    __shared__ int redi[ORD_THREADS / 64 + 1];
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* P = props + (size_t)img * M * 4;
    const float* S = scores + (size_t)img * M;
    const unsigned char* V = valid + (size_t)img * M;
    // ---- maximum coordinate of the valid boxes (torch.where(valid, props, 0).amax()) and the prefix of the valid flags
    for (int m = tid; m < M; m += ORD_THREADS) SL[m] = S[m];
    const int per = (M + ORD_THREADS - 1) / ORD_THREADS;
    const int m0 = min(tid * per, M), m1 = min(m0 + per, M);
    float mx = 0.f;
    int cnt = 0;
    for (int m = m0; m < m1; ++m)
        if (V[m]) {
            const f32x4 b = reinterpret_cast<const f32x4*>(P)[m];
            mx = fmaxf(mx, fmaxf(fmaxf(b[0], b[1]), fmaxf(b[2], b[3])));
            ++cnt;
        }
    mx = wave_max(mx);
    int inc = cnt;                              // inclusive scan of the per-thread counts inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) redi[wave + 1] = inc;
    if (lane == 0) redf[wave] = mx;
    if (tid == 0) redi[0] = 0;
    __syncthreads();
    if (tid == 0)
        for (int w2 = 1; w2 <= ORD_THREADS / 64; ++w2) redi[w2] += redi[w2 - 1];
    __syncthreads();
    mx = redf[0];
    for (int w2 = 1; w2 < ORD_THREADS / 64; ++w2) mx = fmaxf(mx, redf[w2]);
    int run = redi[wave] + inc - cnt;           // valid candidates before this thread's chunk
    for (int m = m0; m < m1; ++m) {
        G[m] = run;
        run += V[m] ? 1 : 0;
    }
    const int total = redi[ORD_THREADS / 64];
    if (tid == 0) {
        G[M] = total;
        counts[img] = total;
        mx_out[img] = mx;
    }
    __syncthreads();
    // ---- merge position of every candidate
    const float step = mx + 1.0f;
    for (int m = tid; m < M; m += ORD_THREADS) {
        int li = 0;
        while (li + 1 < lv.n && m >= lv.l[li + 1].first) ++li;
        int pos;
        if (V[m]) {
            const float s = SL[m];
            pos = 0;
            for (int l2 = 0; l2 < lv.n; ++l2) {
                const int f = lv.l[l2].first, k = lv.l[l2].k;
                const int c = l2 == li ? m - f : count_before(SL + f, k, s, l2 < li);
                pos += G[f + c] - G[f];
            }
        } else {
            pos = total + (m - G[m]);
        }
        order[(size_t)img * M + pos] = m;
        const f32x4 b = reinterpret_cast<const f32x4*>(P)[m];
        const float off = (float)li * step;
        reinterpret_cast<f32x4*>(boxes_sorted)[(size_t)img * M + pos] = f32x4{b[0] + off, b[1] + off, b[2] + off, b[3] + off};
    }
}

__global__ __launch_bounds__(256) void rpn_gather_kernel(int M, int Pn, const float* __restrict__ props,
                                                         const float* __restrict__ scores, const int* __restrict__ order,
                                                         const int* __restrict__ keep, const int* __restrict__ keep_cnt,
                                                         float* __restrict__ dets) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x, img = blockIdx.y;
    if (j >= Pn) return;
    float o[5] = {0.f, 0.f, 0.f, 0.f, -1.0f};
    if (j < keep_cnt[img]) {
        int kk = keep[(size_t)img * M + j];
        kk = kk < 0 ? 0 : (kk >= M ? M - 1 : kk);
        const int m = order[(size_t)img * M + kk];
        const f32x4 b = reinterpret_cast<const f32x4*>(props)[(size_t)img * M + m];
        o[0] = b[0]; o[1] = b[1]; o[2] = b[2]; o[3] = b[3];
        o[4] = scores[(size_t)img * M + m];
    }
    float* d = dets + ((size_t)img * Pn + j) * 5;
#pragma unroll
    for (int c = 0; c < 5; ++c) d[c] = o[c];
}

int fill_levels(RpnLevels& lv, const oadg_rpn_level* levels, int n_levels, int* M) {
    if (!levels || n_levels < 1 || n_levels > OADG_RPN_MAX_LEVELS) return OADG_EARG;
    lv.n = n_levels;
    int first = 0;
    for (int i = 0; i < n_levels; ++i) {
        lv.l[i] = levels[i];
        if (levels[i].k < 0 || levels[i].first != first) return OADG_EARG;
        first += levels[i].k;
    }
    *M = first;
    return OADG_OK;
}

}  // namespace

extern "C" {

int oadg_rpn_decode(const oadg_rpn_level* levels, int n_levels, int n_img, const float* means4, const float* stds4,
                    float max_ratio, const float* lim, int clip, float min_size, float* props, float* scores,
                    unsigned char* valid, void* stream) {
    RpnLevels lv;
    int M = 0;
    const int rc = fill_levels(lv, levels, n_levels, &M);
    if (rc) return rc;
    if (n_img < 1 || !means4 || !stds4 || !props || !scores || !valid || (clip && !lim)) return OADG_EARG;
    for (int i = 0; i < n_levels; ++i)
        if (!levels[i].deltas || !levels[i].anchors || !levels[i].scores || levels[i].A < 1 || levels[i].W < 1 ||
            (levels[i].dtype != 0 && levels[i].dtype != 1))
            return OADG_EARG;
    if (M == 0) return OADG_OK;
    hipLaunchKernelGGL(rpn_decode_kernel, dim3((M + 255) / 256, n_img), dim3(256), 0, (hipStream_t)stream, lv, n_img, M,
                       f32x4{means4[0], means4[1], means4[2], means4[3]}, f32x4{stds4[0], stds4[1], stds4[2], stds4[3]},
                       max_ratio, lim, clip, min_size, props, scores, valid);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

int oadg_rpn_order(const oadg_rpn_level* levels, int n_levels, int n_img, const float* props, const float* scores,
                   const unsigned char* valid, float* boxes_sorted, int* order, int* counts, float* max_coord,
                   void* stream) {
    RpnLevels lv;
    int M = 0;
    const int rc = fill_levels(lv, levels, n_levels, &M);
    if (rc) return rc;
    if (n_img < 1 || !props || !scores || !valid || !boxes_sorted || !order || !counts || !max_coord) return OADG_EARG;
    const size_t lds = (size_t)(2 * M + 1) * sizeof(int);
    if (lds > 150 * 1024) return OADG_EARG;                       // ~19,000 candidates per image
    static size_t lds_set = 0;
    if (lds > 48 * 1024 && lds > lds_set) {
        hipError_t e = hipFuncSetAttribute((const void*)rpn_order_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) return (int)e;
        lds_set = 150 * 1024;
    }
    hipLaunchKernelGGL(rpn_order_kernel, dim3(n_img), dim3(ORD_THREADS), lds, (hipStream_t)stream, lv, M, props, scores,
                       valid, boxes_sorted, order, counts, max_coord);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

int oadg_rpn_gather(int n_img, int M, int P, const float* props, const float* scores, const int* order, const int* keep,
                    const int* keep_cnt, float* dets, void* stream) {
    if (n_img < 1 || M < 1 || P < 1 || !props || !scores || !order || !keep || !keep_cnt || !dets) return OADG_EARG;
    hipLaunchKernelGGL(rpn_gather_kernel, dim3((P + 255) / 256, n_img), dim3(256), 0, (hipStream_t)stream, M, P, props,
                       scores, order, keep, keep_cnt, dets);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

}  // extern "C"

// ================================================================================================ per-level top-k
// scores.sort(dim=1, descending=True, stable=True)[:, :nms_pre] of rpn_head.py:146-155 for every (image, level) row at
// once, without sorting the 393,216 scores of the finest level: a three-pass radix SELECT on the score bits (sigmoid
// outputs are non-negative floats: their bit patterns order like the values) finds the exact k-th largest key T, the
// candidates (all keys > T plus the first `need` keys == T in index order - exactly what a stable sort would keep) are
// compacted in index order by two chunked passes, and one workgroup per row sorts its k candidates in LDS by
// (key descending, index ascending).  Six launches for all rows (torch: ~12 rocPRIM launches per level + sigmoid / cast /
// permute / gather passes).
namespace {

struct SelRow {                 // one (image, level) row of scores
    long score_off;             // offset of the row in the scratch score array
    int n, k, level, img;       // elements, elements to keep
    int chunk0, nchunk;         // its chunks of SEL_C elements in the global chunk list
    int out_off;                // offset of its k outputs inside the level's [I, k] arrays = img * k
};
struct SelLevel {
    const void* cls;            // cls_score of the level, logical [N, A, H, W]
    long sN, sC, sH, sW;
    float* out_scores;          // [I, k]
    long long* out_index;       // [I, k]
    int H, W, A, dtype;
};
constexpr int SEL_MAX_ROWS = 48;
struct SelArgs {                 // passed by value (kernel arguments: no upload, no staging copy on the launch path)
    SelLevel lv[OADG_RPN_MAX_LEVELS];
    SelRow rows[SEL_MAX_ROWS];
    int cand_off[SEL_MAX_ROWS];
    int n_levels, n_rows, n_chunks;
};
constexpr int SEL_C = 4096;      // elements per chunk (256 threads x 16 consecutive elements)

__device__ __forceinline__ float rpn_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// Histograms are accumulated per workgroup in LDS and flushed bin by bin (non-zero bins only): sigmoid outputs of one level
// crowd into a handful of top-bit bins (8 bins per octave), and saturated scores are EQUAL - per-element global atomics on
// ten addresses serialise (measured: 2.7 ms for the 2.1 M scores of a step against ~20 us through LDS).
__device__ __forceinline__ void hist_zero(int* lh, int nbins) {
    for (int b = threadIdx.x; b < nbins; b += 256) lh[b] = 0;
    __syncthreads();
}
__device__ __forceinline__ void hist_flush(const int* lh, int* gh, int nbins) {
    __syncthreads();
    for (int b = threadIdx.x; b < nbins; b += 256) {
        const int c = lh[b];
        if (c) atomicAdd(gh + b, c);
    }
}

__device__ __forceinline__ int row_of_chunk(const SelArgs& a, int chunk) {
    int lo = 0, hi = a.n_rows - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.rows[mid].chunk0 <= chunk) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// pass 0: scores (sigmoid of the head output, in (h, w, a) order) to scratch + histogram of key bits [31:20]
__global__ __launch_bounds__(256) void sel_score_kernel(SelArgs a, float* __restrict__ scores,
                                                        int* __restrict__ hist) {
    __shared__ int lh[4096];
    const int r = row_of_chunk(a, blockIdx.x);
    const SelRow row = a.rows[r];
    const SelLevel L = a.lv[row.level];
    const int base = (blockIdx.x - row.chunk0) * SEL_C;
    const bool select = row.n > row.k;
    if (select) hist_zero(lh, 4096);
    for (int t = 0; t < SEL_C / 256; ++t) {
        const int i = base + t * 256 + threadIdx.x;
        const bool in = i < row.n;
        float s = 0.f;
        if (in) {
            const int an = i % L.A, pix = i / L.A;
            const int w = pix % L.W, h = pix / L.W;
            const long off = (long)row.img * L.sN + (long)an * L.sC + (long)h * L.sH + (long)w * L.sW;
            s = rpn_sigmoid(load_delta(L.cls, off, L.dtype));
            scores[row.score_off + i] = s;
        }
        if (select && in) atomicAdd(&lh[__builtin_bit_cast(unsigned, s) >> 20], 1);
    }
    if (select) hist_flush(lh, hist + (size_t)r * 4096, 4096);
}

// the bin (from the top) in which the cumulative count crosses `want`; returns the bin, `above` = elements in higher bins
__device__ int pick_bin(const int* __restrict__ hist, int nbins, int want, int* above_out, int* sh) {
    // 256 threads: each sums nbins / 256 consecutive bins (from the top), block scan, then the crossing thread refines
    const int per = nbins / 256, tid = threadIdx.x;
    const int hi = nbins - 1 - tid * per;
    int s = 0;
    for (int j = 0; j < per; ++j) s += hist[hi - j];
    sh[tid] = s;
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int t2 = 0; t2 < 256; ++t2) { const int v = sh[t2]; sh[t2] = acc; acc += v; }
    }
    __syncthreads();
    const int before = sh[tid];
    __syncthreads();
    if (before < want && before + s >= want) {
        int acc = before;
        for (int j = 0; j < per; ++j) {
            const int c = hist[hi - j];
            if (acc + c >= want) { sh[0] = hi - j; sh[1] = acc; break; }
            acc += c;
        }
    }
    __syncthreads();
    const int bin = sh[0];
    *above_out = sh[1];
    __syncthreads();
    return bin;
}

struct SelState { unsigned prefix; int above; };      // per row: key bits fixed so far, elements strictly above them

// pass 1 / 2: refine with the next bits.  LEVEL 1: pick from hist12a -> histogram bits [19:8]; LEVEL 2: pick from hist12b ->
// histogram bits [7:0]
template <int LEVEL>
__global__ __launch_bounds__(256) void sel_refine_kernel(SelArgs a,
                                                         const float* __restrict__ scores, const int* __restrict__ hist_in,
                                                         int* __restrict__ hist_out, SelState* __restrict__ state) {
    __shared__ int sh[256];
    const int r = row_of_chunk(a, blockIdx.x);
    const SelRow row = a.rows[r];
    if (row.n <= row.k) return;
    SelState st = LEVEL == 1 ? SelState{0u, 0} : state[(size_t)(LEVEL - 2) * a.n_rows + r];
    int above;
    const int nb = 4096;
    const int bin = pick_bin(hist_in + (size_t)r * nb, nb, row.k - st.above, &above, sh);
    const unsigned prefix = LEVEL == 1 ? ((unsigned)bin << 20) : (st.prefix | ((unsigned)bin << 8));
    const int tot_above = st.above + above;
    if (blockIdx.x == row.chunk0 && threadIdx.x == 0) state[(size_t)(LEVEL - 1) * a.n_rows + r] = SelState{prefix, tot_above};
    const int base = (blockIdx.x - row.chunk0) * SEL_C;
    const unsigned mask = LEVEL == 1 ? 0xFFF00000u : 0xFFFFFF00u;
    constexpr int NB = LEVEL == 1 ? 4096 : 256;
    __shared__ int lh[NB];
    hist_zero(lh, NB);
    for (int t = 0; t < SEL_C / 256; ++t) {
        const int i = base + t * 256 + threadIdx.x;
        const bool in = i < row.n;
        const unsigned key = in ? __builtin_bit_cast(unsigned, scores[row.score_off + i]) : 0u;
        const bool hit = in && (key & mask) == prefix;
        const int b = LEVEL == 1 ? (int)((key >> 8) & 0xFFFu) : (int)(key & 0xFFu);
        if (hit) atomicAdd(&lh[b], 1);
    }
    hist_flush(lh, hist_out + (size_t)r * NB, NB);
}

// pass 3: the exact threshold key T and how many keys == T are kept; per chunk the counts of keys > T and == T
__global__ __launch_bounds__(256) void sel_count_kernel2(SelArgs a,
                                                         const float* __restrict__ scores, const int* __restrict__ hist8,
                                                         SelState* __restrict__ state, int* __restrict__ ccount) {
    __shared__ int sh[256];
    __shared__ int red[8];
    const int r = row_of_chunk(a, blockIdx.x);
    const SelRow row = a.rows[r];
    unsigned T = 0u;
    if (row.n > row.k) {
        const SelState st = state[(size_t)1 * a.n_rows + r];
        // 256 bins: thread t owns bin 255 - t
        const int tid = threadIdx.x;
        const int c = hist8[(size_t)r * 256 + 255 - tid];
        sh[tid] = c;
        __syncthreads();
        if (tid == 0) {
            int acc = 0;
            const int want = row.k - st.above;
            int pick = 0, ab = 0;
            for (int t2 = 0; t2 < 256; ++t2) {
                if (acc + sh[t2] >= want) { pick = 255 - t2; ab = acc; break; }
                acc += sh[t2];
            }
            red[0] = pick; red[1] = st.above + ab;
        }
        __syncthreads();
        T = st.prefix | (unsigned)red[0];
        if (blockIdx.x == row.chunk0 && threadIdx.x == 0) state[(size_t)2 * a.n_rows + r] = SelState{T, red[1]};
        __syncthreads();
    }
    const int base = (blockIdx.x - row.chunk0) * SEL_C;
    int gt = 0, eq = 0;
    for (int t = 0; t < SEL_C / 256; ++t) {
        const int i = base + t * 256 + threadIdx.x;
        if (i < row.n) {
            const unsigned key = __builtin_bit_cast(unsigned, scores[row.score_off + i]);
            if (row.n <= row.k) ++gt; else { gt += key > T; eq += key == T; }
        }
    }
    gt = wave_sum_i(gt); eq = wave_sum_i(eq);
    if ((threadIdx.x & 63) == 0) { red[2 + (threadIdx.x >> 6)] = gt; sh[threadIdx.x >> 6] = eq; }
    __syncthreads();
    if (threadIdx.x == 0) {
        ccount[2 * blockIdx.x] = red[2] + red[3] + red[4] + red[5];
        ccount[2 * blockIdx.x + 1] = sh[0] + sh[1] + sh[2] + sh[3];
    }
}

// pass 4: ordered compaction - the kept elements of the row in index order: cand[row][0 .. k)
__global__ __launch_bounds__(256) void sel_scatter_kernel(SelArgs a,
                                                          const float* __restrict__ scores, const SelState* __restrict__ state,
                                                          const int* __restrict__ ccount, unsigned long long* __restrict__ cand) {
    __shared__ int sh[260];
    const int r = row_of_chunk(a, blockIdx.x);
    const SelRow row = a.rows[r];
    const bool all = row.n <= row.k;
    unsigned T = 0u;
    int need_eq = 0;
    if (!all) {
        const SelState st = state[(size_t)2 * a.n_rows + r];
        T = st.prefix;
        need_eq = row.k - st.above;                      // keys == T to keep (the first ones in index order)
    }
    // keys > T / == T in the chunks of this row before this one
    int gt_before = 0, eq_before = 0;
    for (int c = row.chunk0 + threadIdx.x; c < (int)blockIdx.x; c += 256) { gt_before += ccount[2 * c]; eq_before += ccount[2 * c + 1]; }
    gt_before = wave_sum_i(gt_before); eq_before = wave_sum_i(eq_before);
    if ((threadIdx.x & 63) == 0) { sh[256 + (threadIdx.x >> 6)] = gt_before; sh[threadIdx.x >> 6] = eq_before; }
    __syncthreads();
    gt_before = sh[256] + sh[257] + sh[258] + sh[259];
    eq_before = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    // this thread's 16 consecutive elements
    const int base = (blockIdx.x - row.chunk0) * SEL_C + threadIdx.x * 16;
    unsigned keys[16];
    int gt = 0, eq = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int i = base + j;
        keys[j] = i < row.n ? __builtin_bit_cast(unsigned, scores[row.score_off + i]) : 0u;
        if (i < row.n) { if (all || keys[j] > T) ++gt; else if (keys[j] == T) ++eq; }
    }
    // block-exclusive prefix of (gt, eq) over the threads
    int igt = gt, ieq = eq;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t1 = __shfl_up(igt, o, 64), t2 = __shfl_up(ieq, o, 64);
        if (lane >= o) { igt += t1; ieq += t2; }
    }
    if (lane == 63) { sh[wave] = igt; sh[8 + wave] = ieq; }
    __syncthreads();
    int wgt = 0, weq = 0;
    for (int w2 = 0; w2 < wave; ++w2) { wgt += sh[w2]; weq += sh[8 + w2]; }
    int pgt = gt_before + wgt + igt - gt, peq = eq_before + weq + ieq - eq;      // ranks of this thread's first gt / eq element
    // an element's output position = (number of kept elements before it in index order): kept = gt elements and the eq
    // elements of rank < need_eq; eq elements kept before index i = min(eq rank, need_eq)
    unsigned long long* out = cand + a.cand_off[r];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int i = base + j;
        if (i >= row.n) break;
        const bool is_gt = all || keys[j] > T, is_eq = !all && keys[j] == T;
        if (is_gt || (is_eq && peq < need_eq)) {
            const int pos = pgt + (peq < need_eq ? peq : need_eq);
            // composite sort key: score bits high, (inverted) position low - descending order = score desc, index asc
            out[pos] = ((unsigned long long)keys[j] << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i);
        }
        pgt += is_gt; peq += is_eq;
    }
}

// pass 5: one workgroup per row sorts its k candidates (bitonic, descending on the composite key) and writes
// scores [k] / index [k]
__global__ __launch_bounds__(1024) void sel_sort_kernel(SelArgs a,
                                                        const unsigned long long* __restrict__ cand) {
    extern __shared__ unsigned long long sk[];
    const SelRow row = a.rows[blockIdx.x];
    const int k = row.k < row.n ? row.k : row.n;
    int n2 = 1;
    while (n2 < k) n2 <<= 1;
    const unsigned long long* in = cand + a.cand_off[blockIdx.x];
    for (int i = threadIdx.x; i < n2; i += 1024) sk[i] = i < k ? in[i] : 0ull;       // padding sorts last
    __syncthreads();
    for (int size = 2; size <= n2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < n2 / 2; t += 1024) {
                const int lo = ((t / stride) * stride * 2) + (t % stride), hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const unsigned long long x = sk[lo], y = sk[hi];
                if ((x < y) == desc) { sk[lo] = y; sk[hi] = x; }
            }
            __syncthreads();
        }
    const SelLevel L = a.lv[row.level];
    for (int i = threadIdx.x; i < k; i += 1024) {
        const unsigned long long v = sk[i];
        L.out_scores[row.out_off + i] = __builtin_bit_cast(float, (unsigned)(v >> 32));
        L.out_index[row.out_off + i] = (long long)(0xFFFFFFFFu - (unsigned)(v & 0xFFFFFFFFull));
    }
}

}  // namespace

extern "C" {

// Workspace layout (ints unless noted), n_rows = n_img * n_levels, n_chunks = sum of ceil(n / 4096):
//   hist12a [n_rows][4096] | hist12b [n_rows][4096] | hist8 [n_rows][256] |
//   state [3][n_rows] | ccount [n_chunks][2] | scores fp32 [sum n] | cand u64 [sum min(n, k)]
size_t oadg_rpn_topk_workspace_bytes(const int* level_n, int n_levels, int n_img, int nms_pre) {
    if (!level_n || n_levels < 1 || n_img < 1) return 0;
    size_t rows = (size_t)n_levels * n_img, chunks = 0, nsum = 0, ksum = 0;
    for (int l = 0; l < n_levels; ++l) {
        chunks += (size_t)n_img * ((level_n[l] + SEL_C - 1) / SEL_C);
        nsum += (size_t)n_img * level_n[l];
        ksum += (size_t)n_img * (level_n[l] < nms_pre ? level_n[l] : nms_pre);
    }
    size_t b = rows * (4096 + 4096 + 256) * 4 + 3 * rows * sizeof(SelState) + chunks * 8;
    b = (b + 15) / 16 * 16 + nsum * 4;
    b = (b + 15) / 16 * 16 + ksum * 8 + 64;
    return b;
}

// cls[l]: cls_score of level l (dtype 0 fp32 / 1 bf16, logical [N, A, H, W], element strides strides[4 l .. 4 l + 3] =
// sN, sC, sH, sW); out_scores[l] [n_img, k_l] fp32, out_index[l] [n_img, k_l] int64 with k_l = min(nms_pre, A H W): the
// sigmoid scores of every image in stable descending order and their positions (h*W + w)*A + a.
int oadg_rpn_topk(const void* const* cls, const long* strides, const int* dims /* [l][3] = H, W, A */, int dtype,
                  int n_levels, int n_img, int nms_pre, float* const* out_scores, long long* const* out_index,
                  void* workspace, size_t workspace_bytes, void* stream) {
    if (!cls || !strides || !dims || !out_scores || !out_index || !workspace || n_levels < 1 ||
        n_levels > OADG_RPN_MAX_LEVELS || n_img < 1 || nms_pre < 1 || (dtype != 0 && dtype != 1))
        return OADG_EARG;
    int level_n[OADG_RPN_MAX_LEVELS];
    for (int l = 0; l < n_levels; ++l) level_n[l] = dims[3 * l] * dims[3 * l + 1] * dims[3 * l + 2];
    if (workspace_bytes < oadg_rpn_topk_workspace_bytes(level_n, n_levels, n_img, nms_pre)) return OADG_ESIZE;
    SelArgs a;
    a.n_levels = n_levels;
    const int n_rows = n_levels * n_img;
    a.n_rows = n_rows;
    if (n_rows > SEL_MAX_ROWS) return OADG_EARG;                 // (the row table travels as kernel arguments)
    SelRow* hrows = a.rows;
    int* hoff = a.cand_off;
    int chunk = 0, coff = 0, kmax = 0;
    long soff = 0;
    for (int l = 0; l < n_levels; ++l) {
        SelLevel& L = a.lv[l];
        L.cls = cls[l]; L.sN = strides[4 * l]; L.sC = strides[4 * l + 1]; L.sH = strides[4 * l + 2]; L.sW = strides[4 * l + 3];
        L.H = dims[3 * l]; L.W = dims[3 * l + 1]; L.A = dims[3 * l + 2]; L.dtype = dtype;
        L.out_scores = out_scores[l]; L.out_index = out_index[l];
        if (!L.cls || !L.out_scores || !L.out_index || level_n[l] < 1) return OADG_EARG;
        const int k = level_n[l] < nms_pre ? level_n[l] : nms_pre;
        kmax = k > kmax ? k : kmax;
        for (int i = 0; i < n_img; ++i) {
            SelRow& r = hrows[l * n_img + i];
            r.score_off = soff; r.n = level_n[l]; r.k = k; r.level = l; r.img = i; r.chunk0 = chunk;
            r.nchunk = (level_n[l] + SEL_C - 1) / SEL_C; r.out_off = i * k;
            hoff[l * n_img + i] = coff;
            chunk += r.nchunk; soff += level_n[l]; coff += k;
        }
    }
    a.n_chunks = chunk;
    int n2 = 1;
    while (n2 < kmax) n2 <<= 1;
    if ((size_t)n2 * 8 > 150 * 1024) return OADG_EARG;                      // k <= 16384 candidates per row
    unsigned char* w = (unsigned char*)workspace;
    int* hist_a = (int*)w;                    w += (size_t)n_rows * 4096 * 4;
    int* hist_b = (int*)w;                    w += (size_t)n_rows * 4096 * 4;
    int* hist_c = (int*)w;                    w += (size_t)n_rows * 256 * 4;
    SelState* state = (SelState*)w;           w += (size_t)3 * n_rows * sizeof(SelState);
    int* ccount = (int*)w;                    w += (size_t)chunk * 8;
    w = (unsigned char*)(((uintptr_t)w + 15) / 16 * 16);
    float* scores = (float*)w;                w += (size_t)soff * 4;
    w = (unsigned char*)(((uintptr_t)w + 15) / 16 * 16);
    unsigned long long* cand = (unsigned long long*)w;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(hist_a, 0, (size_t)n_rows * (4096 + 4096 + 256) * 4, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(sel_score_kernel, dim3(chunk), dim3(256), 0, st, a, scores, hist_a);
    OADG_LAUNCH_CHECK();
    hipLaunchKernelGGL(sel_refine_kernel<1>, dim3(chunk), dim3(256), 0, st, a, scores, hist_a, hist_b, state);
    OADG_LAUNCH_CHECK();
    hipLaunchKernelGGL(sel_refine_kernel<2>, dim3(chunk), dim3(256), 0, st, a, scores, hist_b, hist_c, state);
    OADG_LAUNCH_CHECK();
    hipLaunchKernelGGL(sel_count_kernel2, dim3(chunk), dim3(256), 0, st, a, scores, hist_c, state, ccount);
    OADG_LAUNCH_CHECK();
    hipLaunchKernelGGL(sel_scatter_kernel, dim3(chunk), dim3(256), 0, st, a, scores, state, ccount, cand);
    OADG_LAUNCH_CHECK();
    static size_t lds_set = 0;
    const size_t lds = (size_t)n2 * 8;
    if (lds > 48 * 1024 && lds > lds_set) {
        e = hipFuncSetAttribute((const void*)sel_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) return (int)e;
        lds_set = 150 * 1024;
    }
    hipLaunchKernelGGL(sel_sort_kernel, dim3(n_rows), dim3(1024), lds, st, a, cand);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

}  // extern "C"
