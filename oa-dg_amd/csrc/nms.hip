// Greedy IoU non-maximum suppression, batched over images, for gfx950.
//
// Replaces, behind the C ABI in include/oadg_hip.h:
//   mmcv.ops.nms / mmcv.ops.batched_nms (mmcv-full, not vendored; semantics SURVEY.md A.4) as called at
//     mmdet/models/dense_heads/rpn_head.py:231
//
// Contract: boxes of every image are already sorted by descending score (the caller's sort defines the
// tie order) and already carry the per-class coordinate offset of batched_nms.  Box i suppresses a later
// box j when IoU(i, j) > thr, with IoU = inter / (area_i + area_j - inter) in fp32 (offset = 0).
//
// Two kernels, no host round trip (mmcv copies the bit matrix to the host for the greedy scan):
//   1. nms_mask_kernel : 64x64 tiles of the upper triangle -> one 64-bit suppression word per (row, tile)
//   2. nms_scan_kernel : one wave per image walks the rows in order, 64 at a time; the `removed` bit set lives in
//      registers (lane l owns words l, l+64, ...), the within-chunk dependency chain runs on v_readlane of the
//      chunk's diagonal block, the off-diagonal updates are independent loads.
#include "common.h"

namespace {

constexpr int MAX_WORDS_PER_LANE = 8;  // M <= 64*64*8 = 32768 boxes per image

__device__ __forceinline__ float iou_f32(const float4 a, const float4 b) {
    const float left = fmaxf(a.x, b.x), right = fminf(a.z, b.z);
    const float top = fmaxf(a.y, b.y), bottom = fminf(a.w, b.w);
    const float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
    const float inter = w * h;
    const float sa = (a.z - a.x) * (a.w - a.y);
    const float sb = (b.z - b.x) * (b.w - b.y);
    return inter / (sa + sb - inter);
}

__global__ __launch_bounds__(64) void nms_mask_kernel(const float4* __restrict__ boxes,
                                                      const int* __restrict__ counts, int Mmax,
                                                      int words, float thr,
                                                      unsigned long long* __restrict__ mask) {
    const int img = blockIdx.z, rb = blockIdx.y, cb = blockIdx.x;
    if (cb < rb) return;
    const int M = min(counts[img], Mmax);
    if (rb * 64 >= M || cb * 64 >= M) return;
    __shared__ float4 colbox[64];
    const float4* b = boxes + (size_t)img * Mmax;
    const int t = threadIdx.x;
    const int cidx = cb * 64 + t;
    colbox[t] = cidx < M ? b[cidx] : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    const int r = rb * 64 + t;
    if (r >= M) return;
    const float4 me = b[r];
    unsigned long long bits = 0ull;
    const int ncol = min(64, M - cb * 64);
    const int start = (rb == cb) ? t + 1 : 0;
    for (int j = start; j < ncol; ++j)
        if (iou_f32(me, colbox[j]) > thr) bits |= 1ull << j;
    mask[((size_t)img * Mmax + r) * words + cb] = bits;
}

__global__ __launch_bounds__(64) void nms_scan_kernel(const unsigned long long* __restrict__ mask,
                                                      const int* __restrict__ counts, int Mmax,
                                                      int words, int max_keep, int* __restrict__ keep,
                                                      int* __restrict__ keep_cnt) {
    const int img = blockIdx.x, lane = threadIdx.x;
    const int M = min(counts[img], Mmax);
    const unsigned long long* mk = mask + (size_t)img * Mmax * words;
    int* out = keep + (size_t)img * Mmax;
    unsigned long long removed[MAX_WORDS_PER_LANE];
#pragma unroll
    for (int s = 0; s < MAX_WORDS_PER_LANE; ++s) removed[s] = 0ull;
    int nkeep = 0;
    const int nchunks = (M + 63) / 64;
    for (int cw = 0; cw < nchunks && nkeep < max_keep; ++cw) {
        // fetch this chunk's removed word from its owner lane
        unsigned long long mine = 0ull;
#pragma unroll
        for (int s = 0; s < MAX_WORDS_PER_LANE; ++s)
            if (s == (cw >> 6)) mine = removed[s];
        const unsigned lo = __shfl((unsigned)mine, cw & 63, 64);
        const unsigned hi = __shfl((unsigned)(mine >> 32), cw & 63, 64);
        unsigned long long cur = ((unsigned long long)hi << 32) | lo;
        const int nb = min(64, M - cw * 64);
        // The serial part of the greedy scan stays inside the chunk and inside registers: lane l holds the diagonal
        // word of row cw*64 + l (which boxes of this chunk row l suppresses); walking the 64 bits needs one
        // v_readlane pair per kept box instead of a dependent global load.
        const int myrow = cw * 64 + lane;
        const unsigned long long diag = lane < nb ? mk[(size_t)myrow * words + cw] : 0ull;
        unsigned long long kept = 0ull;
        int nk = nkeep;
        for (int bit = 0; bit < nb && nk < max_keep; ++bit) {
            if ((cur >> bit) & 1ull) continue;
            kept |= 1ull << bit;
            ++nk;
            const unsigned dlo = __builtin_amdgcn_readlane((int)(unsigned)diag, bit);
            const unsigned dhi = __builtin_amdgcn_readlane((int)(unsigned)(diag >> 32), bit);
            cur |= ((unsigned long long)dhi << 32) | dlo;
        }
        if ((kept >> lane) & 1ull) out[nkeep + __popcll(kept & ((1ull << lane) - 1ull))] = myrow;
        nkeep = nk;
        // the kept rows' words right of the diagonal: independent loads, no serial dependency
        // (8 rows per round so that the loads of a round are in flight together)
        unsigned long long todo = kept;
        while (todo) {
            int bits[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                bits[u] = todo ? __ffsll((long long)todo) - 1 : -1;
                todo &= todo - 1ull;
            }
#pragma unroll
            for (int s = 0; s < MAX_WORDS_PER_LANE; ++s) {
                const int w = lane + 64 * s;
                if (w > cw && w < nchunks) {
                    unsigned long long v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        v[u] = bits[u] >= 0 ? mk[(size_t)(cw * 64 + bits[u]) * words + w] : 0ull;
#pragma unroll
                    for (int u = 0; u < 8; ++u) removed[s] |= v[u];
                }
            }
        }
    }
    if (lane == 0) keep_cnt[img] = nkeep;
}

}  // namespace

extern "C" {

size_t oadg_nms_workspace_bytes(int n_images, int Mmax) {
    if (n_images <= 0 || Mmax <= 0) return 0;
    const size_t words = (size_t)(Mmax + 63) / 64;
    return (size_t)n_images * Mmax * words * sizeof(unsigned long long);
}

// boxes [n_images, Mmax, 4] fp32 (x1,y1,x2,y2), counts [n_images] device ints (valid boxes per image),
// keep [n_images, Mmax] int32 (indices into the sorted order, ascending), keep_cnt [n_images].
int oadg_nms_batched(const float* boxes, const int* counts, int n_images, int Mmax, float iou_thr,
                     int max_keep, void* workspace, size_t workspace_bytes, int* keep, int* keep_cnt,
                     void* stream) {
    if (!boxes || !counts || !workspace || !keep || !keep_cnt) return OADG_EARG;
    if (n_images < 1 || Mmax < 1 || Mmax > 64 * 64 * MAX_WORDS_PER_LANE) return OADG_EARG;
    if (workspace_bytes < oadg_nms_workspace_bytes(n_images, Mmax)) return OADG_ESIZE;
    if (max_keep <= 0 || max_keep > Mmax) max_keep = Mmax;
    hipStream_t st = (hipStream_t)stream;
    const int words = (Mmax + 63) / 64;
    // rows of partially filled images keep stale words otherwise: the scan only reads words of valid
    // rows at or right of the diagonal, all of which the mask kernel writes, so no memset is needed.
    hipLaunchKernelGGL(nms_mask_kernel, dim3(words, words, n_images), dim3(64), 0, st,
                       (const float4*)boxes, counts, Mmax, words, iou_thr, (unsigned long long*)workspace);
    OADG_LAUNCH_CHECK();
    hipLaunchKernelGGL(nms_scan_kernel, dim3(n_images), dim3(64), 0, st,
                       (const unsigned long long*)workspace, counts, Mmax, words, max_keep, keep, keep_cnt);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

}  // extern "C"
