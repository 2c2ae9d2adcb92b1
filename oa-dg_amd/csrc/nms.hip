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
//      chunk's diagonal block, the updates of the next two chunks are OR-reductions of prefetched words and the
//      farther ones loads consumed a chunk later: no memory round trip on the chain (round 2: 370 -> ~100 us).
#include "common.h"
#include <type_traits>

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

// OR over the 64 lanes, the same value in every lane: four DPP row rotations inside each row of 16 lanes, then one
// v_readlane per row (ds_bpermute-based shuffles cost a ~6-deep LDS round-trip chain on the scan's critical path)
__device__ __forceinline__ unsigned wave_or32(unsigned v) {
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);     // row_ror:8
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false);     // row_ror:4
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xf, 0xf, false);     // row_ror:2
    v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xf, 0xf, false);     // row_ror:1
    return (unsigned)__builtin_amdgcn_readlane((int)v, 0) | (unsigned)__builtin_amdgcn_readlane((int)v, 16) |
           (unsigned)__builtin_amdgcn_readlane((int)v, 32) | (unsigned)__builtin_amdgcn_readlane((int)v, 48);
}
__device__ __forceinline__ unsigned long long wave_or64(unsigned long long v) {
    return ((unsigned long long)wave_or32((unsigned)(v >> 32)) << 32) | wave_or32((unsigned)v);
}

// One wave per image.  The greedy dependency chain must not contain a global-memory round trip per 64-row chunk (149
// chunks at 9536 proposals: the first form of this kernel waited ~2 us per chunk for the kept rows' words, 370 us per
// launch).  Per chunk c:
//   * lane l holds three words of row 64 c + l, prefetched one chunk ahead: the diagonal word (what the row suppresses
//     inside its own chunk) and the words of chunks c + 1 and c + 2;
//   * the serial walk over the chunk's 64 bits runs on v_readlane of the diagonal words;
//   * the kept rows' contributions to the next two chunks are an OR-reduction over the kept lanes of the prefetched
//     words - no load on the chain;
//   * their contributions to chunks > c + 2 are plain loads whose results are OR-ed in one chunk LATER (the first 8 kept
//     rows; further ones, rare, immediately): they are first needed at chunk c + 3.
// SLOTS = removed-set words per lane (64 SLOTS chunks); ROUNDS = rounds of 8 kept rows per chunk whose far words are
// consumed a chunk later (RPN proposals keep 10-30 rows per chunk until max_keep is reached).
template <int SLOTS, int ROUNDS>
__global__ __launch_bounds__(64) void nms_scan_kernel(const unsigned long long* __restrict__ mask,
                                                      const int* __restrict__ counts, int Mmax,
                                                      int words, int max_keep, int* __restrict__ keep,
                                                      int* __restrict__ keep_cnt) {
    const int img = blockIdx.x, lane = threadIdx.x;
    const int M = min(counts[img], Mmax);
    const unsigned long long* mk = mask + (size_t)img * Mmax * words;
    int* out = keep + (size_t)img * Mmax;
    unsigned long long removed[SLOTS];       // lane l owns words l, l + 64, ...
    unsigned long long pend[2][ROUNDS * 8][SLOTS];   // far words of the first ROUNDS * 8 kept rows of chunks c - 2 / c - 1
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        removed[s] = 0ull;
#pragma unroll
        for (int u = 0; u < ROUNDS * 8; ++u) pend[0][u][s] = pend[1][u][s] = 0ull;
    }
    int nkeep = 0, pend_rounds[2] = {0, 0};
    const int nchunks = (M + 63) / 64;
    // words (c, c + 1, c + 2) of row 64 c + lane; only words of valid rows at / right of the diagonal and of existing
    // chunks are ever written by the mask kernel
    auto row_words = [&](int c, unsigned long long& d, unsigned long long& n1, unsigned long long& n2) {
        const int row = c * 64 + lane;
        const bool ok = c < nchunks && row < M;
        const unsigned long long* r = mk + (size_t)(ok ? row : 0) * words;
        d = ok ? r[c] : 0ull;
        n1 = (ok && c + 1 < nchunks) ? r[c + 1] : 0ull;
        n2 = (ok && c + 2 < nchunks) ? r[c + 2] : 0ull;
    };
    auto merge = [&](int w, unsigned long long v) {       // removed word w |= v (at its owner lane)
#pragma unroll
        for (int s = 0; s < SLOTS; ++s)
            if (s == (w >> 6) && lane == (w & 63)) removed[s] |= v;
    };
    unsigned long long diag, near1, near2;
    row_words(0, diag, near1, near2);
    auto chunk = [&](int cw, auto set_c) {
        constexpr int SET = decltype(set_c)::value;
        // this chunk's removed word from its owner lane (every contribution of earlier chunks has been merged)
        unsigned long long mine = 0ull;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s)
            if (s == (cw >> 6)) mine = removed[s];
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)mine, cw & 63);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(mine >> 32), cw & 63);
        unsigned long long cur = ((unsigned long long)hi << 32) | lo;
        unsigned long long ndiag, nnear1, nnear2;
        row_words(cw + 1, ndiag, nnear1, nnear2);          // in flight during the walk
        const int nb = min(64, M - cw * 64);
        const int myrow = cw * 64 + lane;
        // the walk visits only the rows still alive (10-30 of 64 for RPN proposals), on the scalar unit: the removed
        // word is uniform, the next survivor is a find-first-set, its diagonal word comes by v_readlane
        unsigned long long avail = ~cur & (nb == 64 ? ~0ull : ((1ull << nb) - 1ull));
        unsigned long long kept = 0ull;
        int nk = nkeep;
        while (avail && nk < max_keep) {
            const int bit = __ffsll((long long)avail) - 1;
            kept |= 1ull << bit;
            ++nk;
            const unsigned dlo = __builtin_amdgcn_readlane((int)(unsigned)diag, bit);
            const unsigned dhi = __builtin_amdgcn_readlane((int)(unsigned)(diag >> 32), bit);
            const unsigned long long d = ((unsigned long long)dhi << 32) | dlo;
            avail &= ~d & ~((2ull << bit) - 1ull);      // minus what this row suppresses, minus the visited bits
        }
        const bool me_kept = (kept >> lane) & 1ull;
        if (me_kept) out[nkeep + __popcll(kept & ((1ull << lane) - 1ull))] = myrow;
        nkeep = nk;
        // near words: OR over the kept rows of what each lane prefetched
        merge(cw + 1, wave_or64(me_kept ? near1 : 0ull));
        merge(cw + 2, wave_or64(me_kept ? near2 : 0ull));
        // far words of the kept rows of chunk cw - 2 (loaded two chunks ago into this set: first needed at chunk cw + 1)
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd)
            if (rd < pend_rounds[SET]) {                       // (uniform) only the rounds that were issued
#pragma unroll
                for (int s = 0; s < SLOTS; ++s)
#pragma unroll
                    for (int u = 0; u < 8; ++u) removed[s] |= pend[SET][rd * 8 + u][s];
            }
        // far words (> cw + 2) of this chunk's kept rows: 8 rows per round, the first round is consumed a chunk later
        unsigned long long todo = kept;
        auto far_round = [&](int* bits) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                bits[u] = todo ? __ffsll((long long)todo) - 1 : -1;
                todo &= todo - 1ull;
            }
        };
        pend_rounds[SET] = 0;
#pragma unroll
        for (int rd = 0; rd < ROUNDS; ++rd) {                  // deferred rounds, as many as there are kept rows
            if (!todo) break;
            ++pend_rounds[SET];
            int bits[8];
            far_round(bits);
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                const int w = lane + 64 * s;
                const bool far = w > cw + 2 && w < nchunks;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    pend[SET][rd * 8 + u][s] = (far && bits[u] >= 0) ? mk[(size_t)(cw * 64 + bits[u]) * words + w] : 0ull;
            }
        }
        while (todo) {                                         // more kept rows than deferred slots: immediately
            int bits[8];
            far_round(bits);
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                const int w = lane + 64 * s;
                const bool far = w > cw + 2 && w < nchunks;
                unsigned long long v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    v[u] = (far && bits[u] >= 0) ? mk[(size_t)(cw * 64 + bits[u]) * words + w] : 0ull;
#pragma unroll
                for (int u = 0; u < 8; ++u) removed[s] |= v[u];
            }
        }
        diag = ndiag; near1 = nnear1; near2 = nnear2;
    };
    for (int cw = 0; cw < nchunks && nkeep < max_keep; cw += 2) {
        chunk(cw, std::integral_constant<int, 0>{});
        if (cw + 1 < nchunks && nkeep < max_keep) chunk(cw + 1, std::integral_constant<int, 1>{});
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
    if (words <= 64 * 3)        // <= 12288 boxes per image (RPN: 9536): three words per lane, 16 deferred rows per chunk (two sets)
        hipLaunchKernelGGL((nms_scan_kernel<3, 2>), dim3(n_images), dim3(64), 0, st,
                           (const unsigned long long*)workspace, counts, Mmax, words, max_keep, keep, keep_cnt);
    else
        hipLaunchKernelGGL((nms_scan_kernel<MAX_WORDS_PER_LANE, 1>), dim3(n_images), dim3(64), 0, st,
                           (const unsigned long long*)workspace, counts, Mmax, words, max_keep, keep, keep_cnt);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

}  // extern "C"
