// Host-side helper (no device code): the first k entries of torch.randperm(n) on the CPU generator, in
// O(k + n/1000) instead of the O(n) cache-missing shuffle.
//
// RandomSampler draws ``torch.randperm(n)[:num]`` with n = number of candidate anchors (~520k per image for
// the RPN's negatives, mmdet/core/bbox/samplers/random_sampler.py:58).  ATen's CPU randperm is a forward
// Fisher-Yates shuffle - for i in [0, n-1): z = mt19937() % (n - i); swap(r[i], r[i + z]) - so r[0..k) is
// final after k iterations; the remaining n-1-k iterations only consume one 32-bit draw each.  This
// function replays exactly that on a copy of the generator's MT19937 state: identical indices, identical
// generator state afterwards, so seeded runs stay sample-for-sample equal to the reference.
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "oadg_hip.h"

namespace {
constexpr int MT_N = 624, MT_M = 397;
constexpr uint32_t MATRIX_A = 0x9908b0dfu, UMASK = 0x80000000u, LMASK = 0x7fffffffu;

struct MT {
    uint32_t s[MT_N];
    int left;
    uint32_t next;
    static inline uint32_t twist(uint32_t u, uint32_t v) {
        return (((u & UMASK) | (v & LMASK)) >> 1) ^ ((v & 1u) ? MATRIX_A : 0u);
    }
    void reload() {
        uint32_t* p = s;
        left = MT_N;
        next = 0;
        for (int j = MT_N - MT_M + 1; --j; p++) *p = p[MT_M] ^ twist(p[0], p[1]);
        for (int j = MT_M; --j; p++) *p = p[MT_M - MT_N] ^ twist(p[0], p[1]);
        *p = p[MT_M - MT_N] ^ twist(p[0], s[0]);
    }
    inline uint32_t draw() {
        if (--left == 0) reload();
        uint32_t y = s[next++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
};

struct SparsePerm {   // position -> value for the few displaced positions (identity elsewhere)
    int64_t* key;
    int64_t* val;
    size_t cap;
    explicit SparsePerm(size_t n) {
        cap = 64;
        while (cap < 4 * n + 16) cap <<= 1;
        key = (int64_t*)malloc(cap * sizeof(int64_t));
        val = (int64_t*)malloc(cap * sizeof(int64_t));
        for (size_t i = 0; i < cap; ++i) key[i] = -1;
    }
    ~SparsePerm() { free(key); free(val); }
    size_t slot(int64_t p) const {
        size_t h = ((uint64_t)p * 0x9E3779B97F4A7C15ull) & (cap - 1);
        while (key[h] != -1 && key[h] != p) h = (h + 1) & (cap - 1);
        return h;
    }
    int64_t get(int64_t p) const { size_t h = slot(p); return key[h] == p ? val[h] : p; }
    void set(int64_t p, int64_t v) { size_t h = slot(p); key[h] = p; val[h] = v; }
};
}  // namespace

extern "C" int oadg_host_randperm_prefix(uint64_t* state624, int* left, uint64_t* next, int64_t n, int64_t k,
                                          int64_t* out) {
    if (!state624 || !left || !next || !out || n < 0 || k < 0) return -1;
    if (k > n) k = n;
    MT mt;
    for (int i = 0; i < MT_N; ++i) mt.s[i] = (uint32_t)state624[i];
    mt.left = *left;
    mt.next = (uint32_t)*next;
    SparsePerm perm((size_t)k);
    const int64_t total = n > 0 ? n - 1 : 0;
    const int64_t head = k < total ? k : total;
    for (int64_t i = 0; i < head; ++i) {
        const uint32_t r = mt.draw();
        const int64_t j = i + (int64_t)(r % (uint64_t)(n - i));
        const int64_t vi = perm.get(i), vj = perm.get(j);
        out[i] = vj;
        perm.set(j, vi);
    }
    // the other draws only advance the engine: skip whole blocks between reloads
    int64_t m = total - head;
    while (m > 0) {
        if (mt.left > 1) {
            const int64_t take = m < (int64_t)(mt.left - 1) ? m : (int64_t)(mt.left - 1);
            mt.left -= (int)take;
            mt.next += (uint32_t)take;
            m -= take;
        } else {
            (void)mt.draw();
            --m;
        }
    }
    if (k > head && k <= n) {   // entries past the last drawn position keep whatever the swaps left there
        for (int64_t i = head; i < k; ++i) out[i] = perm.get(i);
    }
    for (int i = 0; i < MT_N; ++i) state624[i] = mt.s[i];
    *left = mt.left;
    *next = mt.next;
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// OA-DG's random proposals (mmdet/models/detectors/two_stage.py:389-419, generate_random_bboxes_xy) drawn IN PLACE from
// numpy's global legacy generator: up to 500 rejection trials per image, each np.random.randint(0, w), randint(0, h),
// uniform(*scales), uniform(*ratios) and a fp32 IoU test against the image's gt boxes - ~100 trials per step of 4 + 4
// images, 1.5 - 2 ms of interpreter time on the training thread (tools/probe/step_cprofile.py).  `mt` is the address
// numpy publishes for the bit generator's state (np.random.mtrand._rand._bit_generator.ctypes.state_address: struct
// { uint32_t key[624]; int pos; }, numpy/random/src/mt19937/mt19937.h), so the draws consume the same stream, draw for
// draw, as the reference's python loop:
//   randint(0, n), legacy RandomState: masked rejection on 32-bit draws (_bounded_integers: buffered_bounded_masked_uint32;
//     n == 1 returns 0 without a draw);  uniform(a, b) = a + (b - a) * ((x >> 5) * 2^26 + (y >> 6)) / 2^53;
//   the IoU is mmdet/core/evaluation/bbox_overlaps.py's fp32 arithmetic, operation for operation (no contraction:
//   -ffp-contract=off), compared with the thresholds the way the caller's numpy compares them (the caller rounds them).
// n_gt < 0: no IoU test (bboxes_xy=None).  out [num][5] doubles (x1, y1, x2, y2, 1); returns the number of boxes, < 0 on error.
namespace {
struct NpMT {
    uint32_t key[MT_N];
    int pos;
};
inline void np_mt_gen(NpMT* s) {
    uint32_t y;
    int i;
    for (i = 0; i < MT_N - MT_M; i++) {
        y = (s->key[i] & UMASK) | (s->key[i + 1] & LMASK);
        s->key[i] = s->key[i + MT_M] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
    }
    for (; i < MT_N - 1; i++) {
        y = (s->key[i] & UMASK) | (s->key[i + 1] & LMASK);
        s->key[i] = s->key[i + (MT_M - MT_N)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
    }
    y = (s->key[MT_N - 1] & UMASK) | (s->key[0] & LMASK);
    s->key[MT_N - 1] = s->key[MT_M - 1] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX_A);
    s->pos = 0;
}
inline uint32_t np_u32(NpMT* s) {
    if (s->pos == MT_N) np_mt_gen(s);
    uint32_t y = s->key[s->pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}
inline double np_double(NpMT* s) {
    const int32_t a = np_u32(s) >> 5, b = np_u32(s) >> 6;
    return (a * 67108864.0 + b) / 9007199254740992.0;
}
inline int64_t np_randint0(NpMT* s, int64_t n) {       // np.random.randint(0, n), n >= 1
    const uint32_t rng = (uint32_t)(n - 1);
    if (rng == 0) return 0;
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    uint32_t v;
    while ((v = (np_u32(s) & mask)) > rng) {}
    return (int64_t)v;
}
}  // namespace

extern "C" int oadg_np_random_bboxes(void* mt, int img_width, int img_height, int num, const float* gts, int n_gt,
                                     double scale_lo, double scale_hi, double ratio_lo, double ratio_hi, int max_iters,
                                     double iou_max, double iou_min, double* out) {
    if (!mt || img_width < 1 || img_height < 1 || num < 0 || max_iters < 0 || (num > 0 && !out) || (n_gt > 0 && !gts)) return -1;
    if (n_gt == 0) return -2;                       // the reference takes np.max of an empty array here: the caller's business
    if ((int64_t)img_width > 0x7fffffffLL || (int64_t)img_height > 0x7fffffffLL) return -1;
    NpMT* s = (NpMT*)mt;
    if (s->pos < 0 || s->pos > MT_N) return -1;
    const double scale_rng = scale_hi - scale_lo, ratio_rng = ratio_hi - ratio_lo;
    const float eps = (float)1e-6;
    int total = 0;
    for (int it = 0; it < max_iters && total < num; ++it) {
        const int64_t x1 = np_randint0(s, img_width), y1 = np_randint0(s, img_height);
        const double u = scale_lo + scale_rng * np_double(s);
        const double scale = u * (double)img_height * (double)img_width;
        const double ratio = ratio_lo + ratio_rng * np_double(s);
        const int64_t w = (int64_t)sqrt(scale / ratio), h = (int64_t)sqrt(scale * ratio);
        const int64_t x2 = x1 + w < img_width ? x1 + w : img_width, y2 = y1 + h < img_height ? y1 + h : img_height;
        if (n_gt > 0) {
            const float bx1 = (float)x1, by1 = (float)y1, bx2 = (float)x2, by2 = (float)y2;
            const float area1 = (bx2 - bx1) * (by2 - by1);
            float best = 0.f;
            for (int g = 0; g < n_gt; ++g) {
                const float* q = gts + 4 * (size_t)g;
                const float area2 = (q[2] - q[0]) * (q[3] - q[1]);
                float ww = (bx2 < q[2] ? bx2 : q[2]) - (bx1 > q[0] ? bx1 : q[0]);
                float hh = (by2 < q[3] ? by2 : q[3]) - (by1 > q[1] ? by1 : q[1]);
                ww = ww > 0.f ? ww : 0.f;
                hh = hh > 0.f ? hh : 0.f;
                const float inter = ww * hh;
                float uni = area1 + area2 - inter;
                uni = uni > eps ? uni : eps;
                const float iou = inter / uni;
                if (g == 0 || iou > best) best = iou;
            }
            if ((double)best > iou_max || (double)best < iou_min) continue;
        }
        double* o = out + 5 * (size_t)total;
        o[0] = (double)x1; o[1] = (double)y1; o[2] = (double)x2; o[3] = (double)y2; o[4] = 1.0;
        ++total;
    }
    return total;
}
