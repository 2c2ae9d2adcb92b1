// Host-side helper (no device code): the first k entries of torch.randperm(n) on the CPU generator, in
// O(k + n/1000) instead of the O(n) cache-missing shuffle.
//
// RandomSampler draws ``torch.randperm(n)[:num]`` with n = number of candidate anchors (~520k per image for
// the RPN's negatives, mmdet/core/bbox/samplers/random_sampler.py:58).  ATen's CPU randperm is a forward
// Fisher-Yates shuffle - for i in [0, n-1): z = mt19937() % (n - i); swap(r[i], r[i + z]) - so r[0..k) is
// final after k iterations; the remaining n-1-k iterations only consume one 32-bit draw each.  This
// function replays exactly that on a copy of the generator's MT19937 state: identical indices, identical
// generator state afterwards, so seeded runs stay sample-for-sample equal to the reference.
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
