// RandomSampler.sample of the RoI head for a whole batch ON THE DEVICE, with ATen's CPU generator stream (bit-exact):
//   mmdet/core/bbox/samplers/base_sampler.py:38-103   sample(): positives first (num * pos_fraction), then negatives
//   mmdet/core/bbox/samplers/random_sampler.py:32-82  _sample_pos / _sample_neg / random_choice:
//        candidates = nonzero(gt_inds > 0) resp. (== 0) in ascending order; when there are more than wanted,
//        chosen = candidates[torch.randperm(n_candidates)[:wanted]], then .unique() (= sorted)
// torch.randperm(n) on the CPU generator (ATen randperm_cpu, n < 2^32 / 20) is a forward Fisher-Yates shuffle fed by the
// 32-bit MT19937 outputs - for i in [0, n-1): z = mt() % (n - i); swap(r[i], r[i + z]) - so r[0..k) is final after k
// iterations and the other n - 1 - k iterations only consume one draw each (csrc/host_rng.hip replays the same on the host
// for the RPN's 520k candidates).  Rounds 1-4 read the candidate counts back to the host at this point - the one host
// wait of the step behind which the device idles (the RoI head's launches are enqueued from zero lead) - drew there and
// uploaded the ranks.  Here the engine state lives in device memory (624 words + left + next, the layout of
// at::mt19937_data_pod) and ONE launch does the whole of it for all images:
//   workgroup b = image b: counts the candidates of images 0..b (the draws images 0..b-1 consume are a function of those
//   counts), twists the state forward to its own segment of the stream (<= 8 twists for 4 x 1100 candidates; every
//   workgroup repeats the few KB of work instead of waiting for its predecessor), runs the k Fisher-Yates steps of the
//   prefix on a dense rank array in LDS (one lane, ~80 cycles per step), marks the chosen ranks and compacts the chosen
//   candidates in index order.  The last workgroup writes the advanced engine state back.
// Output rows have a FIXED capacity of `num` per image (positives, then negatives): with the usual >= num candidates
// per image every downstream shape is known on the host without a read; an image with fewer raises flags[b] bit 0 and the
// trainer repeats the step through the host path (apis.TrainEngine).
#include "common.h"
#include "../../include/oadg_hip.h"

namespace {

constexpr int RS_MAXN = 4096;          // rows (gts + proposals) per image the LDS arrays hold
constexpr int RS_MAX_IMAGES = 8;
constexpr int MT_N = 624, MT_M = 397;

struct RoiSampleArgs {
    const long long* gt[RS_MAX_IMAGES];
    int n[RS_MAX_IMAGES];
    int B, num, num_pos_exp;
    float neg_pos_ub;
    const unsigned* mt;        // [626]: state[624], left, next - read-only for the whole launch
    unsigned* mt_out;          // [626]: the advanced state (written by the last image's workgroup only)
    long long* sel;            // [B][num]
    int* counts;               // [B][2] = k_pos, k_neg
    int* flags;                // [B]: bit 0 = fewer than num rows sampled, bit 1 = image larger than RS_MAXN (nothing sampled)
};

__device__ __forceinline__ unsigned mt_twist(unsigned u, unsigned v) {
    return (((u & 0x80000000u) | (v & 0x7fffffffu)) >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ unsigned mt_temper(unsigned y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// at::mt19937::next_state() on the LDS copy: three dependent phases of <= 227 independent words + the last word
__device__ void mt_reload(unsigned* s) {
    const int t = threadIdx.x;
    unsigned v = 0;
    if (t < MT_N - MT_M) v = s[t + MT_M] ^ mt_twist(s[t], s[t + 1]);
    __syncthreads();
    if (t < MT_N - MT_M) s[t] = v;
    __syncthreads();
    const int i2 = t + (MT_N - MT_M);
    if (t < MT_N - MT_M) v = s[i2 - (MT_N - MT_M)] ^ mt_twist(s[i2], s[i2 + 1]);
    __syncthreads();
    if (t < MT_N - MT_M) s[i2] = v;
    __syncthreads();
    const int i3 = t + 2 * (MT_N - MT_M);
    if (i3 < MT_N - 1) v = s[i3 - (MT_N - MT_M)] ^ mt_twist(s[i3], s[i3 + 1]);
    __syncthreads();
    if (i3 < MT_N - 1) s[i3] = v;
    __syncthreads();
    if (t == 0) s[MT_N - 1] = s[MT_M - 1] ^ mt_twist(s[MT_N - 1], s[0]);
    __syncthreads();
}

__device__ __forceinline__ int block_excl_scan(int v, int* buf, int* total) {
    // exclusive prefix sum over the 256 threads (buf: 256 + 4 ints of LDS); *total = the sum
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    __syncthreads();
    if (lane == 63) buf[w] = incl;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) base += i < w ? buf[i] : 0;
    *total = buf[0] + buf[1] + buf[2] + buf[3];
    return base + incl - v;
}

__device__ __forceinline__ bool rs_cand(long long v, int mode) { return mode == 0 ? v > 0 : v == 0; }

__global__ __launch_bounds__(256) void roi_sample_kernel(const RoiSampleArgs a) {
    __shared__ unsigned s[MT_N];
    __shared__ unsigned draws[2][512 + 8];          // the first k draws of this image's positive / negative permutation
    __shared__ int arr[RS_MAXN];
    __shared__ unsigned char chosen[RS_MAXN];
    __shared__ int sbuf[8];
    __shared__ int cnt[RS_MAX_IMAGES][2];
    const int img = blockIdx.x, tid = threadIdx.x;

    // ---- candidate counts of images 0 .. img (compile-time indices into the by-value argument block)
    bool too_big = false;
#pragma unroll
    for (int j = 0; j < RS_MAX_IMAGES; ++j) {
        if (j > img || j >= a.B) continue;
        const long long* g = a.gt[j];
        const int n = a.n[j];
        too_big = too_big || n > RS_MAXN;
        int cp = 0, cn = 0;
        for (int i = tid; i < n; i += 256) {
            const long long v = g[i];
            cp += v > 0 ? 1 : 0;
            cn += v == 0 ? 1 : 0;
        }
        cp = wave_sum_i(cp);
        cn = wave_sum_i(cn);
        __syncthreads();
        if ((tid & 63) == 0) { sbuf[tid >> 6] = cp; sbuf[4 + (tid >> 6)] = cn; }
        __syncthreads();
        if (tid == 0) {
            cnt[j][0] = sbuf[0] + sbuf[1] + sbuf[2] + sbuf[3];
            cnt[j][1] = sbuf[4] + sbuf[5] + sbuf[6] + sbuf[7];
        }
    }
    __syncthreads();
    if (too_big || a.num > 512) {                   // outside this kernel's domain: the trainer takes the host path
        if (tid == 0) { a.flags[img] = 3; a.counts[img * 2] = 0; a.counts[img * 2 + 1] = 0; }
        // (too_big covers images 0 .. img: the last workgroup sees every image - nothing is drawn, the state goes back as
        //  it came)
        if (img == a.B - 1)
            for (int i = tid; i < MT_N + 2; i += 256) a.mt_out[i] = a.mt[i];
        return;
    }
    // ---- the plan of every image up to this one (base_sampler.py:71-100), uniform
    long off = 0;                                   // draws consumed by images 0 .. img-1
    int np = 0, nn = 0, kp = 0, kn = 0;
    for (int j = 0; j <= img; ++j) {
        np = cnt[j][0]; nn = cnt[j][1];
        kp = np < a.num_pos_exp ? np : a.num_pos_exp;
        int ne = a.num - kp;
        if (a.neg_pos_ub >= 0.f) {
            const int ub = (int)(a.neg_pos_ub * (float)(kp > 1 ? kp : 1));
            ne = ne < ub ? ne : ub;
        }
        kn = nn < ne ? nn : ne;
        if (j < img) off += (np > kp ? np - 1 : 0) + (nn > kn ? nn - 1 : 0);
    }
    const bool draw_p = np > kp, draw_n = nn > kn;  // (n > wanted: torch.randperm(n) runs and consumes n - 1 draws)
    const long q0p = off, q0n = off + (draw_p ? np - 1 : 0);
    const long total = q0n + (draw_n ? nn - 1 : 0); // engine position after this image
    const int wantp = draw_p ? kp : 0, wantn = draw_n ? kn : 0;

    // ---- the engine: this image's draws out of the stream
    for (int i = tid; i < MT_N; i += 256) s[i] = a.mt[i];
    const int left = (int)a.mt[MT_N], next = (int)a.mt[MT_N + 1];
    __syncthreads();
    const long R = left - 1;                        // draws left in the current block (at s[next ..])
    const bool last = img == a.B - 1;
    const long q_end = last ? total : (wantn ? q0n + wantn : (wantp ? q0p + wantp : 0));   // one past the last draw needed
    long g0 = 0, g1 = R;                            // draw range of the block in s
    int base = next;
    int reloads = 0;
    while (true) {
        for (int sgm = 0; sgm < 2; ++sgm) {
            const long qa = sgm ? q0n : q0p;
            const int want = sgm ? wantn : wantp;
            const long lo = qa > g0 ? qa : g0, hi = qa + want < g1 ? qa + want : g1;
            for (long q = lo + tid; q < hi; q += 256) draws[sgm][q - qa] = mt_temper(s[base + (int)(q - g0)]);
        }
        if (q_end <= g1) break;
        __syncthreads();
        mt_reload(s);
        ++reloads;
        g0 = g1; g1 += MT_N; base = 0;
    }
    __syncthreads();
    if (last) {                                     // the advanced state: what torch's generator holds after the same draws
        int nl, nx;
        if (reloads == 0) { nl = left - (int)total; nx = next + (int)total; }
        else { const int u = (int)(total - g0); nl = MT_N - u + 1; nx = u; }
        // (into its OWN buffer: the other workgroups of this launch read a.mt in no defined order)
        for (int i = tid; i < MT_N; i += 256) a.mt_out[i] = s[i];
        if (tid == 0) { a.mt_out[MT_N] = (unsigned)nl; a.mt_out[MT_N + 1] = (unsigned)nx; }
    }

    // ---- per class: Fisher-Yates prefix on the candidate ranks, then the chosen candidates in index order
    const long long* g = nullptr;
    int n = 0;
#pragma unroll
    for (int j = 0; j < RS_MAX_IMAGES; ++j)
        if (j == img) { g = a.gt[j]; n = a.n[j]; }
    long long* out = a.sel + (long)img * a.num;
    for (int mode = 0; mode < 2; ++mode) {
        const int nc = mode ? nn : np, k = mode ? kn : kp;
        const bool all = !(mode ? draw_n : draw_p);
        if (k > 0 && !all) {
            unsigned* d = draws[mode];
            for (int q = tid; q < k; q += 256) d[q] = (unsigned)q + d[q] % (unsigned)(nc - q);
            for (int i = tid; i < nc; i += 256) { arr[i] = i; chosen[i] = 0; }
            __syncthreads();
            if (tid == 0)
                for (int q = 0; q < k; ++q) {
                    const int j = (int)d[q];
                    const int x = arr[q], y = arr[j];
                    arr[q] = y; arr[j] = x;
                }
            __syncthreads();
            for (int q = tid; q < k; q += 256) chosen[arr[q]] = 1;
        }
        __syncthreads();
        if (k > 0) {
            // thread t owns rows [t * per, (t + 1) * per): candidate rank base, then output base
            const int per = (n + 255) / 256;
            const int r0 = tid * per, r1 = r0 + per < n ? r0 + per : n;
            int c = 0;
            for (int i = r0; i < r1; ++i) c += rs_cand(g[i], mode) ? 1 : 0;
            int tot;
            int rank = block_excl_scan(c, sbuf, &tot);
            int m = 0;
            if (all) m = c;
            else { int r = rank; for (int i = r0; i < r1; ++i) if (rs_cand(g[i], mode)) { m += chosen[r]; ++r; } }
            int o = block_excl_scan(m, sbuf, &tot);
            for (int i = r0; i < r1; ++i)
                if (rs_cand(g[i], mode)) {
                    if (all || chosen[rank]) out[o++] = i;
                    ++rank;
                }
        }
        __syncthreads();
        out += k;
    }
    if (tid == 0) {
        a.counts[img * 2] = kp;
        a.counts[img * 2 + 1] = kn;
        a.flags[img] = (kp + kn != a.num) ? 1 : 0;
    }
}

}  // namespace

extern "C" int oadg_roi_sample_max_rows(void) { return RS_MAXN; }

extern "C" int oadg_roi_sample_device(const oadg_roi_sample_image* images_host, int B, int num, int num_pos_exp,
                                      float neg_pos_ub, const uint32_t* mt_state, uint32_t* mt_state_out, int64_t* sel,
                                      int* counts, int* flags, void* stream) {
    if (!images_host || B < 1 || B > RS_MAX_IMAGES || num < 1 || num_pos_exp < 0 || num_pos_exp > num || !mt_state ||
        !mt_state_out || mt_state_out == mt_state || !sel || !counts || !flags)
        return OADG_EARG;
    RoiSampleArgs a;
    for (int i = 0; i < RS_MAX_IMAGES; ++i) { a.gt[i] = nullptr; a.n[i] = 0; }
    for (int i = 0; i < B; ++i) {
        if (images_host[i].n < 0 || (images_host[i].n > 0 && !images_host[i].gt_inds)) return OADG_EARG;
        a.gt[i] = (const long long*)images_host[i].gt_inds;
        a.n[i] = images_host[i].n;
    }
    a.B = B; a.num = num; a.num_pos_exp = num_pos_exp; a.neg_pos_ub = neg_pos_ub;
    a.mt = mt_state; a.mt_out = mt_state_out; a.sel = (long long*)sel; a.counts = counts; a.flags = flags;
    hipLaunchKernelGGL(roi_sample_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, a);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}
