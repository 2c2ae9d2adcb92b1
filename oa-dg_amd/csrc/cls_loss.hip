// OA-Loss classification consistency: view-1 cross-entropy + two-view Jensen-Shannon divergence,
// fused forward and fused backward, for gfx950.
//
// Replaces, behind the C ABI in include/oadg_hip.h:
//   mmdet/models/losses/oadg/cross_entropy_loss_plus.py:11-58     cross_entropy        (RoI head)
//   mmdet/models/losses/oadg/cross_entropy_loss_plus.py:82-130    binary_cross_entropy (RPN)
//   mmdet/models/losses/oadg/cross_entropy_loss_plus.py:264-319   jsdv1_3_2aug
//   mmdet/models/losses/oadg/cross_entropy_loss_plus.py:418-500   CrossEntropyLossPlus.forward
//
// Rows [0, R/2) are view 1 (the clean image), rows [R/2, R) view 2 (OA-Mix); row r pairs with r + R/2.
//   loss = loss_weight * sum_{r<R/2} w_r CE(x_r, y_r) / avg_factor
//        + lambda      * sum_{r<R/2} JSD(p(x_r), p(x_{r+R/2})) / avg_factor
// JSD = 1/2 sum_c [ p1 (ln p1 - ln M) + p2 (ln p2 - ln M) ],  M = clamp((p1+p2)/2, 1e-7, 1)
// with p = [sigmoid(x), 1 - sigmoid(x)] for the 1-logit RPN rows and softmax(x) for RoI rows.
// (jsdv1_3_2aug's "/ len(p_aug1)" divides by 1 - the tensor was reshaped to (1, R/2, C) - :299-310.)
// A probability that is exactly 0 contributes 0 to value and gradient (xlogy convention; the
// reference's autograd yields NaN there, which we do not reproduce).
//
// HBM-bound streaming kernels: one pass to reduce (per-block fp64 partials, fixed order => run-to-run
// deterministic), one pass to write d(logits).  Algorithmic traffic RPN: 2 x (4 B logit) + 8 B label +
// 4 B weight per pair forward, + 8 B of gradient backward.
#include <cstring>
#include "common.h"
#include "../../include/oadg_hip.h"

namespace {

constexpr int IGNORE_INDEX = -100;
constexpr int MAXBLOCKS = 2048;

__device__ __forceinline__ float xlogy_term(float t, float logm) {
    // F.kl_div(logM, t) = t * (ln t - logM), 0 at t == 0
    return t > 0.f ? t * (logf(t) - logm) : 0.f;
}
// d/dt of 1/2 [t1(ln t1 - ln M) + t2(ln t2 - ln M)] w.r.t. t1, M = clamp((t1+t2)/2, 1e-7, 1)
__device__ __forceinline__ float jsd_dterm(float t1, float t2, float mraw, float m, float logm) {
    if (!(t1 > 0.f)) return 0.f;
    float g = 0.5f * (logf(t1) + 1.0f - logm);
    if (mraw >= 1e-7f && mraw <= 1.0f) g -= 0.5f * (t1 + t2) * 0.5f / m;
    return g;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float bce_logits(float x, float t) {
    // torch binary_cross_entropy_with_logits: (1-t) x + max(-x,0) + log(exp(-max) + exp(-x-max))
    const float mx = fmaxf(-x, 0.f);
    return (1.0f - t) * x + mx + logf(expf(-mx) + expf(-x - mx));
}

// ---------------------------------------------------------------- sigmoid (RPN) rows
template <bool BWD>
__global__ void sig_kernel(const float* __restrict__ x, const int64_t* __restrict__ labels,
                           const float* __restrict__ weights, long half, float k_ce, float k_jsd,
                           const float* __restrict__ gout, double* __restrict__ part,
                           float* __restrict__ dx) {
    __shared__ double red[16];
    double ce = 0.0, js = 0.0;
    const float g0 = BWD ? (gout ? gout[0] : 1.0f) : 0.f;
    for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < half; r += (long)gridDim.x * blockDim.x) {
        const float x1 = x[r], x2 = x[r + half];
        const int64_t lab = labels[r];
        const bool valid = lab >= 0 && lab != IGNORE_INDEX;
        const float t = (valid && lab == 0) ? 1.0f : 0.0f;  // _expand_onehot_labels, 1 channel
        const float w = weights ? (valid ? weights[r] : 0.f) : (valid ? 1.f : 0.f);
        const float p1 = sigmoidf_(x1), q1 = 1.0f - p1;
        const float p2 = sigmoidf_(x2), q2 = 1.0f - p2;
        const float mpr = (p1 + p2) / 2.0f, mqr = (q1 + q2) / 2.0f;
        const float mp = fminf(fmaxf(mpr, 1e-7f), 1.0f), mq = fminf(fmaxf(mqr, 1e-7f), 1.0f);
        const float lmp = logf(mp), lmq = logf(mq);
        if (!BWD) {
            ce += (double)(w * bce_logits(x1, t));
            const float kl1 = xlogy_term(p1, lmp) + xlogy_term(q1, lmq);
            const float kl2 = xlogy_term(p2, lmp) + xlogy_term(q2, lmq);
            js += (double)((kl1 + kl2) / 2.0f);
        } else {
            const float d1 = jsd_dterm(p1, p2, mpr, mp, lmp) - jsd_dterm(q1, q2, mqr, mq, lmq);
            const float d2 = jsd_dterm(p2, p1, mpr, mp, lmp) - jsd_dterm(q2, q1, mqr, mq, lmq);
            const float g1 = k_ce * w * (p1 - t) + k_jsd * p1 * (1.0f - p1) * d1;
            const float g2 = k_jsd * p2 * (1.0f - p2) * d2;
            dx[r] = g0 * g1;
            dx[r + half] = g0 * g2;
        }
    }
    if (!BWD) {
        const double a = block_sum_d(ce, red);
        const double b = block_sum_d(js, red);
        if (threadIdx.x == 0) { part[2 * blockIdx.x] = a; part[2 * blockIdx.x + 1] = b; }
    }
}

// ---------------------------------------------------------------- softmax (RoI) rows, one wave per pair
constexpr int MAXC_PER_LANE = 4;  // C <= 256

template <bool BWD>
__global__ void sm_kernel(const float* __restrict__ x, const int64_t* __restrict__ labels,
                          const float* __restrict__ weights, long half, int C, float k_ce, float k_jsd,
                          const float* __restrict__ gout, double* __restrict__ part,
                          float* __restrict__ dx) {
    __shared__ double red[16];
    const int lane = threadIdx.x & 63;
    const long wid = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long nw = (long)gridDim.x * (blockDim.x >> 6);
    const float g0 = BWD ? (gout ? gout[0] : 1.0f) : 0.f;
    double ce = 0.0, js = 0.0;
    for (long r = wid; r < half; r += nw) {
        const float* r1 = x + r * C;
        const float* r2 = x + (r + half) * C;
        float a1[MAXC_PER_LANE], a2[MAXC_PER_LANE];
        float m1 = -INFINITY, m2 = -INFINITY;
#pragma unroll
        for (int k = 0; k < MAXC_PER_LANE; ++k) {
            const int c = lane + 64 * k;
            a1[k] = c < C ? r1[c] : -INFINITY;
            a2[k] = c < C ? r2[c] : -INFINITY;
            m1 = fmaxf(m1, a1[k]);
            m2 = fmaxf(m2, a2[k]);
        }
        m1 = wave_max(m1);
        m2 = wave_max(m2);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < MAXC_PER_LANE; ++k) {
            const int c = lane + 64 * k;
            a1[k] = c < C ? expf(a1[k] - m1) : 0.f;
            a2[k] = c < C ? expf(a2[k] - m2) : 0.f;
            s1 += a1[k];
            s2 += a2[k];
        }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        const int64_t lab = labels[r];
        const bool valid = lab != IGNORE_INDEX && lab >= 0 && lab < C;
        const float w = valid ? (weights ? weights[r] : 1.f) : 0.f;
        float jrow = 0.f, dot1 = 0.f, dot2 = 0.f, plab = 0.f;
        float gd1[MAXC_PER_LANE], gd2[MAXC_PER_LANE];
#pragma unroll
        for (int k = 0; k < MAXC_PER_LANE; ++k) {
            const int c = lane + 64 * k;
            gd1[k] = gd2[k] = 0.f;
            if (c < C) {
                const float p1 = a1[k] / s1, p2 = a2[k] / s2;
                a1[k] = p1;
                a2[k] = p2;
                const float mr = (p1 + p2) / 2.0f;
                const float m = fminf(fmaxf(mr, 1e-7f), 1.0f);
                const float lm = logf(m);
                if (c == (int)lab) plab = p1;
                if (!BWD) {
                    jrow += (xlogy_term(p1, lm) + xlogy_term(p2, lm)) / 2.0f;
                } else {
                    gd1[k] = jsd_dterm(p1, p2, mr, m, lm);
                    gd2[k] = jsd_dterm(p2, p1, mr, m, lm);
                    dot1 += p1 * gd1[k];
                    dot2 += p2 * gd2[k];
                }
            }
        }
        if (!BWD) {
            jrow = wave_sum(jrow);
            plab = wave_sum(plab);  // exactly one lane holds it
            if (lane == 0) {
                js += (double)jrow;
                // F.cross_entropy = -log_softmax[label]; log p = (x - m) - log s
                if (valid) ce += (double)(w * -logf(plab));
            }
        } else {
            dot1 = wave_sum(dot1);
            dot2 = wave_sum(dot2);
#pragma unroll
            for (int k = 0; k < MAXC_PER_LANE; ++k) {
                const int c = lane + 64 * k;
                if (c < C) {
                    const float p1 = a1[k], p2 = a2[k];
                    const float onehot = (valid && c == (int)lab) ? 1.f : 0.f;
                    const float g1 = k_ce * w * (p1 - onehot) + k_jsd * p1 * (gd1[k] - dot1);
                    const float g2 = k_jsd * p2 * (gd2[k] - dot2);
                    dx[r * C + c] = g0 * g1;
                    dx[(r + half) * C + c] = g0 * g2;
                }
            }
        }
    }
    if (!BWD) {
        const double a = block_sum_d(ce, red);
        const double b = block_sum_d(js, red);
        if (threadIdx.x == 0) { part[2 * blockIdx.x] = a; part[2 * blockIdx.x + 1] = b; }
    }
}

__global__ void cls_fin_kernel(const double* __restrict__ part, int nblocks, float k_ce, float k_jsd,
                               float* __restrict__ out) {
    __shared__ double red[16];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) { a += part[2 * i]; b += part[2 * i + 1]; }
    a = block_sum_d(a, red);
    b = block_sum_d(b, red);
    if (threadIdx.x == 0) {
        const float ce = (float)a * k_ce, js = (float)b * k_jsd;
        out[0] = ce + js;  // CrossEntropyLossPlus: loss_cls + lambda * additional
        out[1] = ce;
        out[2] = js;
    }
}

// ================================================================================================ fused RPN loss
// AnchorHead.loss (mmdet/models/dense_heads/anchor_head.py:402-544) for ALL pyramid levels in one forward and one backward
// launch, reading the RPN head's output where it lies: the fused cls+reg head writes one channel-padded NHWC map per level
// (channels [0, A) = objectness logits, [A, 5A) = box deltas, rest padding).  Per level the reference permutes / casts /
// reshapes both outputs and the four target tensors, runs CrossEntropyLossPlus (BCE on the view-1 rows + JSD between the
// views, cross_entropy_loss_plus.py:82-130,264-319) and L1LossPlus (view-1 rows, smooth_l1_loss_plus.py), and autograd
// walks all of that back: ~75 launches forward and ~50 backward per step.  The sum over the levels is what the detector
// logs (base.py:234-277 sums the per-level list), so one fp64 accumulation over all levels replaces five fp32 ones.
// Rows: anchor j = level offset + (h*W + w)*A + a of image i pairs with the same anchor of image i + B/2 (view 2).
struct RpnLossLevels {
    oadg_rpn_loss_level l[8];
    int n;
    long pixels;            // sum of H*W
};

__device__ __forceinline__ float ld_map(const void* p, long off, int dtype) {
    if (dtype == 0) return reinterpret_cast<const float*>(p)[off];
    return __builtin_bit_cast(float, (unsigned)reinterpret_cast<const unsigned short*>(p)[off] << 16);
}

__device__ __forceinline__ int level_of_pixel(const RpnLossLevels& lv, long q) {
    int li = 0;
    while (li + 1 < lv.n && q >= lv.l[li + 1].pix0) ++li;
    return li;
}

// AF > 0: the number of anchors per pixel is the compile-time constant AF (5*AF <= 16) and the map is channel-contiguous
// bf16 with 16-byte aligned rows - the 16 leading channels of both views arrive as two 16-byte loads each and the box
// weights / targets of an anchor as one float4 (the generic form issues 2-byte loads 2*Cy bytes apart: latency-bound).
// Same expressions in the same order: bit-identical results.
__device__ __forceinline__ void ld_row16(const void* y, long off, float* x) {
    const bf16x8* p = reinterpret_cast<const bf16x8*>(reinterpret_cast<const unsigned short*>(y) + off);
    const bf16x8 lo = p[0], hi = p[1];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        x[e] = __builtin_bit_cast(float, (unsigned)(unsigned short)lo[e] << 16);
        x[8 + e] = __builtin_bit_cast(float, (unsigned)(unsigned short)hi[e] << 16);
    }
}

template <int AF>
__global__ __launch_bounds__(256) void rpn_loss_fwd_kernel(RpnLossLevels lv, int B, int A_rt, long At, int dtype,
                                                           const int64_t* __restrict__ labels,
                                                           const float* __restrict__ label_w,
                                                           const float* __restrict__ bbox_t,
                                                           const float* __restrict__ bbox_w, double* __restrict__ part) {
    constexpr bool FAST = AF > 0;
    const int A = FAST ? AF : A_rt;
    __shared__ double red[16];
    const long total = (long)(B / 2) * lv.pixels;
    double ce = 0.0, js = 0.0, l1 = 0.0;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int img = (int)(idx / lv.pixels);
        const long q = idx - (long)img * lv.pixels;
        const int li = level_of_pixel(lv, q);
        const oadg_rpn_loss_level L = lv.l[li];
        const int pix = (int)(q - L.pix0), w = pix % L.W, h = pix / L.W;
        const long o1 = (long)img * L.sN + (long)h * L.sH + (long)w * L.sW;
        const long o2 = o1 + (long)(B / 2) * L.sN;
        const long j0 = (long)L.first + (long)pix * A;
        float xa[16], xb[16];
        if (FAST) { ld_row16(L.y, o1, xa); ld_row16(L.y, o2, xb); }
        for (int a = 0; a < A; ++a) {       // (constant trip count in the FAST form: unrolled, xa / xb stay in registers)
            const long row = (long)img * At + j0 + a;
            const float x1 = FAST ? xa[a] : ld_map(L.y, o1 + (long)a * L.sC, dtype);
            const float x2 = FAST ? xb[a] : ld_map(L.y, o2 + (long)a * L.sC, dtype);
            const int64_t lab = labels[row];
            const bool valid = lab >= 0 && lab != IGNORE_INDEX;
            const float t = (valid && lab == 0) ? 1.0f : 0.0f;
            const float wv = valid ? label_w[row] : 0.f;
            const float p1 = sigmoidf_(x1), q1 = 1.0f - p1, p2 = sigmoidf_(x2), q2 = 1.0f - p2;
            const float mp = fminf(fmaxf((p1 + p2) / 2.0f, 1e-7f), 1.0f), mq = fminf(fmaxf((q1 + q2) / 2.0f, 1e-7f), 1.0f);
            const float lmp = logf(mp), lmq = logf(mq);
            ce += (double)(wv * bce_logits(x1, t));
            js += (double)(((xlogy_term(p1, lmp) + xlogy_term(q1, lmq)) + (xlogy_term(p2, lmp) + xlogy_term(q2, lmq))) / 2.0f);
            if (FAST) {
                const float4 bw4 = reinterpret_cast<const float4*>(bbox_w)[row];
                if (bw4.x != 0.f || bw4.y != 0.f || bw4.z != 0.f || bw4.w != 0.f) {
                    const float4 bt4 = reinterpret_cast<const float4*>(bbox_t)[row];
                    const float bwv[4] = {bw4.x, bw4.y, bw4.z, bw4.w}, btv[4] = {bt4.x, bt4.y, bt4.z, bt4.w};
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (bwv[c] != 0.f) l1 += (double)(fabsf(xa[(FAST ? AF : 0) + a * 4 + c] - btv[c]) * bwv[c]);
                }
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float bw = bbox_w[row * 4 + c];
                    if (bw != 0.f) {
                        const float pr = ld_map(L.y, o1 + (long)(A + a * 4 + c) * L.sC, dtype);
                        l1 += (double)(fabsf(pr - bbox_t[row * 4 + c]) * bw);
                    }
                }
            }
        }
    }
    const double s0 = block_sum_d(ce, red), s1 = block_sum_d(js, red), s2 = block_sum_d(l1, red);
    if (threadIdx.x == 0) { part[3 * blockIdx.x] = s0; part[3 * blockIdx.x + 1] = s1; part[3 * blockIdx.x + 2] = s2; }
}

__global__ void rpn_loss_fin_kernel(const double* __restrict__ part, int nblocks, float k_ce, float k_jsd, float k_l1,
                                    float* __restrict__ out) {
    __shared__ double red[16];
    double a = 0.0, b = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x) { a += part[3 * i]; b += part[3 * i + 1]; c += part[3 * i + 2]; }
    a = block_sum_d(a, red); b = block_sum_d(b, red); c = block_sum_d(c, red);
    if (threadIdx.x == 0) {
        const float ce = (float)a * k_ce, js = (float)b * k_jsd;
        out[0] = ce + js; out[1] = ce; out[2] = js; out[3] = (float)c * k_l1;
    }
}

// one thread per (image, pixel) computes the 5A live channels of its gradient row; the workgroup then writes the Cy-channel
// rows (bf16 / fp32 like y, padding zeroed) cooperatively in 16-byte pieces, consecutive lanes on consecutive pieces: a
// thread writing its own 256-byte row put every store instruction on 64 different cache lines (0.42 ms for 358 MB)
template <bool COOP, int AF>
__global__ __launch_bounds__(256) void rpn_loss_bwd_kernel(RpnLossLevels lv, int B, int A_rt, long At, int dtype,
                                                           const int64_t* __restrict__ labels,
                                                           const float* __restrict__ label_w,
                                                           const float* __restrict__ bbox_t,
                                                           const float* __restrict__ bbox_w, float k_ce, float k_jsd,
                                                           float k_l1, const float* __restrict__ g_cls,
                                                           const float* __restrict__ g_box) {
    extern __shared__ unsigned char smem_bwd[];
    constexpr bool FAST = AF > 0;                                     // see rpn_loss_fwd_kernel
    static_assert(!FAST || COOP, "the vector-load form always stores cooperatively");
    const int A = FAST ? AF : A_rt;
    const int esz = dtype == 0 ? 4 : 2;
    const int NZ = FAST ? 16 : (5 * A + 7) & ~7;                       // live channels, whole 8-channel chunks
    unsigned char** rowptr = reinterpret_cast<unsigned char**>(smem_bwd);          // [256]
    unsigned char* vals = smem_bwd + 256 * sizeof(unsigned char*);                 // [256][NZ] elements
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = idx < (long)B * lv.pixels;
    if (!COOP && !live) return;
    int Cy_blk = 0;
    if (live) {
    const int img = (int)(idx / lv.pixels);
    const long q = idx - (long)img * lv.pixels;
    const int li = level_of_pixel(lv, q);
    const oadg_rpn_loss_level L = lv.l[li];
    const int pix = (int)(q - L.pix0), w = pix % L.W, h = pix / L.W;
    const bool v1 = img < B / 2;
    const int i1 = v1 ? img : img - B / 2;                    // the view-1 image of the pair (labels / targets live there)
    const long o1 = (long)i1 * L.sN + (long)h * L.sH + (long)w * L.sW, o2 = o1 + (long)(B / 2) * L.sN;
    const long j0 = (long)L.first + (long)pix * A;
    const float gc = g_cls ? g_cls[0] : 1.0f, gb = g_box ? g_box[0] : 1.0f;
    const size_t grow = (((size_t)img * L.H + h) * L.W + w) * L.Cy;      // gy: dense NHWC [N, H, W, Cy]
    Cy_blk = L.Cy;
    if (COOP) rowptr[threadIdx.x] = reinterpret_cast<unsigned char*>(L.gy) + grow * esz;
    const int cend = COOP ? NZ : L.Cy;
    float xa[16], xb[16];
    if (FAST) { ld_row16(L.y, o1, xa); ld_row16(L.y, o2, xb); }
    for (int c8 = 0; c8 < cend; c8 += 8) {  // (two iterations in the FAST form: unrolled, xa / xb stay in registers)
        float g[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ch = c8 + e;
            float v = 0.f;
            if (ch < A) {
                const long row = (long)i1 * At + j0 + ch;
                const float x1 = FAST ? xa[ch & 15] : ld_map(L.y, o1 + (long)ch * L.sC, dtype);
                const float x2 = FAST ? xb[ch & 15] : ld_map(L.y, o2 + (long)ch * L.sC, dtype);
                const float p1 = sigmoidf_(x1), q1 = 1.0f - p1, p2 = sigmoidf_(x2), q2 = 1.0f - p2;
                const float mpr = (p1 + p2) / 2.0f, mqr = (q1 + q2) / 2.0f;
                const float mp = fminf(fmaxf(mpr, 1e-7f), 1.0f), mq = fminf(fmaxf(mqr, 1e-7f), 1.0f);
                const float lmp = logf(mp), lmq = logf(mq);
                if (v1) {
                    const int64_t lab = labels[row];
                    const bool valid = lab >= 0 && lab != IGNORE_INDEX;
                    const float t = (valid && lab == 0) ? 1.0f : 0.0f;
                    const float wv = valid ? label_w[row] : 0.f;
                    const float d1 = jsd_dterm(p1, p2, mpr, mp, lmp) - jsd_dterm(q1, q2, mqr, mq, lmq);
                    v = gc * (k_ce * wv * (p1 - t) + k_jsd * p1 * (1.0f - p1) * d1);
                } else {
                    const float d2 = jsd_dterm(p2, p1, mpr, mp, lmp) - jsd_dterm(q2, q1, mqr, mq, lmq);
                    v = gc * (k_jsd * p2 * (1.0f - p2) * d2);
                }
            } else if (ch < 5 * A && v1) {
                const int a = (ch - A) >> 2, c = (ch - A) & 3;
                const long row = (long)i1 * At + j0 + a;
                const float bw = bbox_w[row * 4 + c];
                if (bw != 0.f) {
                    const float d = (FAST ? xa[ch & 15] : ld_map(L.y, o1 + (long)ch * L.sC, dtype)) - bbox_t[row * 4 + c];
                    v = gb * k_l1 * bw * (d > 0.f ? 1.0f : (d < 0.f ? -1.0f : 0.f));
                }
            }
            g[e] = v;
        }
        if (dtype == 0) {
            float* o = COOP ? reinterpret_cast<float*>(vals) + (size_t)threadIdx.x * NZ + c8
                            : reinterpret_cast<float*>(L.gy) + grow + c8;
            *reinterpret_cast<f32x4*>(o) = f32x4{g[0], g[1], g[2], g[3]};
            *reinterpret_cast<f32x4*>(o + 4) = f32x4{g[4], g[5], g[6], g[7]};
        } else {
            bf16x8 v8;
#pragma unroll
            for (int e = 0; e < 8; ++e) v8[e] = (short)f32_to_bf16(g[e]);
            unsigned short* o = COOP ? reinterpret_cast<unsigned short*>(vals) + (size_t)threadIdx.x * NZ + c8
                                     : reinterpret_cast<unsigned short*>(L.gy) + grow + c8;
            *reinterpret_cast<bf16x8*>(o) = v8;
        }
    }
    } else if (COOP) {
        rowptr[threadIdx.x] = nullptr;
    }
    if (!COOP) return;
    // every level has the same Cy (checked by the launcher for this form): P 16-byte pieces per row
    __shared__ int cy_sh;
    if (threadIdx.x == 0) cy_sh = Cy_blk;                              // thread 0 of a launched block is always live
    __syncthreads();
    const int P = cy_sh * esz / 16, liveP = NZ * esz / 16;
    for (int k = threadIdx.x; k < 256 * P; k += 256) {
        const int px = k / P, piece = k - px * P;
        unsigned char* rp = rowptr[px];
        if (!rp) continue;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (piece < liveP) v = *reinterpret_cast<const f32x4*>(vals + ((size_t)px * NZ * esz) + (size_t)piece * 16);
        *reinterpret_cast<f32x4*>(rp + (size_t)piece * 16) = v;
    }
}

// the vector-load kernels: 3 anchors per pixel, bf16, channels contiguous, every pixel row 16-byte aligned
bool rpn_loss_fast3(const RpnLossLevels& lv, int A, int dtype) {
    if (A != 3 || dtype != 1) return false;
    for (int i = 0; i < lv.n; ++i) {
        const oadg_rpn_loss_level& l = lv.l[i];
        if (l.sC != 1 || l.Cy < 16 || (l.sN & 7) || (l.sH & 7) || (l.sW & 7) || ((uintptr_t)l.y & 15)) return false;
    }
    return true;
}

int fill_rpn_loss(RpnLossLevels& lv, const oadg_rpn_loss_level* levels, int n_levels, int A, bool need_gy) {
    if (!levels || n_levels < 1 || n_levels > 8 || A < 1) return OADG_EARG;
    lv.n = n_levels;
    long pix = 0, first = 0;
    for (int i = 0; i < n_levels; ++i) {
        lv.l[i] = levels[i];
        if (!levels[i].y || (need_gy && !levels[i].gy) || levels[i].H < 1 || levels[i].W < 1 || levels[i].Cy < 5 * A ||
            (levels[i].Cy & 7) || levels[i].pix0 != pix || levels[i].first != first)
            return OADG_EARG;
        pix += (long)levels[i].H * levels[i].W;
        first += (long)levels[i].H * levels[i].W * A;
    }
    lv.pixels = pix;
    return OADG_OK;
}

int grid_for(long items, int per_block) {
    long g = (items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > MAXBLOCKS) g = MAXBLOCKS;
    return (int)g;
}


// ================================================================================================ RoI head: box loss + accuracy
// BBoxHead.loss's regression term and its logged accuracy (mmdet/models/roi_heads/bbox_heads/bbox_head.py:397-460,
// mmdet/models/losses/accuracy.py) in one forward and one backward launch.  The reference gathers the positive rows
// (pos_inds = labels in [0, C)), picks each row's 4 deltas of ITS class (bbox_pred.view(K, -1, 4)[pos, labels[pos]]),
// gathers targets and weights likewise and runs L1 / SmoothL1 (the fork's ...LossPlus: on the leading chunk of the
// positives = view 1 - `reg_limit` = one past the last row that takes part): ~17 launches forward, ~28 backward (index
// put with a sort).  Here: thread r owns row r; the loss is accumulated in fp64 in a fixed order (deterministic), the
// backward writes the whole [K, 4 C] gradient (zeros + the positives' 4 entries) in bbox_pred's dtype.
__device__ __forceinline__ float ld_any(const void* p, long i, int dtype) {
    return dtype == 0 ? reinterpret_cast<const float*>(p)[i] : bf16_to_f32(reinterpret_cast<const unsigned short*>(p)[i]);
}

__global__ __launch_bounds__(1024) void roi_reg_acc_fwd_kernel(const void* __restrict__ bbox_pred, int pred_dtype,
                                                               const void* __restrict__ cls, int cls_dtype,
                                                               const int64_t* __restrict__ labels,
                                                               const float* __restrict__ targets,
                                                               const float* __restrict__ weights, int K, int C, int n_reg,
                                                               int n_cls, int reg_limit, float beta, float avg_factor,
                                                               float loss_weight, float* __restrict__ out) {
    __shared__ double red[16];
    double sum = 0.0, correct = 0.0;
    for (int r = threadIdx.x; r < K; r += blockDim.x) {
        const int64_t l = labels[r];
        if (cls) {                                     // top-1 of the row (first maximum), compared with the label
            float best = ld_any(cls, (long)r * n_cls, cls_dtype);
            int arg = 0;
            for (int c = 1; c < n_cls; ++c) {
                const float v = ld_any(cls, (long)r * n_cls + c, cls_dtype);
                if (v > best) { best = v; arg = c; }
            }
            correct += (arg == (int)l) ? 1.0 : 0.0;
        }
        if (r < reg_limit && l >= 0 && l < C) {
            const long col = n_reg == 4 ? 0 : (long)l * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float pv = ld_any(bbox_pred, (long)r * n_reg + col + c, pred_dtype);
                const float d = fabsf(pv - targets[(long)r * 4 + c]);
                const float e = beta > 0.f ? (d < beta ? 0.5f * d * d / beta : d - 0.5f * beta) : d;
                sum += (double)(e * weights[(long)r * 4 + c]);
            }
        }
    }
    const double s = block_sum_d(sum, red), k = block_sum_d(correct, red);
    if (threadIdx.x == 0) {
        out[0] = loss_weight * ((float)s / avg_factor);
        out[1] = (float)k * (100.0f / (float)(K > 0 ? K : 1));
    }
}

__global__ __launch_bounds__(256) void roi_reg_bwd_kernel(const void* __restrict__ bbox_pred, int pred_dtype,
                                                          const int64_t* __restrict__ labels,
                                                          const float* __restrict__ targets,
                                                          const float* __restrict__ weights, int K, int C, int n_reg,
                                                          int reg_limit, float beta, float avg_factor, float loss_weight,
                                                          const float* __restrict__ gout, void* __restrict__ grad) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;            // one 4-element group of the [K, n_reg] gradient
    const int groups = n_reg / 4;
    if (i >= (long)K * groups) return;
    const int r = (int)(i / groups), gcol = (int)(i - (long)r * groups);
    const int64_t l = labels[r];
    float g[4] = {0.f, 0.f, 0.f, 0.f};
    if (r < reg_limit && l >= 0 && l < C && (n_reg == 4 || gcol == (int)l)) {
        const float k = (gout[0] * loss_weight) / avg_factor;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float d = ld_any(bbox_pred, (long)r * n_reg + gcol * 4 + c, pred_dtype) - targets[(long)r * 4 + c];
            const float ad = fabsf(d);
            const float de = (beta > 0.f && ad < beta) ? d / beta : (d > 0.f ? 1.0f : (d < 0.f ? -1.0f : 0.f));
            g[c] = k * weights[(long)r * 4 + c] * de;
        }
    }
    if (pred_dtype == 0) {
        reinterpret_cast<f32x4*>(grad)[i] = f32x4{g[0], g[1], g[2], g[3]};
    } else {
        unsigned short* o = reinterpret_cast<unsigned short*>(grad) + i * 4;
        const unsigned lo = (unsigned)f32_to_bf16(g[0]) | ((unsigned)f32_to_bf16(g[1]) << 16);
        const unsigned hi = (unsigned)f32_to_bf16(g[2]) | ((unsigned)f32_to_bf16(g[3]) << 16);
        reinterpret_cast<uint2*>(o)[0] = make_uint2(lo, hi);
    }
}


// ================================================================================================ _parse_losses
// BaseDetector._parse_losses (mmdet/models/detectors/base.py:234-277) for scalar loss values: log_vars[name] = the sum of
// its entries (a value or a per-level list, python's left-to-right float adds starting from 0), loss = the sum of the
// variables whose name contains 'loss', in dictionary order; packed = [log_vars..., loss].  One thread: ~10 numbers.
struct ParseLossArgs {
    const float* v[OADG_PARSE_LOSSES_MAX];
    int name_of[OADG_PARSE_LOSSES_MAX];
    int n, n_names;
    unsigned is_loss;          // bit i: name i takes part in the total
};

__global__ void parse_losses_kernel(const ParseLossArgs a, float* __restrict__ out, float* __restrict__ total_out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float total = 0.f;
    for (int name = 0; name < a.n_names; ++name) {
        float s = 0.f;
        bool first = true;
#pragma unroll
        for (int i = 0; i < OADG_PARSE_LOSSES_MAX; ++i)
            if (i < a.n && a.name_of[i] == name) {
                // a single tensor is its own mean; a list is python's sum(): 0 + v0 + v1 + ...
                s = first ? (0.f + a.v[i][0]) : (s + a.v[i][0]);
                first = false;
            }
        out[name] = s;
        if ((a.is_loss >> name) & 1u) total = total + s;
    }
    out[a.n_names] = total;
    if (total_out) total_out[0] = total;
}

}  // namespace

extern "C" {

size_t oadg_cls_loss_workspace_bytes(void) { return (size_t)MAXBLOCKS * 2 * sizeof(double); }

// mode: 0 = sigmoid rows (C must be 1), 1 = softmax rows
int oadg_ce_jsd_fwd(const float* logits, const int64_t* labels, const float* weights, long R, int C,
                    int mode, float avg_factor, float loss_weight, float lambda_jsd, void* workspace,
                    size_t workspace_bytes, float* out3, void* stream) {
    if (!logits || !labels || !workspace || !out3) return OADG_EARG;
    if (R < 0 || (R & 1) || C < 1 || C > 64 * MAXC_PER_LANE || !(avg_factor > 0.f)) return OADG_EARG;
    if (mode == 0 && C != 1) return OADG_EARG;
    if (workspace_bytes < oadg_cls_loss_workspace_bytes()) return OADG_ESIZE;
    hipStream_t st = (hipStream_t)stream;
    const long half = R / 2;
    double* part = (double*)workspace;
    const float k_ce = loss_weight / avg_factor, k_jsd = lambda_jsd / avg_factor;
    int g;
    if (mode == 0) {
        g = grid_for(half, 256 * 4);
        hipLaunchKernelGGL((sig_kernel<false>), dim3(g), dim3(256), 0, st, logits, labels, weights, half,
                           k_ce, k_jsd, (const float*)nullptr, part, (float*)nullptr);
    } else {
        g = grid_for(half, 4);
        hipLaunchKernelGGL((sm_kernel<false>), dim3(g), dim3(256), 0, st, logits, labels, weights, half, C,
                           k_ce, k_jsd, (const float*)nullptr, part, (float*)nullptr);
    }
    OADG_LAUNCH_CHECK();
    hipLaunchKernelGGL(cls_fin_kernel, dim3(1), dim3(256), 0, st, (const double*)part, g, k_ce, k_jsd, out3);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

int oadg_ce_jsd_bwd(const float* logits, const int64_t* labels, const float* weights, long R, int C,
                    int mode, float avg_factor, float loss_weight, float lambda_jsd, const float* grad_out,
                    float* dlogits, void* stream) {
    if (!logits || !labels || !dlogits) return OADG_EARG;
    if (R < 0 || (R & 1) || C < 1 || C > 64 * MAXC_PER_LANE || !(avg_factor > 0.f)) return OADG_EARG;
    if (mode == 0 && C != 1) return OADG_EARG;
    hipStream_t st = (hipStream_t)stream;
    const long half = R / 2;
    if (half == 0) return OADG_OK;
    const float k_ce = loss_weight / avg_factor, k_jsd = lambda_jsd / avg_factor;
    if (mode == 0) {
        hipLaunchKernelGGL((sig_kernel<true>), dim3(grid_for(half, 256 * 4)), dim3(256), 0, st, logits,
                           labels, weights, half, k_ce, k_jsd, grad_out, (double*)nullptr, dlogits);
    } else {
        hipLaunchKernelGGL((sm_kernel<true>), dim3(grid_for(half, 4)), dim3(256), 0, st, logits, labels,
                           weights, half, C, k_ce, k_jsd, grad_out, (double*)nullptr, dlogits);
    }
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

// Fused RPN loss (see above).  levels: per pyramid level the head output y (dtype 0 fp32 / 1 bf16; logical [B, Cy, H, W]
// with element strides sN, sC, sH, sW; channels [0, A) logits, [A, 5A) deltas), its gradient map gy (dense NHWC
// [B, H, W, Cy], backward only), first = anchors before the level, pix0 = pixels before it.  labels / label_weights
// [B, At], bbox_targets / bbox_weights [B, At, 4] (only the view-1 half is read), At = anchors per image.  out4 = {loss_cls
// (= BCE + lambda JSD), BCE part, JSD part, loss_bbox}, each already multiplied by its weight and divided by avg_factor.
size_t oadg_rpn_loss_workspace_bytes(void) { return (size_t)MAXBLOCKS * 3 * sizeof(double); }

int oadg_rpn_loss_fwd(const oadg_rpn_loss_level* levels, int n_levels, int B, int A, long At, int dtype,
                      const int64_t* labels, const float* label_weights, const float* bbox_targets,
                      const float* bbox_weights, float avg_factor, float w_cls, float lambda_jsd, float w_box,
                      void* workspace, size_t workspace_bytes, float* out4, void* stream) {
    RpnLossLevels lv;
    const int rc = fill_rpn_loss(lv, levels, n_levels, A, false);
    if (rc) return rc;
    if (B < 2 || (B & 1) || !labels || !label_weights || !bbox_targets || !bbox_weights || !workspace || !out4 ||
        !(avg_factor > 0.f) || (dtype != 0 && dtype != 1))
        return OADG_EARG;
    if (workspace_bytes < oadg_rpn_loss_workspace_bytes()) return OADG_ESIZE;
    hipStream_t st = (hipStream_t)stream;
    const int g = grid_for((long)(B / 2) * lv.pixels, 256 * 2);
    if (rpn_loss_fast3(lv, A, dtype) && ((uintptr_t)bbox_targets & 15) == 0 && ((uintptr_t)bbox_weights & 15) == 0)
        hipLaunchKernelGGL(rpn_loss_fwd_kernel<3>, dim3(g), dim3(256), 0, st, lv, B, A, At, dtype, labels, label_weights,
                           bbox_targets, bbox_weights, (double*)workspace);
    else
        hipLaunchKernelGGL(rpn_loss_fwd_kernel<0>, dim3(g), dim3(256), 0, st, lv, B, A, At, dtype, labels, label_weights,
                           bbox_targets, bbox_weights, (double*)workspace);
    OADG_LAUNCH_CHECK();
    hipLaunchKernelGGL(rpn_loss_fin_kernel, dim3(1), dim3(256), 0, st, (const double*)workspace, g, w_cls / avg_factor,
                       lambda_jsd / avg_factor, w_box / avg_factor, out4);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

int oadg_rpn_loss_bwd(const oadg_rpn_loss_level* levels, int n_levels, int B, int A, long At, int dtype,
                      const int64_t* labels, const float* label_weights, const float* bbox_targets,
                      const float* bbox_weights, float avg_factor, float w_cls, float lambda_jsd, float w_box,
                      const float* grad_cls, const float* grad_box, void* stream) {
    RpnLossLevels lv;
    const int rc = fill_rpn_loss(lv, levels, n_levels, A, true);
    if (rc) return rc;
    if (B < 2 || (B & 1) || !labels || !label_weights || !bbox_targets || !bbox_weights || !(avg_factor > 0.f) ||
        (dtype != 0 && dtype != 1))
        return OADG_EARG;
    const long total = (long)B * lv.pixels;
    // cooperative row stores need one Cy for all levels and the live channels of 256 pixels in LDS
    const int esz = dtype == 0 ? 4 : 2, NZ = (5 * A + 7) & ~7;
    bool coop = (size_t)256 * NZ * esz <= 48 * 1024;
    for (int i = 1; i < lv.n; ++i) coop = coop && lv.l[i].Cy == lv.l[0].Cy;
    if (coop && rpn_loss_fast3(lv, A, dtype) && ((uintptr_t)bbox_targets & 15) == 0 && ((uintptr_t)bbox_weights & 15) == 0) {
        const size_t smem = 256 * sizeof(void*) + (size_t)256 * 16 * esz;
        hipLaunchKernelGGL((rpn_loss_bwd_kernel<true, 3>), dim3((unsigned)((total + 255) / 256)), dim3(256), smem,
                           (hipStream_t)stream, lv, B, A, At, dtype, labels, label_weights, bbox_targets, bbox_weights,
                           w_cls / avg_factor, lambda_jsd / avg_factor, w_box / avg_factor, grad_cls, grad_box);
    } else if (coop) {
        const size_t smem = 256 * sizeof(void*) + (size_t)256 * NZ * esz;
        hipLaunchKernelGGL((rpn_loss_bwd_kernel<true, 0>), dim3((unsigned)((total + 255) / 256)), dim3(256), smem,
                           (hipStream_t)stream, lv, B, A, At, dtype, labels, label_weights, bbox_targets, bbox_weights,
                           w_cls / avg_factor, lambda_jsd / avg_factor, w_box / avg_factor, grad_cls, grad_box);
    } else {
        hipLaunchKernelGGL((rpn_loss_bwd_kernel<false, 0>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                           (hipStream_t)stream, lv, B, A, At, dtype, labels, label_weights, bbox_targets, bbox_weights,
                           w_cls / avg_factor, lambda_jsd / avg_factor, w_box / avg_factor, grad_cls, grad_box);
    }
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

int oadg_roi_reg_acc_fwd(const void* bbox_pred, int pred_dtype, const void* cls_score, int cls_dtype, const int64_t* labels,
                         const float* bbox_targets, const float* bbox_weights, int K, int num_classes, int n_reg,
                         int n_cls, int reg_limit, float beta, float avg_factor, float loss_weight, float* out2,
                         void* stream) {
    if (!bbox_pred || !labels || !bbox_targets || !bbox_weights || !out2 || K < 0 || num_classes < 1 ||
        (n_reg != 4 && n_reg != 4 * num_classes) || (cls_score && n_cls < 1) || !(avg_factor > 0.f) || beta < 0.f ||
        (pred_dtype != 0 && pred_dtype != 1) || (cls_dtype != 0 && cls_dtype != 1))
        return OADG_EARG;
    hipLaunchKernelGGL(roi_reg_acc_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, bbox_pred, pred_dtype, cls_score,
                       cls_dtype, labels, bbox_targets, bbox_weights, K, num_classes, n_reg, n_cls, reg_limit, beta,
                       avg_factor, loss_weight, out2);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

int oadg_roi_reg_bwd(const void* bbox_pred, int pred_dtype, const int64_t* labels, const float* bbox_targets,
                     const float* bbox_weights, int K, int num_classes, int n_reg, int reg_limit, float beta,
                     float avg_factor, float loss_weight, const float* grad_out, void* grad_pred, void* stream) {
    if (!bbox_pred || !labels || !bbox_targets || !bbox_weights || !grad_out || !grad_pred || K < 0 || num_classes < 1 ||
        (n_reg != 4 && n_reg != 4 * num_classes) || !(avg_factor > 0.f) || beta < 0.f || (pred_dtype != 0 && pred_dtype != 1))
        return OADG_EARG;
    const long total = (long)K * (n_reg / 4);
    if (total == 0) return OADG_OK;
    hipLaunchKernelGGL(roi_reg_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, bbox_pred,
                       pred_dtype, labels, bbox_targets, bbox_weights, K, num_classes, n_reg, reg_limit, beta, avg_factor,
                       loss_weight, grad_out, grad_pred);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

int oadg_parse_losses(const float* const* values_host, const int* name_of_host, int n, int n_names, unsigned is_loss_mask,
                      float* packed, float* total_out, void* stream) {
    if (!values_host || !name_of_host || !packed || n < 1 || n > OADG_PARSE_LOSSES_MAX || n_names < 1 || n_names > 31)
        return OADG_EARG;
    ParseLossArgs a;
    std::memset(&a, 0, sizeof(a));
    for (int i = 0; i < n; ++i) {
        if (!values_host[i] || name_of_host[i] < 0 || name_of_host[i] >= n_names) return OADG_EARG;
        a.v[i] = values_host[i];
        a.name_of[i] = name_of_host[i];
    }
    a.n = n; a.n_names = n_names; a.is_loss = is_loss_mask;
    hipLaunchKernelGGL(parse_losses_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, packed, total_out);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

}  // extern "C"
