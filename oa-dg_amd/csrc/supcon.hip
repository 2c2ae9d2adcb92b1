// OA-Loss instance-level supervised-contrastive loss (forward + backward) for gfx950.
//
// Replaces, behind the C ABI in include/oadg_hip.h:
//   mmdet/models/losses/oadg/contrastive_loss_plus.py:31-50   (ContrastiveLossPlus.forward)
//   mmdet/models/losses/oadg/contrastive_loss.py:170-232      (supcontrast: mask construction)
//   mmdet/models/losses/oadg/contrastive_loss.py:147-167      (supcontrast_mask: NT-Xent reduction)
//
// Math.  F^ = normalize(F);  S = F^ F^T / tau;  shift m = 1/tau (= row max, S_ii);
//   E_ij = exp(S_ij - m) for j != i;  Z_i = sum_j E_ij
//   P_ij = [i!=j] ( [l_i == l_j != bg]  +  [j == twin(i)] [l_i == l_j == bg] )
//   loss = -1/B sum_i ( sum_j P_ij (S_ij - m) - n_i log Z_i ) / (n_i + 1e-8),   n_i = sum_j P_ij
// The B x B matrices of the reference (seven 17 MB fp32 masks at B = 2082) are never materialised:
// the similarity tile lives in MFMA accumulators and masks are recomputed from labels/indices.
//
// Kernel structure (wave64, f32-input MFMA 32x32x2 = exact f32 FMA chains):
//   * one wave owns a 32-row tile X of F^ (B-operand fragments stay in VGPRs for the whole kernel)
//   * the 4 waves of a block share each streamed 32-row tile Y through LDS (row stride D+2 floats:
//     conflict-free ds_read_b64 for the A operand, conflict-free ds_read_b32 rows for the 2nd GEMM)
//   * tile S[y][x] lands in the accumulator with lane <-> x, so every row statistic of X is a
//     lane-local sum over accumulator registers (no cross-lane reduction besides one half swap)
//   * backward feeds the accumulator registers straight back as the A operand of the second GEMM
//     dF^_X += W[y][x] * F^_Y  (C-layout == A-layout under lane<->x), so no LDS transpose.
//   * Y is split over blockIdx.y; partials are reduced by a finalize kernel (deterministic, no atomics
//     on floats).
#include "common.h"

namespace {

constexpr int TILE = 32;
constexpr int WAVES = 4;
constexpr int MAX_LABELS = 1024;

// workspace header (ints), followed by float arrays; see oadg_supcon_workspace_bytes
struct WsHeader {
    int bg_label;      // max label
    int error;         // bit0: label out of range
    int skip;          // 1 when #fg <= min_samples -> loss 0, grads 0
    int pad;
    int cnt[MAX_LABELS];
};

__device__ __forceinline__ int label_of(const int64_t* labels, int n_labels, int i) {
    // rows beyond the labelled block are the random proposals, labelled like the last row
    // (contrastive_loss_plus.py:44-47)
    int64_t v = labels[i < n_labels ? i : n_labels - 1];
    return (int)v;
}

__device__ __forceinline__ int twin_of(int i, int ori, int rp) {
    // contrastive_loss.py:203-206: [0,ori)<->[ori,2ori) ; [2ori,2ori+rp)<->[2ori+rp,2ori+2rp)
    if (i < ori) return i + ori;
    if (i < 2 * ori) return i - ori;
    const int j = i - 2 * ori;
    if (j < rp) return i + rp;
    if (j < 2 * rp) return i - rp;
    return -1;
}

// PREP_ROWS rows per workgroup (a wave takes every fourth).  The label statistics go through an LDS histogram: one global
// atomic per (workgroup, label that occurs in it) - round 3 issued two per ROW onto the same ~9 addresses, 17,000
// serialised same-address atomics = 99 us on the critical path of the loss section (integer sums / maxima: any order
// gives the same result).
constexpr int PREP_ROWS = 32;
__global__ __launch_bounds__(256) void supcon_prep_kernel(const float* __restrict__ feats,
                                                          const int64_t* __restrict__ labels, int B, int D, int n_labels,
                                                          float* __restrict__ fhat, float* __restrict__ invnorm,
                                                          WsHeader* hdr) {
    __shared__ int s_cnt[MAX_LABELS];
    __shared__ int s_max, s_err;
    for (int i = threadIdx.x; i < MAX_LABELS; i += 256) s_cnt[i] = 0;
    if (threadIdx.x == 0) { s_max = 0; s_err = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = blockIdx.x * PREP_ROWS;
    for (int rr = wave; rr < PREP_ROWS; rr += 4) {
        const int row = row0 + rr;
        if (row >= B) break;
        const float* f = feats + (size_t)row * D;
        float ss = 0.f;
        for (int d = lane; d < D; d += 64) { float v = f[d]; ss += v * v; }
        ss = wave_sum(ss);
        // F.normalize: x / max(||x||, 1e-12)
        const float nrm = fmaxf(sqrtf(ss), 1e-12f);
        const float inv = 1.0f / nrm;
        for (int d = lane; d < D; d += 64) fhat[(size_t)row * D + d] = f[d] / nrm;
        if (lane == 0) {
            invnorm[row] = inv;
            int l = label_of(labels, n_labels, row);
            if (l < 0 || l >= MAX_LABELS) { atomicOr(&s_err, 1); l = l < 0 ? 0 : MAX_LABELS - 1; }
            atomicAdd(&s_cnt[l], 1);
            atomicMax(&s_max, l);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < MAX_LABELS; i += 256)
        if (s_cnt[i]) atomicAdd(&hdr->cnt[i], s_cnt[i]);
    if (threadIdx.x == 0) {
        atomicMax(&hdr->bg_label, s_max);
        if (s_err) atomicOr(&hdr->error, 1);
    }
}

template <int D>
struct Smem {
    float y[TILE][D + 2];
    int lab[TILE];
    float ay[TILE];
    float wy[TILE];
};

// row index inside a 32x32 accumulator tile held by (register r, lane half h)
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

template <int D, bool BWD>
__global__ __launch_bounds__(256, 1) void supcon_tile_kernel(
    const float* __restrict__ fhat, const int64_t* __restrict__ labels, int B, int n_labels, int ori,
    int rp, float inv_t, const WsHeader* __restrict__ hdr, const float* __restrict__ row_a,
    const float* __restrict__ row_w, float* __restrict__ partZ, float* __restrict__ partA,
    float* __restrict__ dpart, int tiles_per_split) {
    __shared__ Smem<D> sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int xt = blockIdx.x * WAVES + wave;
    const int x = xt * TILE + c;
    const bool xok = x < B;
    const int bg = hdr->bg_label;
    const int nyt = (B + TILE - 1) / TILE;
    const int yt0 = blockIdx.y * tiles_per_split;
    const int yt1 = min(nyt, yt0 + tiles_per_split);

    // B-operand fragments of the owned tile: bx[2u+e] = F^[x][4u + 2h + e]
    float bx[D / 2];
#pragma unroll
    for (int u = 0; u < D / 4; ++u) {
        float2 v = make_float2(0.f, 0.f);
        if (xok) v = *reinterpret_cast<const float2*>(fhat + (size_t)x * D + 4 * u + 2 * h);
        bx[2 * u] = v.x;
        bx[2 * u + 1] = v.y;
    }
    const int lx = xok ? label_of(labels, n_labels, x) : -1;
    const int tx = xok ? twin_of(x, ori, rp) : -1;
    float ax = 0.f, wx = 0.f;
    if (BWD && xok) { ax = row_a[x]; wx = row_w[x]; }

    float Z = 0.f, A = 0.f;
    f32x16 dacc[BWD ? D / 32 : 1];
    if (BWD) {
#pragma unroll
        for (int i = 0; i < D / 32; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) dacc[i][r] = 0.f;
    }

    for (int yt = yt0; yt < yt1; ++yt) {
        __syncthreads();  // previous tile fully consumed
        // cooperative, coalesced load of the Y tile (float2 granules; rows >= B are zero)
        for (int idx = tid; idx < TILE * (D / 2); idx += 256) {
            const int r = idx / (D / 2), c2 = idx - r * (D / 2);
            const int y = yt * TILE + r;
            float2 v = make_float2(0.f, 0.f);
            if (y < B) v = *reinterpret_cast<const float2*>(fhat + (size_t)y * D + 2 * c2);
            *reinterpret_cast<float2*>(&sm.y[r][2 * c2]) = v;
        }
        if (tid < TILE) {
            const int y = yt * TILE + tid;
            sm.lab[tid] = y < B ? label_of(labels, n_labels, y) : -2;
            if (BWD) {
                sm.ay[tid] = y < B ? row_a[y] : 0.f;
                sm.wy[tid] = y < B ? row_w[y] : 0.f;
            }
        }
        __syncthreads();

        // S[y][x] = sum_k F^_Y[y][k] F^_X[x][k]
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int u = 0; u < D / 4; ++u) {
            const float2 a = *reinterpret_cast<const float2*>(&sm.y[c][4 * u + 2 * h]);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bx[2 * u], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bx[2 * u + 1], acc, 0, 0, 0);
        }

        f32x16 w;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int yr = acc_row(r, h);
            const int y = yt * TILE + yr;
            const bool valid = xok && (y < B) && (y != x);
            const float l = acc[r] * inv_t - inv_t;  // S - rowmax
            const float e = valid ? expf(l) : 0.f;
            const int ly = sm.lab[yr];
            const bool pos = valid && (lx == ly) && ((lx != bg) || (y == tx));
            if (!BWD) {
                Z += e;
                A += pos ? l : 0.f;
            } else {
                w[r] = e * (sm.ay[yr] + ax) - (pos ? (sm.wy[yr] + wx) : 0.f);
            }
        }
        if (BWD) {
            // dF^_X[x][d] += sum_y W[y][x] F^_Y[y][d]
#pragma unroll
            for (int cc = 0; cc < D / 32; ++cc) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float b = sm.y[acc_row(r, h)][cc * 32 + c];
                    dacc[cc] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[r], b, dacc[cc], 0, 0, 0);
                }
            }
        }
    }

    if (!BWD) {
        Z += __shfl_xor(Z, 32, 64);
        A += __shfl_xor(A, 32, 64);
        if (h == 0 && xok) {
            partZ[(size_t)blockIdx.y * B + x] = Z;
            partA[(size_t)blockIdx.y * B + x] = A;
        }
    } else {
        float* dst = dpart + (size_t)blockIdx.y * B * D;
#pragma unroll
        for (int cc = 0; cc < D / 32; ++cc) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int xr = xt * TILE + acc_row(r, h);
                if (xr < B) dst[(size_t)xr * D + cc * 32 + c] = dacc[cc][r];
            }
        }
    }
}

// one block: per-row loss, row statistics for the backward, mean over all B rows
__global__ void supcon_fin_kernel(const int64_t* __restrict__ labels, int B, int n_labels, int ori, int rp,
                                  int nsplit, int min_samples, WsHeader* hdr,
                                  const float* __restrict__ partZ, const float* __restrict__ partA,
                                  float* __restrict__ row_a, float* __restrict__ row_w,
                                  float loss_weight, float* __restrict__ out_loss) {
    __shared__ double red[16];
    const int bg = hdr->bg_label;
    const int nfg = B - hdr->cnt[bg];
    const bool skip = nfg <= min_samples;  // contrastive_loss.py:211,229-230
    double acc = 0.0;
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        float Z = 0.f, A = 0.f;
        for (int s = 0; s < nsplit; ++s) { Z += partZ[(size_t)s * B + i]; A += partA[(size_t)s * B + i]; }
        const int li = label_of(labels, n_labels, i);
        float n;
        if (li != bg) {
            n = (float)(hdr->cnt[li] - 1);
        } else {
            const int t = twin_of(i, ori, rp);
            n = (t >= 0 && t < B && label_of(labels, n_labels, t) == bg) ? 1.f : 0.f;
        }
        const float denom = n + 1e-8f;
        const float li_loss = -(A - n * logf(Z)) / denom;
        acc += (double)li_loss;
        row_a[i] = skip ? 0.f : (n / denom) / Z;
        row_w[i] = skip ? 0.f : 1.0f / denom;
    }
    const double tot = block_sum_d(acc, red);
    if (threadIdx.x == 0) {
        hdr->skip = skip ? 1 : 0;
        out_loss[0] = skip ? 0.f : loss_weight * (float)(tot / (double)B);
    }
}

// one wave per row: reduce the Y-split partials, apply d(normalize) and the upstream scalar gradient
__global__ void supcon_bwd_fin_kernel(const float* __restrict__ fhat, const float* __restrict__ invnorm,
                                      const float* __restrict__ dpart, int B, int D, int nsplit,
                                      const WsHeader* __restrict__ hdr, const float* __restrict__ gout,
                                      float scale, float* __restrict__ dfeats) {
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= B) return;
    const float g0 = (gout ? gout[0] : 1.0f) * scale;
    const bool skip = hdr->skip != 0;
    float g[8];  // D <= 512
    float dot = 0.f;
    int k = 0;
    for (int d = lane; d < D; d += 64, ++k) {
        float v = 0.f;
        for (int s = 0; s < nsplit; ++s) v += dpart[((size_t)s * B + row) * D + d];
        g[k] = v;
        dot += v * fhat[(size_t)row * D + d];
    }
    dot = wave_sum(dot);
    const float inv = invnorm[row];
    k = 0;
    for (int d = lane; d < D; d += 64, ++k) {
        const float v = (g[k] - dot * fhat[(size_t)row * D + d]) * inv * g0;
        dfeats[(size_t)row * D + d] = skip ? 0.f : v;
    }
}

struct Plan {
    int nsplit, tiles_per_split, xblocks;
    size_t off_fhat, off_inv, off_a, off_w, off_pz, off_pa, off_dpart, total;
};

Plan make_plan(int B, int D) {
    Plan p;
    const int nyt = oadg_cdiv(B, TILE);
    p.xblocks = oadg_cdiv(B, TILE * WAVES);
    int want = oadg_cdiv(512, p.xblocks);  // ~2 blocks per CU over 256 CUs
    if (want < 1) want = 1;
    if (want > nyt) want = nyt;
    p.tiles_per_split = oadg_cdiv(nyt, want);
    p.nsplit = oadg_cdiv(nyt, p.tiles_per_split);
    size_t o = sizeof(WsHeader);
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) & ~(size_t)255; return r; };
    p.off_fhat = take((size_t)B * D * 4);
    p.off_inv = take((size_t)B * 4);
    p.off_a = take((size_t)B * 4);
    p.off_w = take((size_t)B * 4);
    p.off_pz = take((size_t)p.nsplit * B * 4);
    p.off_pa = take((size_t)p.nsplit * B * 4);
    p.off_dpart = take((size_t)p.nsplit * B * D * 4);
    p.total = o;
    return p;
}

template <int D>
int launch_tiles(bool bwd, const Plan& p, char* ws, const int64_t* labels, int B, int n_labels, int ori,
                 int rp, float inv_t, hipStream_t st) {
    const float* fhat = (const float*)(ws + p.off_fhat);
    WsHeader* hdr = (WsHeader*)ws;
    dim3 grid(p.xblocks, p.nsplit), block(256);
    if (!bwd)
        hipLaunchKernelGGL((supcon_tile_kernel<D, false>), grid, block, 0, st, fhat, labels, B, n_labels,
                           ori, rp, inv_t, hdr, (const float*)nullptr, (const float*)nullptr,
                           (float*)(ws + p.off_pz), (float*)(ws + p.off_pa), (float*)nullptr,
                           p.tiles_per_split);
    else
        hipLaunchKernelGGL((supcon_tile_kernel<D, true>), grid, block, 0, st, fhat, labels, B, n_labels,
                           ori, rp, inv_t, hdr, (const float*)(ws + p.off_a), (const float*)(ws + p.off_w),
                           (float*)nullptr, (float*)nullptr, (float*)(ws + p.off_dpart),
                           p.tiles_per_split);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

}  // namespace

extern "C" {

size_t oadg_supcon_workspace_bytes(int B, int D) {
    if (B <= 0 || D <= 0) return 0;
    return make_plan(B, D).total;
}

int oadg_supcon_fwd(const float* feats, const int64_t* labels, int B, int D, int n_labels, int ori_size,
                    int rp_size, float temper, int min_samples, float loss_weight, void* workspace,
                    size_t workspace_bytes, float* out_loss, void* stream) {
    if (!feats || !labels || !workspace || !out_loss) return OADG_EARG;
    if (B < 2 || n_labels < 1 || n_labels > B || ori_size < 0 || rp_size < 0 || temper <= 0.f)
        return OADG_EARG;
    if (D != 64 && D != 128 && D != 256) return OADG_EARG;
    const Plan p = make_plan(B, D);
    if (workspace_bytes < p.total) return OADG_ESIZE;
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    hipError_t e = hipMemsetAsync(ws, 0, sizeof(WsHeader), st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(supcon_prep_kernel, dim3(oadg_cdiv(B, PREP_ROWS)), dim3(256), 0, st, feats, labels, B, D,
                       n_labels, (float*)(ws + p.off_fhat), (float*)(ws + p.off_inv), (WsHeader*)ws);
    OADG_LAUNCH_CHECK();
    const float inv_t = 1.0f / temper;
    int rc;
    if (D == 256) rc = launch_tiles<256>(false, p, ws, labels, B, n_labels, ori_size, rp_size, inv_t, st);
    else if (D == 128) rc = launch_tiles<128>(false, p, ws, labels, B, n_labels, ori_size, rp_size, inv_t, st);
    else rc = launch_tiles<64>(false, p, ws, labels, B, n_labels, ori_size, rp_size, inv_t, st);
    if (rc) return rc;
    hipLaunchKernelGGL(supcon_fin_kernel, dim3(1), dim3(1024), 0, st, labels, B, n_labels, ori_size,
                       rp_size, p.nsplit, min_samples, (WsHeader*)ws, (const float*)(ws + p.off_pz),
                       (const float*)(ws + p.off_pa), (float*)(ws + p.off_a), (float*)(ws + p.off_w),
                       loss_weight, out_loss);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

int oadg_supcon_bwd(const int64_t* labels, int B, int D, int n_labels, int ori_size, int rp_size,
                    float temper, float loss_weight, const float* grad_out, void* workspace,
                    size_t workspace_bytes, float* dfeats, void* stream) {
    if (!labels || !workspace || !dfeats) return OADG_EARG;
    if (B < 2 || n_labels < 1 || n_labels > B || temper <= 0.f) return OADG_EARG;
    if (D != 64 && D != 128 && D != 256) return OADG_EARG;
    const Plan p = make_plan(B, D);
    if (workspace_bytes < p.total) return OADG_ESIZE;
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const float inv_t = 1.0f / temper;
    int rc;
    if (D == 256) rc = launch_tiles<256>(true, p, ws, labels, B, n_labels, ori_size, rp_size, inv_t, st);
    else if (D == 128) rc = launch_tiles<128>(true, p, ws, labels, B, n_labels, ori_size, rp_size, inv_t, st);
    else rc = launch_tiles<64>(true, p, ws, labels, B, n_labels, ori_size, rp_size, inv_t, st);
    if (rc) return rc;
    // dL/dS = W / B ; dS/dF^ carries 1/tau ; loss_weight and the upstream scalar come last
    const float scale = loss_weight * inv_t / (float)B;
    hipLaunchKernelGGL(supcon_bwd_fin_kernel, dim3(oadg_cdiv(B, 4)), dim3(256), 0, st,
                       (const float*)(ws + p.off_fhat), (const float*)(ws + p.off_inv),
                       (const float*)(ws + p.off_dpart), B, D, p.nsplit, (const WsHeader*)ws, grad_out,
                       scale, dfeats);
    OADG_LAUNCH_CHECK();
    return OADG_OK;
}

// 0 = ok, bit0 = a label was outside [0,1024). Reads the workspace header; caller must have synchronised.
int oadg_supcon_status(const void* workspace_host_copy) {
    if (!workspace_host_copy) return OADG_EARG;
    return ((const WsHeader*)workspace_host_copy)->error;
}

}  // extern "C"
