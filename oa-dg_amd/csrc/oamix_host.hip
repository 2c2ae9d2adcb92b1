// Host-side helper of the batched bbox-only chain (no device code): dependency levels of the per-box steps of
// bbox_augmentation.py:74-88 so that independent boxes share a launch (oadg_oamix_bbox_chain).
//
// Step j must run after every earlier step i whose WRITTEN rect meets j's READ footprint (read after write) or
// whose read footprint meets j's written rect (write after read; write after write is implied, W is inside F).
// level[j] = 1 + max level of those, else 0.  Up to 256 steps: exact pair tests (candidates through a 64-px cell
// grid).  More steps (BASELINE configs[4]: 4096 boxes per image): a conservative bound with no pair tests at all - two
// 16-px cell maps hold the highest level that wrote / read each cell; a step goes one level above everything recorded
// in the cells it touches (shared cell is necessary for overlap, so every true dependency is still ordered; a few
// steps land a level later than they had to).  ~0.5 ms for 4096 boxes.
#include <math.h>
#include <stdint.h>
#include <algorithm>
#include <vector>
#include "common.h"
#include "oadg_hip.h"

extern "C" int oadg_oamix_bbox_levels(const int* rects, const double* minvs, int n, int H, int W, int* level) {
    if (n < 0 || H < 1 || W < 1 || (n > 0 && (!rects || !minvs || !level))) return OADG_EARG;
    constexpr int CELL = 64, FINE = 16;
    const bool exact = n <= 256;
    const int gx = (W + CELL - 1) / CELL, gy = (H + CELL - 1) / CELL;
    const int hx = (W + FINE - 1) / FINE, hy = (H + FINE - 1) / FINE;
    std::vector<std::vector<int>> cells(exact ? (size_t)gx * gy : 0);
    std::vector<int> wrote(exact ? 0 : (size_t)hx * hy, -1), readl(exact ? 0 : (size_t)hx * hy, -1);
    struct Box { int wx0, wy0, wx1, wy1, fx0, fy0, fx1, fy1; };      // inclusive pixel bounds
    std::vector<Box> b((size_t)n);
    for (int j = 0; j < n; ++j) {
        const int x0 = rects[4 * j], y0 = rects[4 * j + 1], x1 = x0 + rects[4 * j + 2] - 1, y1 = y0 + rects[4 * j + 3] - 1;
        if (x0 < 0 || y0 < 0 || x1 >= W || y1 >= H || x1 < x0 || y1 < y0) return OADG_EARG;
        const double* m = minvs + 6 * j;
        // the warp is affine: the warped corners bound every source coordinate; +-2 covers the second bilinear tap and
        // the 1/32-px fixed-point rounding of the kernel's coordinates
        double sxmin = 1e300, sxmax = -1e300, symin = 1e300, symax = -1e300;
        const int cx[4] = {x0, x1, x0, x1}, cy[4] = {y0, y0, y1, y1};
        for (int k = 0; k < 4; ++k) {
            const double sx = m[0] * cx[k] + m[1] * cy[k] + m[2], sy = m[3] * cx[k] + m[4] * cy[k] + m[5];
            sxmin = std::min(sxmin, sx); sxmax = std::max(sxmax, sx);
            symin = std::min(symin, sy); symax = std::max(symax, sy);
        }
        auto clampi = [](double v, int lo, int hi) { return (int)std::min((double)hi, std::max((double)lo, v)); };
        Box& q = b[j];
        q.wx0 = x0; q.wy0 = y0; q.wx1 = x1; q.wy1 = y1;
        q.fx0 = clampi(std::min(floor(sxmin) - 2.0, (double)x0), 0, W - 1);
        q.fy0 = clampi(std::min(floor(symin) - 2.0, (double)y0), 0, H - 1);
        q.fx1 = clampi(std::max(ceil(sxmax) + 2.0, (double)x1), 0, W - 1);
        q.fy1 = clampi(std::max(ceil(symax) + 2.0, (double)y1), 0, H - 1);
        if (!exact) {
            int lv = 0;
            for (int cyi = q.fy0 / FINE; cyi <= q.fy1 / FINE; ++cyi)          // read after write
                for (int cxi = q.fx0 / FINE; cxi <= q.fx1 / FINE; ++cxi)
                    lv = std::max(lv, wrote[(size_t)cyi * hx + cxi] + 1);
            for (int cyi = q.wy0 / FINE; cyi <= q.wy1 / FINE; ++cyi)          // write after read
                for (int cxi = q.wx0 / FINE; cxi <= q.wx1 / FINE; ++cxi)
                    lv = std::max(lv, readl[(size_t)cyi * hx + cxi] + 1);
            level[j] = lv;
            for (int cyi = q.fy0 / FINE; cyi <= q.fy1 / FINE; ++cyi)
                for (int cxi = q.fx0 / FINE; cxi <= q.fx1 / FINE; ++cxi) {
                    int& r = readl[(size_t)cyi * hx + cxi];
                    r = std::max(r, lv);
                }
            for (int cyi = q.wy0 / FINE; cyi <= q.wy1 / FINE; ++cyi)
                for (int cxi = q.wx0 / FINE; cxi <= q.wx1 / FINE; ++cxi) {
                    int& wv = wrote[(size_t)cyi * hx + cxi];
                    wv = std::max(wv, lv);
                }
            continue;
        }
        int lv = 0;
        for (int cyi = q.fy0 / CELL; cyi <= q.fy1 / CELL; ++cyi)
            for (int cxi = q.fx0 / CELL; cxi <= q.fx1 / CELL; ++cxi)
                for (int i : cells[(size_t)cyi * gx + cxi]) {
                    const Box& p = b[i];
                    const bool raw = p.wx0 <= q.fx1 && p.wx1 >= q.fx0 && p.wy0 <= q.fy1 && p.wy1 >= q.fy0;
                    const bool war = p.fx0 <= q.wx1 && p.fx1 >= q.wx0 && p.fy0 <= q.wy1 && p.fy1 >= q.wy0;
                    if ((raw || war) && level[i] + 1 > lv) lv = level[i] + 1;
                }
        level[j] = lv;
        for (int cyi = q.fy0 / CELL; cyi <= q.fy1 / CELL; ++cyi)        // F contains W: every later conflict with j
            for (int cxi = q.fx0 / CELL; cxi <= q.fx1 / CELL; ++cxi)    // is found through one of F's cells
                cells[(size_t)cyi * gx + cxi].push_back(j);
    }
    return OADG_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The host side of one bboxes_only_* op (bbox_augmentation.py:31-88 with the augmix.py:83-188 leaf matrices) for ALL gt
// boxes in one call: what oadg_amd/pipelines/oa_mix.py did with ~25 numpy expressions per op (2 ms of Python per view,
// under the interpreter lock that the training thread needs) - draws -> level / sign -> affine matrix in the dtype cv2
// receives it -> inverse -> live steps -> dependency levels -> level-major step table + tile prefix, written straight
// into the caller's (pinned) staging buffer.  Every expression keeps numpy's operation order (IEEE doubles, float32
// round trips where the reference builds np.float32 matrices, libm cos / sin of the integer angle like math.cos):
// bit-identical matrices (tests/test_hip_oamix.py compares the images with the oracle byte for byte).
//   kind: 0 rotate, 1 shear_x, 2 shear_y, 3 translate_x, 4 translate_y
//   ib [n][4]: int64 box corners (truncated gt boxes); support [n][4]: x0, y0, w, h of the box's mask support (w <= 0 or
//   h <= 0: the mask is empty - the step draws but changes nothing); draws [2 m]: the m drawing boxes' (level, sign)
//   uniforms in box order (boxes with integer width or height < 1 return before drawing, :45-47).
// staging: [n_live] oadg_bbox_step, [n_live + 1] int32 tile prefix, [n_levels + 1] int32 level table (a copy of the host
// array level_first).
// out[0] = n_live, out[1] = n_levels, out[2] = total tiles; *area_sum = sum of the rect areas.
extern "C" size_t oadg_oamix_bbox_plan_bytes(int n) {
    return (size_t)(n > 0 ? n : 0) * sizeof(oadg_bbox_step) + (2 * (size_t)(n > 0 ? n : 0) + 3) * sizeof(int) + 8;
}

extern "C" int oadg_oamix_bbox_plan(int kind, double severity, const long long* ib, const int* support, int n,
                                    const double* draws, int n_draws, int H, int W, void* staging, size_t staging_bytes,
                                    int* level_first, int* out, long long* area_sum) {
    if (kind < 0 || kind > 4 || n < 0 || H < 1 || W < 1 || !out || !area_sum || (n > 0 && (!ib || !support || !staging || !level_first)))
        return OADG_EARG;
    if (staging_bytes < oadg_oamix_bbox_plan_bytes(n)) return OADG_ESIZE;
    out[0] = out[1] = out[2] = 0;
    *area_sum = 0;
    auto f32 = [](double v) { return (double)(float)v; };
    std::vector<int> rects, rows;
    std::vector<double> minvs;
    rects.reserve((size_t)n * 4); rows.reserve(n); minvs.reserve((size_t)n * 6);
    int d = 0;
    for (int i = 0; i < n; ++i) {
        const long long x1i = ib[4 * i], y1i = ib[4 * i + 1], x2i = ib[4 * i + 2], y2i = ib[4 * i + 3];
        if ((x2i - x1i) < 1 || (y2i - y1i) < 1) continue;                 // returns before any draw
        if (2 * d + 1 >= n_draws) return OADG_EARG;
        const double level = 0.1 + (severity - 0.1) * draws[2 * d];
        const bool flip = draws[2 * d + 1] > 0.5;
        ++d;
        const double x1 = (double)x1i, y1 = (double)y1i, x2 = (double)x2i, y2 = (double)y2i;
        const double cx = (x1 + x2) / 2., cy = (y1 + y2) / 2.;
        double M[6] = {0, 0, 0, 0, 0, 0};
        if (kind == 0) {
            long long deg = (long long)trunc(level * 30 / 10);
            if (flip) deg = -deg;
            const double cxf = f32(cx), cyf = f32(cy);                     // cv2.getRotationMatrix2D takes a Point2f
            const double ang = (double)deg * M_PI / 180.0;
            const double alpha = cos(ang) * 1.0, beta = sin(ang) * 1.0;
            M[0] = alpha; M[1] = beta; M[2] = (1 - alpha) * cxf - beta * cyf;
            M[3] = -beta; M[4] = alpha; M[5] = beta * cxf + (1 - alpha) * cyf;
        } else if (kind == 1 || kind == 2) {
            double lvl = level * 0.3 / 10.;
            if (flip) lvl = -lvl;
            if (kind == 1) { M[0] = 1.0; M[1] = f32(-lvl); M[2] = f32(-(-lvl * cy)); M[4] = 1.0; }
            else { M[0] = 1.0; M[3] = f32(-lvl); M[4] = 1.0; M[5] = f32(-(-lvl * cx)); }
        } else {
            const double size = kind == 3 ? (x2 - x1 + 1) : (y2 - y1 + 1);
            double lvl = trunc(level * (size / 3) / 10);
            if (flip) lvl = -lvl;
            M[0] = 1.0; M[4] = 1.0;
            M[kind == 3 ? 2 : 5] = f32(-lvl);
        }
        const int* sp = support + 4 * i;
        if (sp[2] <= 0 || sp[3] <= 0) continue;                            // mask identically zero: image unchanged
        double D = M[0] * M[4] - M[1] * M[3];
        D = D != 0 ? 1.0 / D : 0.0;
        const double A11 = M[4] * D, A22 = M[0] * D, m1 = M[1] * -D, m3 = M[3] * -D;
        const double b1 = -A11 * M[2] - m1 * M[5], b2 = -m3 * M[2] - A22 * M[5];
        const double mi[6] = {A11, m1, b1, m3, A22, b2};
        minvs.insert(minvs.end(), mi, mi + 6);
        rects.insert(rects.end(), sp, sp + 4);
        rows.push_back(i);
    }
    if (2 * d != n_draws) return OADG_EARG;
    const int nl = (int)rows.size();
    out[0] = nl;
    if (nl == 0) return OADG_OK;
    std::vector<int> level(nl);
    const int rc = oadg_oamix_bbox_levels(rects.data(), minvs.data(), nl, H, W, level.data());
    if (rc) return rc;
    std::vector<int> order(nl);
    for (int j = 0; j < nl; ++j) order[j] = j;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return level[a] < level[b]; });
    oadg_bbox_step* steps = (oadg_bbox_step*)staging;
    int* tiles = (int*)((unsigned char*)staging + (size_t)nl * sizeof(oadg_bbox_step));
    const int n_levels = level[order[nl - 1]] + 1;
    long long area_tot = 0, cum = 0, level_base = 0;
    int cur = -1, t = 0;
    tiles[0] = 0;
    for (int p = 0; p < nl; ++p) {
        const int j = order[p];
        while (cur < level[j]) { level_first[++cur] = p; level_base = cum; }
        oadg_bbox_step& s = steps[p];
        for (int q = 0; q < 6; ++q) s.minv[q] = minvs[6 * j + q];
        for (int q = 0; q < 4; ++q) s.rect[q] = rects[4 * j + q];
        s.row = rows[j];
        const long long area = (long long)rects[4 * j + 2] * rects[4 * j + 3];
        s.scratch_off = cum - level_base;      // the rects of one level are disjoint: their packed images fit the H*W*3 scratch
        cum += (3 * area + 3) / 4 * 4;         // every rect starts on a 4-byte boundary of the scratch image
        area_tot += area;
        t += (int)((area + 1023) / 1024);      // one workgroup = 256 threads x 4 pixels
        tiles[p + 1] = t;
    }
    level_first[n_levels] = nl;
    for (int l = 0; l <= n_levels; ++l) tiles[nl + 1 + l] = level_first[l];      // device copy of the level table
    out[1] = n_levels; out[2] = t;
    *area_sum = area_tot;
    return OADG_OK;
}
