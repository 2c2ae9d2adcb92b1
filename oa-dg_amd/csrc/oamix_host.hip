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
