"""RoIAlign backward in isolation on the RoIs of a real bench step (tools/probe, not part of the product):
python tools/probe/roi_bwd_probe.py  -> ms per launch of the tile path and the atomic path (+ OADG_TILE_ABL ablations)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oadg_amd  # noqa: E402,F401
from oadg_amd import Config, build_detector, hip_conv, hip_ops  # noqa: E402
from oadg_amd.apis import TrainEngine, build_optimizer, set_random_seed  # noqa: E402
from oadg_amd.pipelines import DevicePipeline, SyntheticCityscapes  # noqa: E402


def real_rois(dev):
    cache = '/tmp/roi_probe_rois.pt'
    if os.path.exists(cache):
        return torch.load(cache).to(dev)
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/oadg/faster_rcnn_r50_fpn_1x_cityscapes_oadg.py'))
    set_random_seed(0)
    det = build_detector(cfg.model)
    det.init_weights(allow_missing_pretrained=True)
    det = det.to(dev).to(memory_format=torch.channels_last).train()
    det.log_vars_on_host = False
    eng = TrainEngine(det, build_optimizer(det, cfg.optimizer), amp_dtype=torch.bfloat16)
    ds = SyntheticCityscapes(device=dev)
    pipe = DevicePipeline(cfg.data.train.pipeline, dtype=torch.bfloat16)
    for i in range(3):
        eng.step(pipe(*ds.batch(range(i * 4, i * 4 + 4))))
    torch.cuda.synchronize()
    rois = torch.cat([r.float() for r in det.roi_head._last_rois])
    hip_conv.enable(False)
    torch.save(rois.cpu(), cache)
    return rois


def pair_histogram(rois, lvl, strides, tpw=8):
    """(tile, RoI) pairs per 8 x 8 tile and per workgroup of `tpw` consecutive tiles: how uneven is the kernel's work?"""
    import math
    r = rois.cpu()
    lv = lvl.cpu().long()
    for li, s in enumerate(strides):
        H, W = 1024 // s, 2048 // s
        ty, tx = (H + 7) // 8, (W + 7) // 8
        cnt = torch.zeros(8, ty, tx, dtype=torch.int32)
        for row in r[lv == li]:
            n = int(row[0])
            x0, y0, x1, y1 = [float(v) / s - 0.5 for v in row[1:]]
            xl, xh = max(math.floor(x0) - 1, 0), min(math.ceil(x1) + 1, W - 1)
            yl, yh = max(math.floor(y0) - 1, 0), min(math.ceil(y1) + 1, H - 1)
            if xh < 0 or yh < 0 or xl >= W or yl >= H:
                continue
            cnt[n, yl // 8:yh // 8 + 1, xl // 8:xh // 8 + 1] += 1
        flat = cnt.reshape(8, -1)
        pad = (-flat.shape[1]) % tpw
        wg = torch.nn.functional.pad(flat, (0, pad)).reshape(8, -1, tpw).sum(-1).flatten()
        f = flat.flatten().float()
        print(f'level {li}: tiles {f.numel()}, pairs {int(f.sum())}, empty {float((f == 0).float().mean()):.2f}, per tile mean {float(f.mean()):.2f} '
              f'p99 {float(f.quantile(0.99)):.0f} max {int(f.max())}; per workgroup of {tpw}: mean {float(wg.float().mean()):.1f} '
              f'p99 {float(wg.float().quantile(0.99)):.0f} max {int(wg.max())}')


def main():
    dev = torch.device('cuda:0')
    rois = real_rois(dev)
    K = rois.shape[0]
    w, h = rois[:, 3] - rois[:, 1], rois[:, 4] - rois[:, 2]
    lvl = torch.floor(torch.log2(torch.sqrt(w.clamp(min=0) * h.clamp(min=0)) / 56 + 1e-6)).clamp(0, 3)
    print('K', K, 'per level', [int((lvl == i).sum()) for i in range(4)], 'mean side px', float(torch.sqrt(w * h).mean()))
    strides = [4, 8, 16, 32]
    pair_histogram(rois, lvl, strides)
    feats = [torch.randn(8, 256, 1024 // s, 2048 // s, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
             for s in strides]
    gout = torch.randn(K, 256, 7, 7, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    for tiles in (True, False):
        hip_ops.BWD_TILES = tiles
        fg = [f.clone().requires_grad_(True) for f in feats]
        out = hip_ops.roi_align_fpn(fg, rois, 7, [1.0 / s for s in strides])
        hip_ops.TIMERS = {'roi_align_bwd': []}
        for rep in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out.backward(gout, retain_graph=True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
        ev = [a.elapsed_time(b) for a, b, _ in hip_ops.TIMERS['roi_align_bwd']]
        hip_ops.TIMERS = None
        print('tiles' if tiles else 'atomic (+fill+cast)', f'wall {(t1 - t0) * 1e3:.3f} ms, kernel events (ms):', ' '.join(f'{e:.3f}' for e in ev),
              f'(abl {os.environ.get("OADG_TILE_ABL", "0")})')


if __name__ == '__main__':
    main()
