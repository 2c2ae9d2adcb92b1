"""Time (and optionally torch-profile) the detector train step on synthetic collated batches.
usage: python tools/profile_step.py [--n 4] [--h 1024 --w 2048] [--amp bf16|none] [--steps 5] [--prof]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oadg_amd  # noqa: E402
from oadg_amd import Config, build_detector  # noqa: E402
from oadg_amd.apis import TrainEngine, build_optimizer, set_random_seed  # noqa: E402


def synth(n, h, w, dev, n_gt=20):
    g = torch.Generator(device=dev).manual_seed(0)
    shape = (h, w, 3)
    def boxes(k, lo, hi):
        bw = torch.rand(k, generator=g, device=dev) * (hi - lo) + lo
        bh = torch.rand(k, generator=g, device=dev) * (hi - lo) + lo
        x1 = torch.rand(k, generator=g, device=dev) * (w - bw)
        y1 = torch.rand(k, generator=g, device=dev) * (h - bh)
        return torch.stack([x1, y1, x1 + bw, y1 + bh], 1)
    d = dict(img=torch.randn(n, 3, h, w, device=dev, generator=g).contiguous(memory_format=torch.channels_last),
             img2=torch.randn(n, 3, h, w, device=dev, generator=g).contiguous(memory_format=torch.channels_last),
             gt_bboxes=[boxes(n_gt, 24, 400) for _ in range(n)],
             gt_labels=[torch.randint(0, 8, (n_gt,), device=dev, generator=g) for _ in range(n)],
             multilevel_boxes=[boxes(2, 50, 300).round().long().cpu() for _ in range(n)],
             oamix_boxes=[boxes(3, 50, 300).round().long().cpu() for _ in range(n)],
             img_metas=[dict(img_shape=shape, pad_shape=shape, ori_shape=shape, scale_factor=1.0, flip=False)
                        for _ in range(n)])
    d['gt_bboxes2'] = [b.clone() for b in d['gt_bboxes']]
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=4)
    ap.add_argument('--h', type=int, default=1024)
    ap.add_argument('--w', type=int, default=2048)
    ap.add_argument('--amp', default='bf16')
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--prof', action='store_true')
    ap.add_argument('--shapes', action='store_true')
    ap.add_argument('--conv', default='mfma')
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    if a.conv == 'mfma' and a.amp == 'bf16':
        from oadg_amd import hip_conv
        hip_conv.enable()
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/oadg/faster_rcnn_r50_fpn_1x_cityscapes_oadg.py'))
    det = build_detector(cfg.model)
    det.init_weights(allow_missing_pretrained=True)
    det = det.to(dev).to(memory_format=torch.channels_last).train()
    set_random_seed(0)
    eng = TrainEngine(det, build_optimizer(det, cfg.optimizer),
                      amp_dtype=torch.bfloat16 if a.amp == 'bf16' else None)
    def batch():
        d = synth(a.n, a.h, a.w, dev)
        return d
    for _ in range(2):
        out = eng.step(batch())
    torch.cuda.synchronize()
    print('warm log_vars', out['log_vars'])
    batches = [batch() for _ in range(a.steps)]
    torch.cuda.synchronize()
    t0 = time.time()
    for b_ in batches:
        out = eng.step(b_)
    t_host = (time.time() - t0) / a.steps
    torch.cuda.synchronize()
    dt = (time.time() - t0) / a.steps
    print(f'host enqueue {t_host * 1e3:.1f} ms/step')
    print(f'amp={a.amp} n={a.n} {a.h}x{a.w}: {dt * 1e3:.1f} ms/step  {a.n / dt:.2f} img/s  '
          f'mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB')
    if a.prof:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as p:
            eng.step(batch())
            torch.cuda.synchronize()
        import collections
        from torch.autograd import DeviceType
        agg = collections.defaultdict(lambda: [0.0, 0])
        for e in p.events():
            if e.device_type == DeviceType.CUDA:
                agg[e.name][0] += e.device_time
                agg[e.name][1] += 1
        tot = sum(v[0] for v in agg.values())
        print(f'device kernels: {tot / 1e3:.2f} ms, {sum(v[1] for v in agg.values())} launches')
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:70]:
            print(f'{v[0] / 1e3:8.3f} ms {v[1]:5d}  {k[:230]}')
        if a.shapes:
            print(p.key_averages(group_by_input_shape=True).table(sort_by='self_cuda_time_total', row_limit=90,
                                                                  max_name_column_width=50,
                                                                  max_shapes_column_width=90))


if __name__ == '__main__':
    main()
