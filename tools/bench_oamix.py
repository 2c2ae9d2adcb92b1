"""OA-Mix and OA-Loss kernel throughput in isolation (BASELINE configs[1] and configs[4]).

  python tools/bench_oamix.py [--config 2|5|both] [--iters K]

Per configuration, one JSON line: ms per view of the whole device pipeline pass (OA-Mix of every image + Normalize/Pad,
HIP events on the launch stream, host wall beside it), the algorithmic bytes of SURVEY 8d's "materialise-every-step"
model for the S actually drawn  -  3*H*W*(2*S + C + 2) + sum over bbox steps of 3*w*h  -  and the fraction of the 8 TB/s
HBM roofline that is; with the per-box launch path (``per_box``) beside the batched one.  Config 5 adds the OA-Loss
isolate: supcon forward + backward at the contrastive batch of 8 images x 512 RoIs x 2 views + random RoIs.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HBM_PEAK = 8.0e12
F32_MFMA_PEAK = 157.3e12


def bench_pipeline(name, H, W, n_boxes, box_size, batch, iters, version='augmix'):
    import oadg_amd  # noqa: F401
    from oadg_amd import Config
    from oadg_amd.pipelines import DevicePipeline, SyntheticCityscapes
    from oadg_amd.pipelines import oa_mix
    dev = torch.device('cuda:0')
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'oadg', 'faster_rcnn_r50_fpn_1x_cityscapes_oadg.py'))
    for t in cfg.data.train.pipeline:
        if t['type'] == 'OAMix':
            t['version'] = version
    ds = SyntheticCityscapes(img_shape=(H, W), num_boxes=n_boxes, num_classes=8, box_size=box_size, device=dev)
    pipe = DevicePipeline(cfg.data.train.pipeline, dtype=torch.bfloat16)
    imgs, boxes, labels = ds.batch(range(batch))
    res = {}
    modes = ('batched', 'per_box', 'workers4') if batch >= 4 else ('batched', 'per_box')
    for mode in modes:  # two launches per dependency level (host side in one C call) / per box / 4 helper threads
        oa_mix.BATCH_BOXES = mode != 'per_box'
        if mode == 'workers4':      # the images of a batch on four helper threads + streams (DevicePipeline._oamix_parallel)
            pipe = DevicePipeline(cfg.data.train.pipeline, dtype=torch.bfloat16, oamix_workers=4)
        np.random.seed(0)
        pipe(imgs, boxes, labels)                       # warm-up (buffers, first-touch)
        torch.cuda.synchronize()
        pipe.oamix.stats = {}
        np.random.seed(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(iters):
            pipe(imgs, boxes, labels)
        e1.record()
        t_host = time.perf_counter() - t0
        torch.cuda.synchronize()
        views = iters * batch
        st = pipe.oamix.stats
        pipe.oamix.stats = None
        S = st.get('compose_steps', 0)
        model = 3.0 * H * W * (2 * S + (3 + 2) * views) + 3.0 * st.get('bbox_px', 0)
        ms = e0.elapsed_time(e1)
        res[mode] = dict(ms_per_view=round(ms / views, 3), host_ms_per_view=round(t_host * 1e3 / views, 3),
                         model_MB_per_view=round(model / views / 1e6, 1),
                         achieved_GBs=round(model / (ms * 1e-3) / 1e9, 1),
                         frac_of_hbm=round(model / (ms * 1e-3) / HBM_PEAK, 4),
                         compose_steps_per_view=round(S / views, 2),
                         bbox_ops_per_view=round(st.get('bbox_ops', 0) / views, 2),
                         bbox_steps_per_op=round(st.get('bbox_steps', 0) / max(st.get('bbox_ops', 0), 1), 1),
                         levels_per_op=round(st.get('bbox_levels', 0) / max(st.get('bbox_ops', 0), 1), 2))
    oa_mix.BATCH_BOXES = True
    out = dict(bench='oamix', config=name, H=H, W=W, boxes_per_image=n_boxes, batch=batch, iters=iters,
               version=version, **res)
    print(json.dumps(out), flush=True)
    return out


def bench_supcon(n_img=8, per_img=512, n_rand=17, dim=256, iters=20):
    from oadg_amd import hip_ops
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    from inputs import supcon_inputs
    dev = torch.device('cuda:0')
    feats, labels = supcon_inputs(0, n_fg_per_img=100, n_rand=n_rand, n_img=n_img, per_img=per_img, dim=dim)
    K, B = labels.shape[0], feats.shape[0]
    f = torch.tensor(feats, device=dev, requires_grad=True)
    lab = torch.tensor(labels, device=dev)

    def once():
        f.grad = None
        loss = hip_ops.supcon_loss(f, lab, K // 2, (B - K) // 2, 0.06, 10, 0.01)
        loss.backward()
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        once()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 6.0 * B * B * dim          # forward 2 B^2 D + backward 4 B^2 D (SURVEY 8d OA-Loss row)
    out = dict(bench='supcon', B=B, D=dim, contrastive_rows=f'{n_img} img x {per_img} RoIs x 2 views + {B - K} random',
               ms_fwd_bwd=round(ms, 3), achieved_TFLOPs=round(flops / (ms * 1e-3) / 1e12, 2),
               frac_of_f32_mfma_peak=round(flops / (ms * 1e-3) / F32_MFMA_PEAK, 4),
               bytes_if_BxB_materialised_MB=round(4.0 * B * B / 1e6, 1))
    print(json.dumps(out), flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='both', choices=['2', '5', 'both'])
    ap.add_argument('--iters', type=int, default=5)
    a = ap.parse_args()
    if a.config in ('2', 'both'):       # BASELINE configs[1]: bs 4, 1024x2048, 20 boxes of 24..400 px
        bench_pipeline('config2: bs4 1024x2048 20 boxes', 1024, 2048, 20, (24, 400), 4, a.iters)
    if a.config in ('5', 'both'):       # BASELINE configs[4]: 4096 boxes of 8..48 px per image, bs 8
        bench_pipeline('config5: bs8 1024x2048 4096 boxes', 1024, 2048, 4096, (8, 48), 8, max(1, a.iters // 5))
        bench_supcon()


if __name__ == '__main__':
    main()
