"""Generate tests/golden/core_reference.npz from the reference's own core functions (run by path, refload.py):
  * generate_random_bboxes_xy            mmdet/models/detectors/two_stage.py:389-419   (seeds 0-3, H/W as the caller
                                          passes them: img.shape[2:] = (H, W) consumed as (width, height))
  * AnchorGenerator.grid_priors          mmdet/core/anchor/anchor_generator.py
  * bbox_overlaps (torch)                mmdet/core/bbox/iou_calculators/iou2d_calculator.py
  * MaxIoUAssigner + RandomSampler       max_iou_assigner.py, random_sampler.py under torch.manual_seed
  * bbox2delta / delta2bbox              delta_xywh_bbox_coder.py (incl. a zero-size proposal: the fork's guard)
Run here only:  python tests/golden/make_golden_core.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import refload  # noqa: E402
from inputs import synthetic_boxes  # noqa: E402


def proposals_and_gts(seed, n_prop, n_gt, W=320, H=200):
    rs = np.random.RandomState(500 + seed)
    gts = synthetic_boxes(rs, n_gt, H, W, 12, 120)
    props = synthetic_boxes(rs, n_prop, H, W, 6, 150)
    k = min(n_gt, n_prop // 4)
    props[:k] = gts[:k] + rs.uniform(-3, 3, (k, 4)).astype(np.float32)      # some high-IoU proposals
    props[k:2 * k] = gts[:k]                                              # exact duplicates of gts (ties)
    return props.astype(np.float32), gts.astype(np.float32), rs.randint(0, 8, n_gt).astype(np.int64)


def main():
    refload.install()
    out = {}
    gen = refload.ref('mmdet.models.detectors.two_stage', 'generate_random_bboxes_xy')
    for seed in range(4):
        rs = np.random.RandomState(seed)
        gts = synthetic_boxes(rs, 6, 256, 512, 16, 160)
        np.random.seed(seed)
        b = gen((256, 512), num_bboxes=10, bboxes_xy=gts, scales=(0.01, 0.3), ratios=(0.3, 1 / 0.3), iou_max=0.7,
                iou_min=0.0)
        out[f'rand_gts{seed}'], out[f'rand_boxes{seed}'] = gts, np.asarray(b)
        out[f'rand_rng{seed}'] = np.array([np.random.uniform()])
    AG = refload.ref('mmdet.core.anchor.anchor_generator', 'AnchorGenerator')
    ag = AG(strides=[4, 8, 16, 32, 64], ratios=[0.5, 1.0, 2.0], scales=[8])
    for i, a in enumerate(ag.grid_priors([(6, 9), (3, 5), (2, 3), (1, 2), (1, 1)], device='cpu')):
        out[f'anchors{i}'] = a.numpy()
    ov = refload.ref('mmdet.core.bbox.iou_calculators.iou2d_calculator', 'bbox_overlaps')
    MA = refload.ref('mmdet.core.bbox.assigners.max_iou_assigner', 'MaxIoUAssigner')
    RS = refload.ref('mmdet.core.bbox.samplers.random_sampler', 'RandomSampler')
    coder = refload.ref('mmdet.core.bbox.coder.delta_xywh_bbox_coder')
    cfgs = [dict(pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3, match_low_quality=True, ignore_iof_thr=-1),
            dict(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, match_low_quality=False, ignore_iof_thr=-1)]
    samplers = [dict(num=64, pos_fraction=0.5, neg_pos_ub=-1, add_gt_as_proposals=False),
                dict(num=128, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=True)]
    for seed in range(3):
        props, gts, labels = proposals_and_gts(seed, 1000, 5)
        out[f'iou{seed}'] = ov(torch.from_numpy(gts), torch.from_numpy(props)).numpy()
        for c, (acfg, scfg) in enumerate(zip(cfgs, samplers)):
            ar = MA(**acfg).assign(torch.from_numpy(props), torch.from_numpy(gts), None, torch.from_numpy(labels))
            out[f'assign{seed}_{c}_gt_inds'] = ar.gt_inds.numpy().copy()
            out[f'assign{seed}_{c}_max_overlaps'] = ar.max_overlaps.numpy().copy()
            out[f'assign{seed}_{c}_labels'] = ar.labels.numpy().copy()
            torch.manual_seed(seed)
            sr = RS(**scfg).sample(ar, torch.from_numpy(props), torch.from_numpy(gts), torch.from_numpy(labels))
            out[f'sample{seed}_{c}_pos'] = sr.pos_inds.numpy()
            out[f'sample{seed}_{c}_neg'] = sr.neg_inds.numpy()
            out[f'sample{seed}_{c}_rng'] = torch.rand(1).numpy()
        p = torch.from_numpy(props[:64].copy())
        p[3, 2] = p[3, 0]                 # zero-width proposal
        p[5, 3] = p[5, 1]                 # zero-height proposal
        p[7, 2:] = p[7, :2]               # both
        g = torch.from_numpy(np.resize(gts, (64, 4)).copy())
        d = coder.bbox2delta(p, g, (0., 0., 0., 0.), (0.1, 0.1, 0.2, 0.2))
        out[f'delta{seed}_p'], out[f'delta{seed}_g'], out[f'delta{seed}_d'] = p.numpy(), g.numpy(), d.numpy()
        dd = torch.from_numpy(np.random.RandomState(seed).normal(0, 0.5, (64, 4)).astype(np.float32))
        out[f'decode{seed}_d'] = dd.numpy()
        out[f'decode{seed}_b'] = coder.delta2bbox(p, dd, (0., 0., 0., 0.), (1., 1., 1., 1.), max_shape=(200, 320),
                                                 wh_ratio_clip=16 / 1000).numpy()
    path = os.path.join(HERE, 'core_reference.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
