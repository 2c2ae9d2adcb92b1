"""Generate tests/golden/oamix_reference.npz: the GENUINE reference ``OAMix`` pipeline step
(mmdet/datasets/pipelines/oa_mix.py + augmix.py + bbox_augmentation.py, every line reference code) run on seeded
inputs.  OpenCV is not installed in this container, so the reference's ``cv2`` calls resolve to oracle/cv2_stub.py
(the restatement of warpAffine / GaussianBlur / resize / spectral-residual saliency): the fixture pins everything
the reference itself owns - RNG consumption order, region sampling, op dispatch, Pillow colour ops, box-wise and
background-wise blending, object-aware mixing, numpy dtype promotion - and the oracle's OA-Mix must reproduce it
byte for byte.  The cv2 leaves stay "parity unpinned" (DESIGN.md section 5).
Run here only:  python tests/golden/make_golden_oamix.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import refload  # noqa: E402
from inputs import lowpass_image, synthetic_boxes  # noqa: E402
from oracle import cv2_stub  # noqa: E402

CASES = [  # (seed, version, H, W, n_gt)
    (0, 'augmix', 96, 160, 3), (1, 'augmix', 96, 160, 4), (2, 'augmix', 120, 136, 2), (3, 'augmix', 64, 200, 5),
    (4, 'augmix.all', 96, 160, 3), (5, 'augmix.all', 96, 160, 4), (6, 'augmix.all', 120, 136, 1),
    (7, 'augmix.all', 64, 200, 6),
]


def case_inputs(seed, H, W, n_gt):
    rs = np.random.RandomState(1000 + seed)
    img = lowpass_image(rs, H, W)
    gts = synthetic_boxes(rs, n_gt, H, W, 10, min(H, W) // 2)
    return img, gts


def main():
    refload.install(cv2_module=cv2_stub.make_cv2())
    OAMix = refload.ref('mmdet.datasets.pipelines.oa_mix', 'OAMix')
    out = {}
    for seed, version, H, W, n_gt in CASES:
        img, gts = case_inputs(seed, H, W, n_gt)
        t = OAMix(version=version, num_views=2, keep_orig=True, severity=10, random_box_ratio=(3, 1 / 3),
                  random_box_scale=(0.01, 0.1), oa_random_box_scale=(0.005, 0.1), oa_random_box_ratio=(3, 1 / 3),
                  spatial_ratio=4, sigma_ratio=0.3)
        np.random.seed(seed)
        r = t(dict(img=img.copy(), gt_bboxes=gts.copy(), img_fields=['img']))
        tag = f's{seed}'
        out[tag + '_img2'] = np.asarray(r['img2'])
        out[tag + '_img'] = np.asarray(r['img'])
        out[tag + '_gt_bboxes2'] = np.asarray(r['gt_bboxes2'], dtype=np.float32)
        out[tag + '_multilevel_boxes'] = np.asarray(r['multilevel_boxes'], dtype=np.int64)
        out[tag + '_oamix_boxes'] = np.asarray(r['oamix_boxes'], dtype=np.int64)
        out[tag + '_rng_after'] = np.array([np.random.uniform()])      # the global stream was consumed identically
        assert out[tag + '_img2'].dtype == np.uint8 and out[tag + '_img2'].shape == (H, W, 3)
    out['cases'] = np.array([(s, 0 if v == 'augmix' else 1, H, W, n) for s, v, H, W, n in CASES], dtype=np.int64)
    path = os.path.join(HERE, 'oamix_reference.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
