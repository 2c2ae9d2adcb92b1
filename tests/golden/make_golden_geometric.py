"""Generate tests/golden/geometric_reference.npz: the GENUINE reference ``Resize`` and ``RandomFlip``
(mmdet/datasets/pipelines/transforms.py:45-470) on seeded inputs.  mmcv is absent, so ``mmcv.imrescale`` /
``imresize`` / ``imflip`` resolve to stand-ins built from the oracle's cv2.resize restatement (oracle/cvleaves.py) -
the fixture pins what the reference owns: the random draws and their order, ``rescale_size`` rounding, the float32
scale factors, box scaling / clipping / flipping and the meta keys; the resize pixels stay "parity unpinned".
Run here only:  python tests/golden/make_golden_geometric.py
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
from oracle import cvleaves as cv  # noqa: E402

CASES = [   # (seed, H, W, Resize kwargs, flip kwargs)
    (0, 64, 128, dict(img_scale=[(128, 50), (128, 64)], keep_ratio=True), dict(flip_ratio=0.5)),
    (1, 64, 128, dict(img_scale=[(128, 50), (128, 64)], keep_ratio=True), dict(flip_ratio=0.5)),
    (2, 60, 100, dict(img_scale=(160, 90), keep_ratio=True), dict(flip_ratio=0.5)),
    (3, 60, 100, dict(img_scale=(96, 48), keep_ratio=False), dict(flip_ratio=[0.3, 0.3], direction=['horizontal', 'vertical'])),
    (4, 48, 80, dict(img_scale=[(100, 60), (80, 48), (120, 70)], multiscale_mode='value', keep_ratio=True),
     dict(flip_ratio=0.9, direction=['horizontal', 'vertical', 'diagonal'])),
    (5, 48, 80, dict(img_scale=(80, 48), ratio_range=(0.8, 1.4), keep_ratio=True), dict(flip_ratio=0.5)),
]


def _imresize(img, size, return_scale=False, interpolation='bilinear', backend=None, out=None):
    h, w = img.shape[:2]
    r = cv.resize_u8_cv2(img, size)
    return (r, size[0] / w, size[1] / h) if return_scale else r


def _imrescale(img, scale, return_scale=False, interpolation='bilinear', backend=None):
    h, w = img.shape[:2]
    nw, nh, f = cv.imrescale_size(w, h, scale) if isinstance(scale, tuple) else (int(w * float(scale) + 0.5), int(h * float(scale) + 0.5), scale)
    r = cv.resize_u8_cv2(img, (nw, nh))
    return (r, f) if return_scale else r


def _imflip(img, direction='horizontal'):
    return {'horizontal': np.flip(img, 1), 'vertical': np.flip(img, 0), 'diagonal': np.flip(img, (0, 1))}[direction]


def main():
    refload.install()
    mm = sys.modules['mmcv']
    mm.imresize, mm.imrescale, mm.imflip = _imresize, _imrescale, _imflip
    T = refload.ref('mmdet.datasets.pipelines.transforms')
    out = {}
    for seed, H, W, rk, fk in CASES:
        rs = np.random.RandomState(200 + seed)
        img = lowpass_image(rs, H, W)
        gts = synthetic_boxes(rs, 5, H, W, 6, min(H, W) // 2)
        res = dict(img=img.copy(), gt_bboxes=gts.copy(), img_fields=['img'], bbox_fields=['gt_bboxes'],
                   img_shape=img.shape, ori_shape=img.shape)
        np.random.seed(seed)
        res = T.Resize(**rk)(res)
        res = T.RandomFlip(**fk)(res)
        tag = f's{seed}'
        out[tag + '_img'] = np.ascontiguousarray(res['img'])
        out[tag + '_gt_bboxes'] = np.asarray(res['gt_bboxes'], dtype=np.float32)
        out[tag + '_scale_factor'] = np.asarray(res['scale_factor'], dtype=np.float32)
        out[tag + '_flip'] = np.array([int(bool(res['flip']))])
        out[tag + '_flip_direction'] = np.array([str(res['flip_direction'])])
        out[tag + '_rng_after'] = np.array([np.random.uniform()])
    path = os.path.join(HERE, 'geometric_reference.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


if __name__ == '__main__':
    main()
