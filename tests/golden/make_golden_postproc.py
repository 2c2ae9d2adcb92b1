"""Generate tests/golden/postproc_reference.npz: the reference's ``multiclass_nms`` (core/post_processing/bbox_nms.py:8-98)
and ``bbox2result`` (core/bbox/transforms.py:118-139) on seeded inputs, with ``mmcv.ops.batched_nms`` supplied by
oracle/nms.py (un-vendored dependency, parity unpinned).  Run here only:  python tests/golden/make_golden_postproc.py
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
from inputs import postproc_inputs  # noqa: E402
from oracle import nms as ONMS  # noqa: E402

CASES = [  # seed, n boxes, classes, class-specific boxes, score_thr, iou_thr, max_num
    (0, 300, 8, True, 0.05, 0.5, 100), (1, 300, 8, False, 0.05, 0.5, 100), (2, 64, 3, True, 0.3, 0.5, -1),
    (3, 50, 8, True, 0.99, 0.5, 100), (4, 1000, 8, True, 0.02, 0.5, 100), (5, 200, 1, True, 0.1, 0.7, 20)]


def main():
    refload.install(ops=dict(batched_nms=ONMS.batched_nms, nms=ONMS.nms))
    mnms = refload.ref('mmdet.core.post_processing.bbox_nms', 'multiclass_nms')
    b2r = refload.ref('mmdet.core.bbox.transforms', 'bbox2result')
    out = {}
    for seed, n, C, per_class, thr, iou, max_num in CASES:
        boxes, scores = postproc_inputs(seed, n, C, per_class)
        dets, labels = mnms(torch.tensor(boxes), torch.tensor(scores), thr, dict(type='nms', iou_threshold=iou), max_num)
        out[f's{seed}_dets'] = dets.numpy().copy()
        out[f's{seed}_labels'] = labels.numpy().copy()
        res = b2r(dets, labels, C)
        for c, a in enumerate(res):
            out[f's{seed}_res{c}'] = np.asarray(a, dtype=np.float32)
        print(seed, dets.shape, np.bincount(labels.numpy(), minlength=C))
    np.savez_compressed(os.path.join(HERE, 'postproc_reference.npz'), **out)


if __name__ == '__main__':
    main()
