"""generate_random_bboxes_xy through csrc/host_rng.hip (oadg_np_random_bboxes, drawing in place from numpy's global
MT19937 state) against the python trial loop that restates mmdet/models/detectors/two_stage.py:389-419: the same boxes and
the same generator state afterwards, bit for bit.  Host code only: runs without a GPU."""
import numpy as np
import pytest

from oadg_amd import detectors


def _both(seed, skip, fn):
    out = []
    for native in (False, True):
        detectors.NATIVE_RANDOM_BBOXES = native
        np.random.seed(seed)
        np.random.random_sample(skip)                       # move the stream (and the position inside its 624-word block)
        out.append((fn(), np.random.get_state()[1].copy(), np.random.get_state()[2], np.random.uniform()))
    detectors.NATIVE_RANDOM_BBOXES = True
    return out


@pytest.mark.parametrize('seed', range(12))
def test_native_random_bboxes_match_the_python_loop(seed):
    rs = np.random.RandomState(1000 + seed)
    n_gt = int(rs.randint(1, 40))
    W, H = int(rs.choice([1, 2, 257, 1024, 2048, 1333])), int(rs.choice([1, 3, 800, 1024, 2048]))
    x1, y1 = rs.uniform(0, W, n_gt), rs.uniform(0, H, n_gt)
    gts = np.stack([x1, y1, x1 + rs.uniform(1, W / 2 + 2, n_gt), y1 + rs.uniform(1, H / 2 + 2, n_gt)], 1).astype(np.float32)
    iou_max, iou_min = float(rs.choice([1.0, 0.7, 0.3, 0.05])), float(rs.choice([0.0, 0.0, 0.01]))
    num = int(rs.randint(0, 25)) if seed % 3 else (3, 12)
    kw = dict(scales=(0.01, 0.2), ratios=(0.3, 1 / 0.3), iou_max=iou_max, iou_min=iou_min,
              max_iters=int(rs.choice([500, 7])))
    ref, nat = _both(seed, int(rs.randint(0, 1300)), lambda: detectors.generate_random_bboxes_xy((W, H), num, gts, **kw))
    assert ref[0].shape == nat[0].shape and ref[0].dtype == nat[0].dtype and np.array_equal(ref[0], nat[0])
    assert np.array_equal(ref[1], nat[1]) and ref[2] == nat[2] and ref[3] == nat[3]       # state words, position, next draw


def test_native_random_bboxes_without_gt_boxes_and_with_none():
    """bboxes_xy=None: no IoU test; an empty gt array: the reference raises on np.max of an empty array - the python path
    keeps that behaviour (the native function declines the case)"""
    ref, nat = _both(3, 5, lambda: detectors.generate_random_bboxes_xy((640, 480), 9, None))
    assert np.array_equal(ref[0], nat[0]) and np.array_equal(ref[1], nat[1]) and ref[2] == nat[2]
    with pytest.raises(ValueError):
        detectors.generate_random_bboxes_xy((640, 480), 4, np.zeros((0, 4), np.float32))


def test_native_random_bboxes_are_used_by_default_and_leave_other_generators_alone():
    assert detectors.NATIVE_RANDOM_BBOXES
    gts = np.array([[10, 10, 200, 150]], np.float32)
    np.random.seed(0)
    a = detectors._random_bboxes_native(640, 480, 5, gts, (0.01, 0.2), (0.3, 1 / 0.3), 500, 0.7, 0.0)
    assert a is not None and a.shape == (5, 5) and (a[:, 4] == 1).all()
    assert (a[:, 2] <= 640).all() and (a[:, 3] <= 480).all() and (a[:, 0] >= 0).all()
