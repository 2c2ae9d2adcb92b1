"""CPU: the COCO-style bbox evaluator (oadg_amd/evaluation.py, restating pycocotools' COCOeval which is not installed)
against hand-computed cases, and the corruption-benchmark bookkeeping of tools/analysis_tools/test_robustness.py /
robustness_eval.py."""
import numpy as np
import pytest

import oadg_amd  # noqa: F401
from oadg_amd import evaluation as E


def _gt(x1, y1, x2, y2, cat=0, crowd=0):
    return dict(bbox=[x1, y1, x2 - x1, y2 - y1], category=cat, area=float((x2 - x1) * (y2 - y1)), iscrowd=crowd)


def _res(dets, num_classes=1, cat=0):
    out = [np.zeros((0, 5), np.float32) for _ in range(num_classes)]
    out[cat] = np.asarray(dets, np.float32).reshape(-1, 5)
    return out


def test_perfect_detection_and_area_buckets():
    r = E.coco_eval_bbox([[_gt(0, 0, 100, 100)]], [_res([[0, 0, 100, 100, .9]])], 1)
    for k in ('AP', 'AP50', 'AP75', 'APl'):            # (1 - 2e-16: COCOeval divides by tp + fp + eps)
        assert r[k] == pytest.approx(1.0, abs=1e-12)
    assert r['APs'] == r['APm'] == -1.0                      # no ground truth of that size: undefined
    assert r['AR1'] == r['AR10'] == r['AR100'] == r['ARl'] == 1.0


def test_iou_082_counts_for_seven_of_ten_thresholds():
    # IoU(gt 100x100, det 100x82) = 0.82: a true positive at .50 ... .80, a false positive at .85, .90, .95
    r = E.coco_eval_bbox([[_gt(0, 0, 100, 100)]], [_res([[0, 0, 100, 82, .9]])], 1)
    assert r['AP'] == pytest.approx(0.7) and r['AP50'] == pytest.approx(1.0) and r['AP75'] == pytest.approx(1.0) and r['AR100'] == pytest.approx(0.7)


def test_precision_envelope_and_101_point_interpolation():
    # ranked detections: TP (.9, image 0), FP (.8, image 0), TP (.7, image 1); two ground truths.
    # recall = [.5, .5, 1], precision = [1, .5, 2/3] -> envelope [1, 2/3, 2/3]; the 51 recall points 0 ... .50 read 1,
    # the 50 points .51 ... 1.00 read 2/3:  AP = (51 + 50 * 2/3) / 101
    gts = [[_gt(0, 0, 10, 10)], [_gt(0, 0, 10, 10)]]
    res = [_res([[0, 0, 10, 10, .9], [50, 50, 60, 60, .8]]), _res([[0, 0, 10, 10, .7]])]
    r = E.coco_eval_bbox(gts, res, 1)
    exp = (51 + 50 * 2.0 / 3.0) / 101
    assert r['AP'] == pytest.approx(exp, abs=1e-9) and r['AP50'] == pytest.approx(exp, abs=1e-9)
    assert r['APs'] == pytest.approx(exp, abs=1e-9) and r['APm'] == -1.0
    assert r['AR1'] == 1.0 and r['AR100'] == 1.0            # maxDets = 1 keeps the best detection PER IMAGE


def test_detection_inside_a_crowd_region_is_ignored():
    gts = [[_gt(0, 0, 50, 50), _gt(100, 100, 200, 200, crowd=1)]]
    hi = [[110, 110, 150, 150, .95], [0, 0, 50, 50, .9]]     # the top-scored detection lies inside the crowd region
    assert E.coco_eval_bbox(gts, [_res(hi)], 1)['AP'] == pytest.approx(1.0)  # matched to the crowd (IoU = inter / det area): ignored
    no_crowd = [[_gt(0, 0, 50, 50)]]
    assert E.coco_eval_bbox(no_crowd, [_res(hi)], 1)['AP'] == pytest.approx(0.5)   # same boxes without it: a false positive first


def test_unmatched_detection_outside_the_area_range_is_ignored_there():
    gts = [[_gt(0, 0, 40, 40)]]                               # 1600 px^2: medium
    res = [_res([[200, 200, 210, 210, .95], [0, 0, 40, 40, .9]])]      # a small false positive ranked first
    r = E.coco_eval_bbox(gts, res, 1)
    assert r['AP'] == pytest.approx(0.5) and r['APm'] == pytest.approx(1.0) and r['APs'] == -1.0 and r['APl'] == -1.0


def test_classes_are_evaluated_separately_and_averaged():
    gts = [[_gt(0, 0, 100, 100, cat=0), _gt(200, 200, 300, 300, cat=1)]]
    res = [[np.array([[0, 0, 100, 100, .9]], np.float32), np.array([[0, 0, 100, 100, .9]], np.float32),
            np.zeros((0, 5), np.float32)]]                    # class 1 predicts the class-0 box: a miss; class 2 has nothing
    r = E.coco_eval_bbox(gts, res, 3)
    assert r['AP'] == pytest.approx(0.5) and r['AR100'] == pytest.approx(0.5)    # mean over the two classes WITH ground truth


def test_robustness_bookkeeping():
    assert E.select_corruptions(['benchmark'], [0, 1, 2, 3, 4, 5])[0] == E.CORRUPTION_SETS['all'][:15]
    assert len(E.select_corruptions(['all'], [0, 1])[0]) == 19
    assert E.select_corruptions(['None'], [0, 1, 2]) == (['None'], [0])
    assert E.select_corruptions(['fog', 'snow'], [0, 3]) == (['fog', 'snow'], [0, 3])
    assert E.corrupted_img_prefix('/ws/data/cityscapes/leftImg8bit/val/', 'fog', 3) == \
        '/ws/data/cityscapes-c/leftImg8bit/val/fog/3/'
    with pytest.raises(NotImplementedError):
        E.corrupted_img_prefix('/data/other/', 'fog', 1)
    out = {}
    for ci, c in enumerate(['gaussian_noise', 'fog']):
        out[c] = {s: dict(bbox={m: (0.4 if s == 0 else 0.4 - 0.05 * s - 0.1 * ci) for m in E.METRICS}) for s in range(6)}
    agg = E.aggregate_robustness(out)
    assert agg['P']['AP'] == pytest.approx(0.4)
    exp = np.mean([[0.4 - 0.05 * s - 0.1 * ci for s in range(1, 6)] for ci in range(2)])
    assert agg['mPC']['AP'] == pytest.approx(exp, abs=1e-6) and agg['rPC']['AP'] == pytest.approx(exp / 0.4, abs=1e-6)


def test_mmdet_style_max_dets_and_names():
    """CocoDataset.evaluate (coco.py:468-480) sets maxDets = (100, 300, 1000): 150 equally good detections of 150 objects
    give AR@100 = 100 / 150 under COCOeval's default cut (AR100) but the full recall at 300 / 1000, and mAP is taken at
    1000 detections."""
    gts = [[_gt(10 * i, 0, 10 * i + 8, 8) for i in range(150)]]
    dets = [[10 * i, 0, 10 * i + 8, 8, 0.9 - 1e-4 * i] for i in range(150)]
    d = E.coco_eval_bbox(gts, [_res(dets)], 1)
    assert d['AR100'] == pytest.approx(100 / 150) and d['AP'] < 0.7
    m = E.coco_eval_bbox(gts, [_res(dets)], 1, max_dets=E.MMDET_MAX_DETS, names=E.MMDET_METRICS)
    assert list(m) == E.MMDET_METRICS
    assert m['AR@100'] == pytest.approx(100 / 150) and m['AR@300'] == pytest.approx(1.0) and m['AR@1000'] == pytest.approx(1.0)
    assert m['mAP'] == pytest.approx(1.0) and m['mAP_50'] == pytest.approx(1.0)
