"""Inference path against fixtures produced by the reference (tests/golden/make_golden_postproc.py,
make_golden_model.py test): multiclass_nms / bbox2result, and the whole ``simple_test`` of the detector.
CPU: host logic with the oracle's NMS / RoIAlign (oracle.backend.oracle_ops).  GPU: the product path."""
import os

import numpy as np
import pytest
import torch

from inputs import model_batch, named_weights, postproc_inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, 'configs', 'oadg', 'faster_rcnn_r50_fpn_1x_cityscapes_oadg.py')
CASES = [(0, 300, 8, True, 0.05, 0.5, 100), (1, 300, 8, False, 0.05, 0.5, 100), (2, 64, 3, True, 0.3, 0.5, -1),
         (3, 50, 8, True, 0.99, 0.5, 100), (4, 1000, 8, True, 0.02, 0.5, 100), (5, 200, 1, True, 0.1, 0.7, 20)]


def _check_postproc(g, device):
    from oadg_amd.core import bbox2result, multiclass_nms
    for seed, n, C, per_class, thr, iou, max_num in CASES:
        boxes, scores = postproc_inputs(seed, n, C, per_class)
        dets, labels, inds = multiclass_nms(torch.tensor(boxes, device=device), torch.tensor(scores, device=device), thr,
                                            dict(type='nms', iou_threshold=iou), max_num, return_inds=True)
        assert np.array_equal(dets.cpu().numpy(), g[f's{seed}_dets']), seed
        assert np.array_equal(labels.cpu().numpy(), g[f's{seed}_labels']), seed
        assert len(inds) == len(labels)
        for c, a in enumerate(bbox2result(dets, labels, C)):
            assert a.dtype == np.float32 and np.array_equal(a, g[f's{seed}_res{c}'])


def test_multiclass_nms_host_logic_vs_reference(golden_dir):
    from oracle.backend import oracle_ops
    with oracle_ops():
        _check_postproc(np.load(os.path.join(golden_dir, 'postproc_reference.npz')), 'cpu')


@pytest.mark.gpu
def test_multiclass_nms_device_vs_reference(dev, golden_dir):
    _check_postproc(np.load(os.path.join(golden_dir, 'postproc_reference.npz')), dev)


def _build(device):
    from oadg_amd import Config, build_detector
    det = build_detector(Config.fromfile(CFG).model)
    w = named_weights({k: v.shape for k, v in det.state_dict().items()})
    det.load_state_dict({k: torch.as_tensor(v) for k, v in w.items()})
    return det.to(device).eval()


def _metas(g):
    h, w, n = int(g['h']), int(g['w']), int(g['n_img'])
    return [dict(img_shape=(h, w, 3), pad_shape=(h, w, 3), ori_shape=(int(h / 1.25), int(w / 1.25), 3),
                 scale_factor=g['scale_factor'], flip=False, ori_filename=f'{i}.png') for i in range(n)]


def test_simple_test_host_logic_vs_reference(golden_dir, monkeypatch):
    """unfolded BN (operation for operation the reference's network): proposals and detections to 1e-4 abs."""
    from oracle.backend import oracle_ops
    from oadg_amd import layers
    monkeypatch.setattr(layers, 'FOLD_EVAL_BN', False)
    g = np.load(os.path.join(golden_dir, 'model_test_256x512.npz'))
    det = _build('cpu')
    img = torch.tensor(model_batch(int(g['seed']), int(g['n_img']), int(g['h']), int(g['w']))['img'])
    with oracle_ops(), torch.no_grad():
        props = det.rpn_head.simple_test_rpn(det.extract_feat(img), _metas(g))
        res = det(img=[img], img_metas=[_metas(g)], return_loss=False, rescale=True)
    for i in range(int(g['n_img'])):
        assert props[i].shape == g[f'proposals{i}'].shape
        assert np.abs(props[i].numpy() - g[f'proposals{i}']).max() <= 1e-3
        assert len(res[i]) == 8
        for c in range(8):
            ref = g[f'det{i}_c{c}']
            assert res[i][c].shape == ref.shape, (i, c, res[i][c].shape, ref.shape)
            if len(ref):
                assert np.abs(res[i][c] - ref).max() <= 1e-3


@pytest.mark.gpu
def test_simple_test_on_gpu_close_to_reference(dev, golden_dir, monkeypatch):
    """fp32 product path (HIP RoIAlign / NMS, MIOpen convs): discrete NMS decisions may flip on ulp differences, so
    the bar is the detection count per class within 5 and the top-scored boxes matching to 1e-2."""
    monkeypatch.setattr(torch.backends.cudnn, 'deterministic', True)
    g = np.load(os.path.join(golden_dir, 'model_test_256x512.npz'))
    det = _build(dev)
    img = torch.tensor(model_batch(int(g['seed']), int(g['n_img']), int(g['h']), int(g['w']))['img'], device=dev)
    res = det(img=[img], img_metas=[_metas(g)], return_loss=False, rescale=True)
    for i in range(int(g['n_img'])):
        for c in range(8):
            ref = g[f'det{i}_c{c}']
            assert abs(len(res[i][c]) - len(ref)) <= 5
            k = min(10, len(ref), len(res[i][c]))
            if k:
                assert np.abs(res[i][c][:k, 4] - ref[:k, 4]).max() <= 1e-3
