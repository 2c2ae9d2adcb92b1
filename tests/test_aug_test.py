"""Test-time augmentation (SURVEY.md 8f.4 remainder): MultiScaleFlipAug with several scales / flips -> forward_test ->
aug_test (two_stage.py:268-277, dense_test_mixins.py:135-167, test_mixins.py:139-177, merge_augs.py:13-112)."""
import os

import numpy as np
import pytest
import torch

import oadg_amd  # noqa: F401
from oadg_amd.core import bbox_flip, bbox_mapping, bbox_mapping_back, merge_aug_bboxes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_box_mappings_are_inverse_and_flip_is_an_involution():
    rs = np.random.RandomState(0)
    b = torch.tensor(rs.uniform(0, 200, (7, 8)).astype(np.float32))
    b[:, 2::4] += b[:, 0::4]
    b[:, 3::4] += b[:, 1::4]
    shape = (480, 640, 3)
    for d in ('horizontal', 'vertical', 'diagonal'):
        f = bbox_flip(b, shape, d)
        assert torch.allclose(bbox_flip(f, shape, d), b, atol=1e-4)
        assert (f[:, 2::4] >= f[:, 0::4]).all() and (f[:, 3::4] >= f[:, 1::4]).all()
    assert torch.equal(bbox_flip(b, shape, 'horizontal')[:, 0], 640 - b[:, 2])
    sf = np.array([1.5, 0.75, 1.5, 0.75], np.float32)
    b4 = b[:, :4]
    for flip, d in ((False, None), (True, 'horizontal'), (True, 'vertical'), (True, 'diagonal')):
        m = bbox_mapping(b4, shape, sf, flip, d or 'horizontal')
        back = bbox_mapping_back(m, shape, sf, flip, d or 'horizontal')
        assert torch.allclose(back, b4, atol=1e-3)
    # hand-computed: box (10, 20, 30, 60) scaled by 2 inside a 200-wide test image, flipped horizontally
    m = bbox_mapping(torch.tensor([[10., 20., 30., 60.]]), (100, 200, 3), np.array([2, 2, 2, 2], np.float32), True)
    assert m.tolist() == [[140., 40., 180., 120.]]


def test_merge_aug_bboxes_averages_in_the_original_frame():
    boxes = torch.tensor([[10., 10., 50., 40., 12., 8., 52., 44.]])        # one RoI, two classes
    metas = [[dict(img_shape=(100, 200, 3), scale_factor=np.ones(4, np.float32), flip=False, flip_direction=None)],
             [dict(img_shape=(200, 400, 3), scale_factor=np.full(4, 2, np.float32), flip=True, flip_direction='horizontal')]]
    aug2 = bbox_mapping((boxes + 2.0).view(-1, 4), (200, 400, 3), np.full(4, 2, np.float32), True).view(1, 8)   # 2 px off
    s1, s2 = torch.tensor([[0.2, 0.5, 0.3]]), torch.tensor([[0.4, 0.1, 0.5]])
    mb, ms = merge_aug_bboxes([boxes, aug2], [s1, s2], metas, None)
    assert torch.allclose(mb, boxes + 1.0, atol=1e-4)
    assert torch.allclose(ms, torch.tensor([[0.3, 0.3, 0.4]]))


@pytest.mark.gpu
def test_merge_aug_proposals_suppresses_duplicates_across_augmentations(dev):
    from oadg_amd.core import merge_aug_proposals
    rs = np.random.RandomState(1)
    xy = rs.uniform(0, 300, (40, 2)).astype(np.float32)
    p = torch.tensor(np.concatenate([xy, xy + rs.uniform(20, 80, (40, 2)).astype(np.float32),
                                     rs.uniform(0.1, 1, (40, 1)).astype(np.float32)], 1), device=dev)
    shape1, shape2 = (400, 400, 3), (800, 800, 3)
    m1 = dict(img_shape=shape1, scale_factor=np.ones(4, np.float32), flip=False, flip_direction=None)
    m2 = dict(img_shape=shape2, scale_factor=np.full(4, 2, np.float32), flip=True, flip_direction='horizontal')
    p2 = p.clone()
    p2[:, :4] = bbox_mapping(p[:, :4], shape2, m2['scale_factor'], True)
    p2[:, 4] *= 0.9                                                            # the copy scores lower: the originals win
    cfg = dict(nms=dict(type='nms', iou_threshold=0.7), max_per_img=1000)
    merged = merge_aug_proposals([p, p2], [m1, m2], cfg)
    single = merge_aug_proposals([p], [m1], cfg)
    assert merged.shape == single.shape and torch.allclose(merged, single, atol=1e-3)
    assert (merged[:-1, 4] >= merged[1:, 4]).all()
    top5 = merge_aug_proposals([p, p2], [m1, m2], dict(cfg, max_per_img=5))
    assert top5.shape[0] == 5 and torch.equal(top5, merged[:5])


@pytest.mark.gpu
def test_aug_test_end_to_end_two_scales_and_flip(dev):
    """The detector through the reference's test pipeline with MultiScaleFlipAug(img_scale=[...two scales...], flip=True):
    4 augmentations -> forward_test -> aug_test; result format of simple_test, boxes in the ORIGINAL image frame
    (rescale=True); a single augmentation still takes the simple_test path and the flip-only merge of a detector is
    consistent with mirroring the input."""
    from oadg_amd import Config, build_detector, hip_conv
    from oadg_amd.apis import set_random_seed
    from oadg_amd.pipelines import DevicePipeline, SyntheticCityscapes
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'oadg', 'faster_rcnn_r50_fpn_1x_cityscapes_oadg.py'))
    set_random_seed(0)
    det = build_detector(cfg.model, test_cfg=cfg.get('test_cfg'))
    det.init_weights(allow_missing_pretrained=True)
    det = det.to(dev).to(memory_format=torch.channels_last).eval()
    norm = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
    inner = [dict(type='Resize', keep_ratio=True), dict(type='RandomFlip'), dict(type='Normalize', **norm),
             dict(type='Pad', size_divisor=32), dict(type='ImageToTensor', keys=['img']), dict(type='Collect', keys=['img'])]
    pipe = DevicePipeline([dict(type='LoadImageFromFile'),
                           dict(type='MultiScaleFlipAug', img_scale=[(512, 256), (640, 320)], flip=True, transforms=inner)],
                          dtype=torch.float32)
    ds = SyntheticCityscapes(img_shape=(256, 512), num_boxes=6, box_size=(24, 160), device=dev)
    imgs, _, _ = ds.batch([0])
    data = pipe.test_batch(imgs)
    assert len(data['img']) == 4 and len(data['img_metas']) == 4
    flips = [(m[0]['flip'], m[0]['flip_direction']) for m in data['img_metas']]
    assert flips == [(False, None), (True, 'horizontal'), (False, None), (True, 'horizontal')]
    assert [tuple(t.shape[2:]) for t in data['img']] == [(256, 512), (256, 512), (320, 640), (320, 640)]
    assert torch.equal(data['img'][1], data['img'][0].flip(3))                 # the flipped view of the same scale
    with torch.no_grad():
        res = det(return_loss=False, rescale=True, **data)
    assert len(res) == 1 and len(res[0]) == 8
    allb = np.concatenate(res[0])
    assert allb.shape[1] == 5 and np.isfinite(allb).all()
    assert (allb[:, 0] >= -1e-3).all() and (allb[:, 2] <= 512 + 1e-3).all() and (allb[:, 3] <= 256 + 1e-3).all()
    assert len(allb) <= det.roi_head.test_cfg.max_per_img
    one = DevicePipeline([dict(type='LoadImageFromFile'),
                          dict(type='MultiScaleFlipAug', img_scale=(512, 256), flip=False, transforms=inner)],
                         dtype=torch.float32).test_batch(imgs)
    with torch.no_grad():
        res1 = det(return_loss=False, rescale=True, **one)
    assert len(one['img']) == 1 and len(res1) == 1 and len(res1[0]) == 8


def test_aug_test_host_logic_vs_reference_fixture(golden_dir, monkeypatch):
    """``aug_test`` pinned by the reference.  tests/golden/model_aug_test_256x512.npz holds what the
    GENUINE ``TwoStageDetector.aug_test`` (two_stage.py:268-277 -> aug_test_rpn / merge_aug_proposals -> aug_test_bboxes /
    merge_aug_bboxes -> multiclass_nms) returns for four augmentations (two scales x horizontal flip) of one seeded image
    with name-seeded weights (make_golden_model.py aug).  Host logic + oracle ops, unfolded BN: merged proposals and
    detections to 1e-3 abs, same counts per class."""
    from make_golden_model import aug_inputs
    from oracle.backend import oracle_ops
    from oadg_amd import layers
    from test_inference_path import _build
    monkeypatch.setattr(layers, 'FOLD_EVAL_BN', False)
    g = np.load(os.path.join(golden_dir, 'model_aug_test_256x512.npz'))
    det = _build('cpu')
    imgs, metas = aug_inputs(int(g['seed']), int(g['h']), int(g['w']))
    with oracle_ops(), torch.no_grad():
        feats = [det.extract_feat(im) for im in imgs]
        props = det.rpn_head.aug_test_rpn(feats, metas)
        res = det(img=imgs, img_metas=metas, return_loss=False, rescale=True)
    assert len(props) == 1 and props[0].shape == g['merged_proposals'].shape
    # (the reference ranks the merged proposals with an unstable sort, merge_augs.py:78-80: rows with EXACTLY equal scores
    #  - two here - come out in either order; compare in a canonical order of (score desc, x1, y1))
    canon = lambda a: a[np.lexsort((a[:, 1], a[:, 0], -a[:, 4]))]  # noqa: E731
    assert np.abs(canon(props[0].numpy()) - canon(g['merged_proposals'])).max() <= 1e-3
    assert len(res) == 1 and len(res[0]) == 8
    n_det = 0
    for c in range(8):
        ref = g[f'det_c{c}']
        assert res[0][c].shape == ref.shape, (c, res[0][c].shape, ref.shape)
        n_det += len(ref)
        if len(ref):
            assert np.abs(res[0][c] - ref).max() <= 1e-3, c
    assert n_det == 100
