"""CPU: the single-read target path of the RPN head (masked assignment + deferred sampling) against the
reference order of operations (filter by anchor_inside_flags, assign, sample, unmap - anchor_head.py:201-297),
including allowed_border >= 0 (R101-DC5 config) where anchors outside the image are dropped."""
import numpy as np
import pytest
import torch

from oadg_amd.config import ConfigDict
from oadg_amd.dense_heads import RPNHead


def _head(allowed_border, strides, scales):
    train_cfg = ConfigDict(
        assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3,
                      match_low_quality=True, ignore_iof_thr=-1),
        sampler=dict(type='RandomSampler', num=64, pos_fraction=0.5, neg_pos_ub=-1, add_gt_as_proposals=False),
        allowed_border=allowed_border, pos_weight=-1, debug=False)
    return RPNHead(in_channels=8, feat_channels=8,
                   anchor_generator=dict(type='AnchorGenerator', scales=scales, ratios=[0.5, 1.0, 2.0],
                                         strides=strides),
                   loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                   loss_bbox=dict(type='L1Loss', loss_weight=1.0), train_cfg=train_cfg)


@pytest.mark.parametrize('allowed_border,strides,scales,pad', [(-1, [4, 8, 16], [8], (96, 128)),
                                                               (0, [16], [2, 4, 8], (96, 128)),
                                                               (0, [8, 16], [4], (100, 140))])
def test_single_read_targets_equal_reference_order(allowed_border, strides, scales, pad):
    rs = np.random.RandomState(3)
    head = _head(allowed_border, strides, scales)
    H, W = pad
    sizes = [(int(np.ceil(H / s)), int(np.ceil(W / s))) for s in strides]
    metas = [dict(img_shape=(H - 4, W - 6, 3), pad_shape=(H, W, 3)) for _ in range(3)]
    gts = []
    for _ in range(3):
        c = rs.uniform(10, 90, (5, 2)); wh = rs.uniform(8, 60, (5, 2))
        gts.append(torch.tensor(np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)))
    gts[2] = gts[2][:0]      # an image without ground truth
    out = {}
    for mode in ('reference', 'single_read', 'begin_finish'):
        head.reference_order_targets = mode == 'reference'
        head._pending_targets = None
        torch.manual_seed(11)
        if mode == 'begin_finish':
            head.begin_targets((H, W), gts, metas, 'cpu')
        anchors, flags = head.get_anchors(sizes, metas, device='cpu')
        out[mode] = head.get_targets(anchors, flags, gts, metas)
        out[mode + '_rng'] = torch.rand(1).item()
    for mode in ('single_read', 'begin_finish'):
        a, b = out['reference'], out[mode]
        assert a[4:] == b[4:], 'num_total_pos / num_total_neg differ'
        for la, lb in zip(a[:4], b[:4]):
            for x, y in zip(la, lb):
                assert torch.equal(x, y)
        assert out['reference_rng'] == out[mode + '_rng'], 'the torch generator was consumed differently'
