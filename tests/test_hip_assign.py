"""-m gpu parity: the fused batch MaxIoUAssigner (C ABI oadg_max_iou_assign) against the tensor-expression
assigner (core/bbox.py MaxIoUAssigner.assign_masked, itself pinned to the reference by the whole-step fixture in
test_model_parity.py).  Integer outputs and fp32 IoUs must be bit-identical."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _boxes(gen, n, dev, W=512., H=256., lo=4., hi=200.):
    w = torch.rand(n, generator=gen, device=dev) * (hi - lo) + lo
    h = torch.rand(n, generator=gen, device=dev) * (hi - lo) + lo
    x = torch.rand(n, generator=gen, device=dev) * (W - w)
    y = torch.rand(n, generator=gen, device=dev) * (H - h)
    return torch.stack([x, y, x + w, y + h], 1)


@pytest.mark.parametrize('cfg', [
    dict(pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3, match_low_quality=True),     # RPN
    dict(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, match_low_quality=False),    # R-CNN
    dict(pos_iou_thr=0.6, neg_iou_thr=(0.1, 0.4), min_pos_iou=0.0, match_low_quality=True),
])
@pytest.mark.parametrize('shared', [True, False])
def test_assign_many_is_bit_identical(dev, cfg, shared):
    from oadg_amd.core.bbox import MaxIoUAssigner
    gen = torch.Generator(device=dev).manual_seed(3)
    asg = MaxIoUAssigner(**cfg)
    N, counts = 20011, [20, 7, 0, 33]
    gts = [_boxes(gen, c, dev, lo=20.) for c in counts]
    labels = [torch.randint(0, 8, (c,), generator=gen, device=dev) for c in counts]
    if shared:
        boxes = _boxes(gen, N, dev)
        per = [boxes] * len(counts)
    else:
        per = [_boxes(gen, N, dev) for _ in counts]
        boxes = per
    # some boxes coincide exactly with gts (IoU 1 and exact ties for low-quality matching)
    for i, c in enumerate(counts):
        if c and not shared:
            per[i][:c] = gts[i]
            per[i][c:2 * c] = gts[i]
    valids = [torch.rand(N, generator=gen, device=dev) > 0.2, None, None,
              torch.rand(N, generator=gen, device=dev) > 0.5]
    out = asg.assign_many(boxes, valids, gts, labels)
    assert out is not None
    ars, cnt = out
    cnt = cnt.cpu()
    for i in range(len(counts)):
        v = valids[i] if valids[i] is not None else torch.ones(N, dtype=torch.bool, device=dev)
        ref = asg.assign_masked(per[i], v, gts[i], labels[i])
        assert torch.equal(ars[i].gt_inds, ref.gt_inds), i
        if counts[i]:
            assert torch.equal(ars[i].max_overlaps, ref.max_overlaps), i
            assert torch.equal(ars[i].labels, ref.labels), i
        assert int(cnt[i, 0]) == int((ref.gt_inds > 0).sum()) and int(cnt[i, 1]) == int((ref.gt_inds == 0).sum())
        assert ars[i].num_gts == counts[i]


def test_assign_many_rpn_scale(dev):
    """the bench shape: 523,776 anchors shared by 8 images, 20 gts each, no mask, no labels."""
    from oadg_amd.core.bbox import MaxIoUAssigner
    gen = torch.Generator(device=dev).manual_seed(5)
    asg = MaxIoUAssigner(pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3, match_low_quality=True)
    anchors = _boxes(gen, 523776, dev, W=2048., H=1024., lo=8., hi=600.)
    gts = [_boxes(gen, 20, dev, W=2048., H=1024., lo=24., hi=400.) for _ in range(8)]
    ars, cnt = asg.assign_many(anchors, None, gts, None)
    for i in (0, 7):
        ref = asg.assign(anchors, gts[i])
        assert torch.equal(ars[i].gt_inds, ref.gt_inds)
        assert torch.equal(ars[i].max_overlaps, ref.max_overlaps)
        assert ars[i].labels is None


def test_fused_rpn_targets_equal_reference_order(dev):
    """begin_targets (batch assign) + device selection + fused anchor targets against the reference order of
    operations (anchor_inside_flags, assign, sample, encode, unmap - anchor_head.py:201-297) on the GPU: every
    output tensor bit-identical, the CPU generator consumed identically.  Anchor count > 4096 per sampler draw so
    that the O(k) randperm replay is exercised as well."""
    import numpy as np
    from oadg_amd.config import ConfigDict
    from oadg_amd.dense_heads import RPNHead
    train_cfg = ConfigDict(
        assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3,
                      match_low_quality=True, ignore_iof_thr=-1),
        sampler=dict(type='RandomSampler', num=256, pos_fraction=0.5, neg_pos_ub=-1, add_gt_as_proposals=False),
        allowed_border=-1, pos_weight=-1, debug=False)
    strides, H, W = [4, 8, 16, 32], 192, 320
    head = RPNHead(in_channels=8, feat_channels=8,
                   anchor_generator=dict(type='AnchorGenerator', scales=[8], ratios=[0.5, 1.0, 2.0], strides=strides),
                   loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                   loss_bbox=dict(type='L1Loss', loss_weight=1.0), train_cfg=train_cfg).to(dev)
    rs = np.random.RandomState(7)
    sizes = [(int(np.ceil(H / s)), int(np.ceil(W / s))) for s in strides]
    metas = [dict(img_shape=(H, W, 3), pad_shape=(H, W, 3)) for _ in range(4)]
    gts = []
    for n in (9, 4, 0, 30):
        c = rs.uniform([20, 20], [W - 20, H - 20], (n, 2))
        wh = rs.uniform(12, 120, (n, 2))
        gts.append(torch.tensor(np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32), device=dev))
    out = {}
    for mode in ('reference', 'fused'):
        head.reference_order_targets = mode == 'reference'
        head._pending_targets = None
        torch.manual_seed(11)
        if mode == 'fused':
            head.begin_targets((H, W), gts, metas, dev)
            assert head._pending_targets is not None
        anchors, flags = head.get_anchors(sizes, metas, device=dev)
        out[mode] = head.get_targets(anchors, flags, gts, metas)
        out[mode + '_rng'] = torch.rand(1).item()
    a, b = out['reference'], out['fused']
    assert a[4:] == b[4:], (a[4:], b[4:])
    for la, lb in zip(a[:4], b[:4]):
        for x, y in zip(la, lb):
            assert x.shape == y.shape and torch.equal(x, y)
    assert out['reference_rng'] == out['fused_rng']


@pytest.mark.parametrize('pos_weight', [-1, 2.5])
def test_fused_roi_targets_equal_tensor_path(dev, pos_weight):
    """oadg_roi_targets (rois + labels / weights / encoded deltas / absolute gts of every sampled row, plus the roi rows
    of extra box lists) against bbox2roi + BBoxHead._get_targets_batched on real sampling results: bit-identical.
    Images with many, few and no gts; padded proposals (score -1 rows); two views sharing the sampling results."""
    import numpy as np
    from oadg_amd.config import ConfigDict
    from oadg_amd.core import bbox2roi
    from oadg_amd.core.bbox import MaxIoUAssigner, RandomSampler, sample_many_begin
    from oadg_amd.roi_heads import Shared2FCBBoxHead
    gen = torch.Generator(device=dev).manual_seed(3)
    rs = np.random.RandomState(5)
    head = Shared2FCBBoxHead(in_channels=8, fc_out_channels=16, roi_feat_size=7, num_classes=8,
                             loss_bbox=dict(type='L1Loss', loss_weight=1.0)).to(dev)
    cfg = ConfigDict(pos_weight=pos_weight)
    asg = MaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, match_low_quality=False)
    smp = RandomSampler(num=128, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=True)
    props, gts, labels = [], [], []
    for n_gt in (12, 3, 0, 40):
        g = _boxes(gen, n_gt, dev, lo=16., hi=180.)
        p = _boxes(gen, 300, dev, lo=8., hi=220.)
        if n_gt:                                            # some proposals close to gts so that positives exist
            jit = (torch.rand(60, 4, generator=gen, device=dev) - 0.5) * 8
            p[:60] = g[torch.arange(60, device=dev) % n_gt] + jit
        score = torch.rand(300, 1, generator=gen, device=dev)
        score[280:] = -1.                                   # padding rows of a fixed-size proposal list
        props.append(torch.cat([p, score], 1))
        gts.append(g)
        labels.append(torch.tensor(rs.randint(0, 8, n_gt), device=dev, dtype=torch.long))
    valids = [p[:, 4] >= 0 for p in props]
    ars, counts = asg.assign_many(props, valids, gts, labels)
    torch.manual_seed(2)
    first = sample_many_begin(smp, ars, props, gts, labels, counts=counts).finish()
    results = list(first) + list(first)                     # two views (contrastive_roi_head.py:84-97)
    extra = [_boxes(gen, n, dev) for n in (17, 0, 5)]
    extra[2] = torch.cat([extra[2], torch.ones(5, 1, device=dev)], 1)[:, :4]     # row stride 5
    head.FUSED_TARGETS = True
    fused = head.rois_and_targets(results, cfg, extra)
    assert fused is not None
    rois_all, K, t = fused
    ref_rois = bbox2roi([r.bboxes for r in results])
    ref_t = head._get_targets_batched(results, cfg)
    assert K == ref_rois.shape[0] and K > 0
    assert torch.equal(rois_all[:K], ref_rois)
    ref_extra = torch.cat([torch.cat([b.new_full((b.size(0), 1), j), b[:, :4]], 1) for j, b in enumerate(extra)])
    assert torch.equal(rois_all[K:], ref_extra)
    assert int((t[0] < 8).sum()) > 0 and int((t[0] == 8).sum()) > 0
    for a, b, name in zip(t, ref_t, ('labels', 'label_weights', 'bbox_targets', 'bbox_weights', 'absolute')):
        assert a.dtype == b.dtype and a.shape == b.shape, name
        assert torch.equal(a, b), name
    # more box lists than the launch arguments hold: the caller is told to take the tensor path
    assert head.rois_and_targets(results * 5, cfg) is None


@pytest.mark.parametrize('match_low_quality', [False, True])
def test_fused_roi_assign_add_gt_equals_tensor_path(dev, match_low_quality):
    """oadg_roi_assign_add_gt (assignment of the proposals + gts added as proposals, three launches for the batch) against
    assign_many + sample_many_begin (the per-image concatenations of base_sampler.py:38-78 / assign_result.add_gt_): the
    same boxes / gt_inds / labels / overlaps per image, the same candidate counts, and - with the same CPU generator state -
    the same sampled indices and pos_is_gt flags.  Images with many, few and no gts; padding rows (score -1)."""
    import numpy as np
    from oadg_amd.core import bbox as B
    gen = torch.Generator(device=dev).manual_seed(3)
    rs = np.random.RandomState(5)
    asg = B.MaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, match_low_quality=match_low_quality)
    smp = B.RandomSampler(num=128, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=True)
    props, gts, labels = [], [], []
    for n_gt in (12, 3, 0, 40):
        g = _boxes(gen, n_gt, dev, lo=16., hi=180.)
        p = _boxes(gen, 300, dev, lo=8., hi=220.)
        if n_gt:
            jit = (torch.rand(60, 4, generator=gen, device=dev) - 0.5) * 8
            p[:60] = g[torch.arange(60, device=dev) % n_gt] + jit
        score = torch.rand(300, 1, generator=gen, device=dev)
        score[280:] = -1.
        props.append(torch.cat([p, score], 1))
        gts.append(g)
        labels.append(torch.tensor(rs.randint(0, 8, n_gt), device=dev, dtype=torch.long))
    out = {}
    for mode in ('tensor', 'fused'):
        torch.manual_seed(2)
        if mode == 'fused':
            pend = B.roi_assign_sample_begin(asg, smp, props, gts, labels)
            assert pend is not None
        else:
            valids = [p[:, 4] >= 0 for p in props]
            ars, counts = asg.assign_many(props, valids, gts, labels)
            pend = B.sample_many_begin(smp, ars, props, gts, labels, counts=counts)
        pre = [(p[0].gt_inds.clone(), p[0].labels.clone(), p[0].max_overlaps.clone(), p[1].clone(), p[0].num_gts)
               for p in pend.prepared]
        res = pend.finish()
        out[mode] = (pre, [(r.pos_inds.clone(), r.neg_inds.clone(), r.pos_is_gt.clone(), r.bboxes.clone(),
                            r.pos_gt_labels.clone(), r.pos_assigned_gt_inds.clone()) for r in res],
                     torch.rand(1).item())
    a, b = out['tensor'], out['fused']
    assert a[2] == b[2]                                  # the CPU generator was consumed identically
    for i, (x, y) in enumerate(zip(a[0], b[0])):
        assert x[4] == y[4]
        for u, v, name in zip(x[:4], y[:4], ('gt_inds', 'labels', 'max_overlaps', 'bboxes')):
            assert u.shape == v.shape and u.dtype == v.dtype and torch.equal(u, v), (i, name)
    assert sum(int(r[0].numel()) for r in b[1]) > 0
    for i, (x, y) in enumerate(zip(a[1], b[1])):
        for u, v, name in zip(x, y, ('pos_inds', 'neg_inds', 'pos_is_gt', 'bboxes', 'pos_gt_labels', 'pos_assigned_gt_inds')):
            assert u.shape == v.shape and u.dtype == v.dtype and torch.equal(u, v), (i, name)
    # proposal lists of different lengths are outside the kernels' domain: the caller is told to loop
    assert B.roi_assign_sample_begin(asg, smp, [props[0], props[1][:200]], gts[:2], labels[:2]) is None


def test_fused_roi_path_without_any_gt(dev):
    """A batch whose images have no gt boxes at all (Gmax = 0): the fused assignment / targets take the same route as the
    tensor path - every valid proposal is background, no positives, targets all zero, labels = num_classes."""
    from oadg_amd.config import ConfigDict
    from oadg_amd.core import bbox as B
    from oadg_amd.roi_heads import Shared2FCBBoxHead
    gen = torch.Generator(device=dev).manual_seed(9)
    asg = B.MaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, match_low_quality=False)
    smp = B.RandomSampler(num=64, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=True)
    props = []
    for _ in range(2):
        p = _boxes(gen, 100, dev)
        s = torch.rand(100, 1, generator=gen, device=dev)
        s[90:] = -1.
        props.append(torch.cat([p, s], 1))
    gts = [torch.zeros(0, 4, device=dev) for _ in range(2)]
    labels = [torch.zeros(0, dtype=torch.long, device=dev) for _ in range(2)]
    out = {}
    for mode in ('tensor', 'fused'):
        torch.manual_seed(4)
        if mode == 'fused':
            pend = B.roi_assign_sample_begin(asg, smp, props, gts, labels)
            assert pend is not None
        else:
            ars, counts = asg.assign_many(props, [p[:, 4] >= 0 for p in props], gts, labels) or (None, None)
            if ars is None:          # (the batch kernel declines Gmax = 0 inputs it cannot stack: per-image assignment)
                ars = [asg.assign_masked(p[:, :4], p[:, 4] >= 0, g, l) for p, g, l in zip(props, gts, labels)]
            pend = B.sample_many_begin(smp, ars, props, gts, labels, counts=counts)
        res = pend.finish()
        out[mode] = [(r.pos_inds.clone(), r.neg_inds.clone(), r.bboxes.clone()) for r in res]
    for x, y in zip(out['tensor'], out['fused']):
        assert x[0].numel() == 0 and y[0].numel() == 0
        assert torch.equal(x[1], y[1]) and torch.equal(x[2], y[2]) and x[1].numel() == 64
    head = Shared2FCBBoxHead(in_channels=8, fc_out_channels=16, roi_feat_size=7, num_classes=8).to(dev)
    fused = head.rois_and_targets(res, ConfigDict(pos_weight=-1))
    assert fused is not None
    rois, K, t = fused
    assert K == 128 and torch.equal(rois, B.bbox2roi([r.bboxes for r in res]))
    assert int((t[0] != 8).sum()) == 0 and float(t[2].abs().sum()) == 0 and float(t[3].abs().sum()) == 0
    assert torch.equal(t[1], torch.ones(128, device=dev))
