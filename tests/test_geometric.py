"""Resize / RandomFlip (pipelines/geometric.py) against the fixture produced by the GENUINE reference transforms
(tests/golden/make_golden_geometric.py).  CPU: draws, rounding, scale factors, box arithmetic, numpy stream state.
GPU: the whole transform incl. pixels (csrc/imgxform.hip vs the oracle's cv2.resize restatement that produced the
fixture's pixels - resize pixels are parity-unpinned)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
from inputs import lowpass_image, synthetic_boxes  # noqa: E402

CASES = [
    (0, 64, 128, dict(img_scale=[(128, 50), (128, 64)], keep_ratio=True), dict(flip_ratio=0.5)),
    (1, 64, 128, dict(img_scale=[(128, 50), (128, 64)], keep_ratio=True), dict(flip_ratio=0.5)),
    (2, 60, 100, dict(img_scale=(160, 90), keep_ratio=True), dict(flip_ratio=0.5)),
    (3, 60, 100, dict(img_scale=(96, 48), keep_ratio=False), dict(flip_ratio=[0.3, 0.3], direction=['horizontal', 'vertical'])),
    (4, 48, 80, dict(img_scale=[(100, 60), (80, 48), (120, 70)], multiscale_mode='value', keep_ratio=True),
     dict(flip_ratio=0.9, direction=['horizontal', 'vertical', 'diagonal'])),
    (5, 48, 80, dict(img_scale=(80, 48), ratio_range=(0.8, 1.4), keep_ratio=True), dict(flip_ratio=0.5)),
]


def _inputs(seed, H, W):
    rs = np.random.RandomState(200 + seed)
    return lowpass_image(rs, H, W), synthetic_boxes(rs, 5, H, W, 6, min(H, W) // 2)


@pytest.mark.parametrize('case', CASES, ids=[f's{c[0]}' for c in CASES])
def test_host_logic_matches_reference(golden_dir, case):
    from oadg_amd.pipelines.geometric import RandomFlip, Resize
    g = np.load(os.path.join(golden_dir, 'geometric_reference.npz'))
    seed, H, W, rk, fk = case
    img, gts = _inputs(seed, H, W)
    rz, fl = Resize(**rk), RandomFlip(**fk)
    np.random.seed(seed)
    scale, Wn, Hn, sf = rz.plan(H, W)
    b = rz.resize_bboxes(gts, sf, Hn, Wn)
    cur = fl.draw()
    if cur is not None:
        b = fl.bbox_flip(b, (Hn, Wn), cur)
    tag = f's{seed}'
    assert (Hn, Wn) == g[tag + '_img'].shape[:2]
    assert np.array_equal(sf, g[tag + '_scale_factor'])
    assert bool(g[tag + '_flip'][0]) == (cur is not None) and str(cur) == str(g[tag + '_flip_direction'][0])
    assert np.array_equal(b, g[tag + '_gt_bboxes'])
    assert np.random.uniform() == float(g[tag + '_rng_after'][0])


@pytest.mark.gpu
@pytest.mark.parametrize('case', CASES, ids=[f's{c[0]}' for c in CASES])
def test_device_transforms_match_fixture(dev, golden_dir, case):
    from oadg_amd.pipelines.geometric import RandomFlip, Resize
    g = np.load(os.path.join(golden_dir, 'geometric_reference.npz'))
    seed, H, W, rk, fk = case
    img, gts = _inputs(seed, H, W)
    np.random.seed(seed)
    im, b, m1 = Resize(**rk)(torch.from_numpy(img).to(dev), gts)
    im, b, m2 = RandomFlip(**fk)(im, b)
    tag = f's{seed}'
    assert np.array_equal(im.cpu().numpy(), g[tag + '_img'])
    assert np.array_equal(b, g[tag + '_gt_bboxes'])
    assert m2['flip'] == bool(g[tag + '_flip'][0])


@pytest.mark.gpu
def test_resize_kernel_equals_oracle_restatement(dev):
    from oadg_amd import _lib
    from oracle import cvleaves as cv
    rs = np.random.RandomState(0)
    L = _lib.lib()
    for (H, W, Hn, Wn) in ((37, 53, 61, 80), (128, 256, 100, 200), (64, 64, 64, 64), (50, 90, 17, 31), (9, 7, 40, 33)):
        img = rs.randint(0, 256, (H, W, 3), dtype=np.uint8)
        src = torch.from_numpy(img).to(dev)
        dst = torch.empty((Hn, Wn, 3), dtype=torch.uint8, device=dev)
        _lib.check(L.oadg_resize_bilinear_u8(_lib.ptr(src), H, W, 3, _lib.ptr(dst), Hn, Wn, _lib.stream_ptr()), 'resize')
        assert np.array_equal(dst.cpu().numpy(), cv.resize_u8_cv2(img, (Wn, Hn))), (H, W, Hn, Wn)
        for d, ax in ((1, 1), (2, 0), (3, (0, 1))):
            out = torch.empty_like(src)
            _lib.check(L.oadg_flip_u8(_lib.ptr(src), H, W, 3, _lib.ptr(out), d, _lib.stream_ptr()), 'flip')
            assert np.array_equal(out.cpu().numpy(), np.flip(img, ax))


@pytest.mark.gpu
def test_reference_pipeline_list_on_device(dev):
    """The reference's own train pipeline list (configs/OA-DG/.../*_oadg.py: LoadImageFromFile .. Collect) through
    DevicePipeline: Resize + RandomFlip + OAMix + Normalize + Pad == the oracle's composition on the same numpy stream."""
    from oadg_amd.pipelines import DevicePipeline
    from oracle import cvleaves as cv
    from oracle import oamix as OO
    H, W = 192, 384
    norm_cfg = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
    pipeline = [dict(type='LoadImageFromFile'), dict(type='LoadAnnotations', with_bbox=True),
                dict(type='Resize', img_scale=(320, 160), keep_ratio=True), dict(type='RandomFlip', flip_ratio=0.5),
                dict(type='OAMix', version='augmix', num_views=2, keep_orig=True, severity=10,
                     random_box_ratio=(3, 1 / 3), random_box_scale=(0.01, 0.1), oa_random_box_scale=(0.005, 0.1),
                     oa_random_box_ratio=(3, 1 / 3), spatial_ratio=4, sigma_ratio=0.3),
                dict(type='Normalize', **norm_cfg), dict(type='Pad', size_divisor=32), dict(type='DefaultFormatBundle'),
                dict(type='Collect', keys=['img', 'img2', 'gt_bboxes', 'gt_bboxes2', 'gt_labels', 'multilevel_boxes',
                                           'oamix_boxes'])]
    rs = np.random.RandomState(5)
    img, gts = lowpass_image(rs, H, W, 4), synthetic_boxes(rs, 4, H, W, 24, 120)
    for seed in (0, 1, 2):       # covers flip and no-flip
        np.random.seed(seed)
        Wn, Hn, _ = cv.imrescale_size(W, H, (320, 160))
        r = cv.resize_u8_cv2(img, (Wn, Hn))
        sf = np.array([Wn / W, Hn / H, Wn / W, Hn / H], np.float32)
        b = gts * sf
        b[:, 0::2] = np.clip(b[:, 0::2], 0, Wn)
        b[:, 1::2] = np.clip(b[:, 1::2], 0, Hn)
        flip = np.random.choice(['horizontal', None], p=[0.5, 0.5]) is not None
        if flip:
            r = np.ascontiguousarray(r[:, ::-1])
            f = b.copy()
            f[:, 0], f[:, 2] = Wn - b[:, 2], Wn - b[:, 0]
            b = f
        oracle = OO.OAMixOracle(version='augmix')
        ref = oracle(dict(img=r.copy(), gt_bboxes=b.copy()))
        fg = [t[1] for t in oracle.trace if t[0] == 'fg_scores'][0]
        if any(abs(s - 10) < 0.2 for s in fg if s >= 0):
            continue
        pipe = DevicePipeline(pipeline, dtype=torch.float32)
        np.random.seed(seed)
        out = pipe(torch.from_numpy(img[None]).to(dev), [gts], [np.zeros(len(gts), np.int64)])
        mean = np.array(norm_cfg['mean'], np.float32)
        stdinv = (1.0 / np.array(norm_cfg['std'], np.float64)).astype(np.float32)

        def norm(u8):
            return ((u8[..., ::-1].astype(np.float32) - mean) * stdinv).transpose(2, 0, 1)
        Hp, Wp = out['img'].shape[2:]
        assert (Hp, Wp) == ((Hn + 31) // 32 * 32, (Wn + 31) // 32 * 32)
        assert np.array_equal(out['img'][0, :, :Hn, :Wn].cpu().numpy(), norm(r))
        assert np.array_equal(out['img2'][0, :, :Hn, :Wn].cpu().numpy(), norm(ref['img2']))
        assert np.array_equal(out['gt_bboxes'][0].cpu().numpy(), b)
        m = out['img_metas'][0]
        assert m['flip'] == flip and m['img_shape'] == (Hn, Wn, 3) and np.array_equal(m['scale_factor'], sf)


@pytest.mark.gpu
def test_per_sample_multiscale_batch_like_the_reference_collate(dev):
    """The reference's train pipeline draws ONE SCALE PER SAMPLE (Resize._random_scale, transforms.py:177-243); Pad
    rounds every sample up to a multiple of 32 on its own and mmcv's collate pads the batch to the largest sample.
    DevicePipeline (default one_scale_per_batch=False) must do the same: per-image img_shape / pad_shape in img_metas,
    zeros right / below every image; the RPN then ignores anchors outside an image's pad_shape (valid_flags,
    anchor_head.py:171-199) and clips the proposals to ITS img_shape (rpn_head.py:168-171)."""
    import os
    from oadg_amd import Config, build_detector, hip_conv
    from oadg_amd.apis import TrainEngine, build_optimizer, set_random_seed
    from oadg_amd.detectors import integrate_data
    from oadg_amd.pipelines import DevicePipeline, SyntheticCityscapes
    from oadg_amd.pipelines.geometric import Resize
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, 'configs/oadg/faster_rcnn_r50_fpn_1x_cityscapes_oadg_multiscale.py'))
    pipeline = [dict(t) for t in cfg.data.train.pipeline]
    ri = next(i for i, t in enumerate(pipeline) if t['type'] == 'Resize')
    pipeline[ri] = dict(type='Resize', img_scale=[(640, 200), (640, 320)], keep_ratio=True)
    pipeline = [t for t in pipeline if t['type'] != 'RandomFlip']
    ds = SyntheticCityscapes(img_shape=(256, 512), num_boxes=6, box_size=(16, 120), device=dev)
    imgs, boxes, labels = ds.batch([0, 1, 2])
    # the scales the reference would draw: one per sample, in sample order, from the global numpy stream
    np.random.seed(5)
    r = Resize(img_scale=[(640, 200), (640, 320)], keep_ratio=True)
    scales = [r.draw_scale() for _ in range(3)]
    assert len(set(scales)) > 1
    np.random.seed(5)
    pipe = DevicePipeline([t for t in pipeline if t['type'] != 'OAMix'], dtype=torch.float32)
    data = pipe(imgs, boxes, labels)
    metas = data['img_metas']
    from oadg_amd.pipelines.geometric import rescale_size
    for m, sc in zip(metas, scales):
        wn, hn = rescale_size(512, 256, sc)
        assert tuple(m['img_shape']) == (hn, wn, 3)
        assert tuple(m['pad_shape']) == (-(-hn // 32) * 32, -(-wn // 32) * 32, 3)
    Hb, Wb = data['img'].shape[2:]
    assert (Hb, Wb) == (max(m['pad_shape'][0] for m in metas), max(m['pad_shape'][1] for m in metas))
    assert len({tuple(m['img_shape']) for m in metas}) > 1, 'the case must mix image sizes'
    for i, m in enumerate(metas):
        h, w = m['img_shape'][:2]
        assert float(data['img'][i, :, h:, :].abs().max() if h < Hb else 0) == 0.0
        assert float(data['img'][i, :, :, w:].abs().max() if w < Wb else 0) == 0.0
        assert float(data['img'][i, :, :h, :w].abs().max()) > 0
        assert float(data['gt_bboxes'][i][:, 2].max()) <= w and float(data['gt_bboxes'][i][:, 3].max()) <= h
    # the whole OA-DG step on such a batch
    set_random_seed(0)
    det = build_detector(cfg.model)
    det.init_weights(allow_missing_pretrained=True)
    det = det.to(dev).to(memory_format=torch.channels_last).train()
    np.random.seed(5)
    full = DevicePipeline(pipeline, dtype=torch.bfloat16)
    data = full(imgs, boxes, labels)
    assert data['img2'].shape == data['img'].shape
    eng = TrainEngine(det, build_optimizer(det, cfg.optimizer), amp_dtype=torch.bfloat16)
    try:
        out = eng.step(data)
        assert np.isfinite(float(out['loss'])) and float(out['loss']) > 0
        # proposals: clipped per image; RPN targets: nothing is sampled outside an image's own padded extent
        with torch.no_grad(), torch.autocast('cuda', dtype=torch.bfloat16):
            d2 = integrate_data(full(imgs, boxes, labels), det.train_cfg)
            x = det.extract_feat(d2['img'])
            props = det.rpn_head.get_bboxes(*det.rpn_head(x), img_metas=d2['img_metas'], cfg=det.train_cfg.rpn_proposal)
        for p, m in zip(props, d2['img_metas']):
            h, w = m['img_shape'][:2]
            assert len(p) > 0 and float(p[:, 2].max()) <= w and float(p[:, 3].max()) <= h and float(p[:, :4].min()) >= 0
        # (the targets stored on the head belong to the LAST loss call = eng.step(data); integrate_data extended
        #  data['img_metas'] in place to both views)
        lw_l = det.rpn_head.rpn_targets[1]
        stride0 = det.rpn_head.prior_generator.strides[0][0]
        Hs, Ws = data['img'].shape[2:]
        fh, fw = -(-Hs // stride0), -(-Ws // stride0)
        assert len(data['img_metas']) == 2 * len(metas)
        lw = lw_l[0].reshape(2 * len(metas), fh, fw, -1)
        assert float(lw.abs().max()) > 0
        mixed = 0
        for i, m in enumerate(data['img_metas']):
            vh, vw = -(-m['pad_shape'][0] // stride0), -(-m['pad_shape'][1] // stride0)
            if vh < fh:
                mixed += 1
                assert float(lw[i, vh:].abs().max()) == 0.0
            if vw < fw:
                mixed += 1
                assert float(lw[i, :, vw:].abs().max()) == 0.0
        assert mixed > 0
    finally:
        hip_conv.enable(False)
