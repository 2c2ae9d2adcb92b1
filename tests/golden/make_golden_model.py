"""Generate tests/golden/model_*.npz: the REFERENCE detector (every line executed is reference code except
mmcv.ops.RoIAlign / batched_nms, which are oracle/roi_align.py and oracle/nms.py) run for one train_step on
seeded inputs with name-seeded weights.  Run here only:  python tests/golden/make_golden_model.py
"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import refload  # noqa: E402
from inputs import DC5_SCALE, model_batch, named_weights  # noqa: E402
from oracle import nms as ONMS  # noqa: E402
from oracle import roi_align as ORA  # noqa: E402


class OracleRoIAlign(nn.Module):
    def __init__(self, output_size, spatial_scale=1.0, sampling_ratio=0, pool_mode='avg', aligned=True,
                 use_torchvision=False):
        super().__init__()
        self.output_size = (output_size, output_size) if isinstance(output_size, int) else tuple(output_size)
        self.spatial_scale, self.sampling_ratio, self.aligned = spatial_scale, sampling_ratio, aligned
        assert pool_mode == 'avg'

    def forward(self, x, rois):
        return ORA.roi_align(x, rois, self.output_size, self.spatial_scale, self.sampling_ratio, self.aligned)


def build_reference_detector(config='r50fpn'):
    """config 'r50fpn': the reference's own configs/OA-DG/cityscapes/faster_rcnn_r50_fpn_1x_cityscapes_oadg.py;
    'dc5': BASELINE configs[3], which the reference does not ship - the model dict comes from this repo's composed
    configs/oadg/faster_rcnn_r101_dc5_1x_dwd_oadg.py (reference type strings only) and is built by the REFERENCE's
    registries, so every module executed is still reference code."""
    refload.install(ops=dict(RoIAlign=OracleRoIAlign, batched_nms=ONMS.batched_nms, nms=ONMS.nms))
    for m in ['mmdet.core.bbox.assigners.max_iou_assigner', 'mmdet.core.bbox.samplers.random_sampler',
              'mmdet.core.bbox.coder.delta_xywh_bbox_coder', 'mmdet.core.bbox.iou_calculators.iou2d_calculator',
              'mmdet.core.anchor.anchor_generator', 'mmdet.models.losses.oadg.cross_entropy_loss_plus',
              'mmdet.models.losses.oadg.smooth_l1_loss_plus', 'mmdet.models.losses.oadg.contrastive_loss_plus',
              'mmdet.models.backbones.resnet', 'mmdet.models.necks.fpn', 'mmdet.models.dense_heads.rpn_head',
              'mmdet.models.roi_heads.roi_extractors.single_level_roi_extractor',
              'mmdet.models.roi_heads.bbox_heads.contrastive_head',
              'mmdet.models.roi_heads.contrastive_roi_head', 'mmdet.models.detectors.faster_rcnn']:
        refload.ref(m)
    import oadg_amd
    if config == 'dc5':
        cfg = oadg_amd.Config.fromfile(os.path.join(ROOT, 'configs/oadg/faster_rcnn_r101_dc5_1x_dwd_oadg.py'),
                                       import_custom_modules=False)
    else:
        os.environ['OADG_CONFIG_ROOT'] = refload.REF
        cfg = oadg_amd.Config.fromfile(
            os.path.join(refload.REF, 'configs/OA-DG/cityscapes/faster_rcnn_r50_fpn_1x_cityscapes_oadg.py'),
            import_custom_modules=False)
    model_cfg = refload.to_cfg(cfg.to_dict()['model'])
    model_cfg['backbone']['init_cfg'] = None
    det = refload.ref('mmdet.models.builder', 'build_detector')(model_cfg)
    return det


def load_named(model, scale=None):
    sd = model.state_dict()
    w = named_weights({k: v.shape for k, v in sd.items()}, scale)
    model.load_state_dict({k: torch.as_tensor(v) for k, v in w.items()})


def to_batch(b):
    return dict(img=torch.tensor(b['img']), img2=torch.tensor(b['img2']),
                gt_bboxes=[torch.tensor(x) for x in b['gt_bboxes']],
                gt_bboxes2=[torch.tensor(x) for x in b['gt_bboxes']],
                gt_labels=[torch.tensor(x) for x in b['gt_labels']],
                multilevel_boxes=[torch.tensor(x) for x in b['multilevel_boxes']],
                oamix_boxes=[torch.tensor(x) for x in b['oamix_boxes']],
                img_metas=[dict(img_shape=b['img'].shape[2:] + (3,), pad_shape=b['img'].shape[2:] + (3,),
                                ori_shape=b['img'].shape[2:] + (3,), scale_factor=1.0, flip=False)
                           for _ in range(b['img'].shape[0])])


def grad_report(model, samples=True):
    rep = {}
    groups = {}
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        top = '.'.join(n.split('.')[:2])
        groups.setdefault(top, 0.0)
        groups[top] += float(p.grad.double().pow(2).sum())
    for k, v in groups.items():
        rep['gn_' + k] = np.float64(np.sqrt(v))
    params = dict(model.named_parameters())
    for n in ['roi_head.bbox_head.fc_cls.weight', 'roi_head.bbox_head.fc_cont.2.weight',
              'rpn_head.rpn_cls.weight', 'neck.fpn_convs.0.conv.bias', 'backbone.layer4.2.conv3.weight']:
        if samples and n in params:
            rep['g_' + n] = params[n].grad.detach().flatten()[:4096].numpy().copy()
    return rep


def main(h=256, w=512, n_img=2, seed=0, config='r50fpn', name=None, n_gt=12, n_cls=8, samples=True):
    det = build_reference_detector(config)
    load_named(det, DC5_SCALE if config == 'dc5' else None)
    det.train()
    integrate = refload.ref('mmdet.models.detectors.base', 'integrate_data')
    batch = model_batch(seed, n_img, h, w, n_gt=n_gt, n_cls=n_cls)
    data = to_batch(batch)
    torch.manual_seed(seed)
    np.random.seed(seed)
    t0 = time.time()
    data = integrate(data, det.train_cfg)
    losses = det(**data)
    loss, log_vars = det._parse_losses(losses)
    loss.backward()
    print('reference step', time.time() - t0, 's', log_vars)
    out = dict(h=np.int64(h), w=np.int64(w), n_img=np.int64(n_img), seed=np.int64(seed), n_gt=np.int64(n_gt),
               n_cls=np.int64(n_cls), ref_step_seconds=np.float64(time.time() - t0))
    for k, v in log_vars.items():
        out['lv_' + k] = np.float64(v)
    out.update(grad_report(det, samples))
    # intermediate pins: sampled RoIs and labels (indices exact), random proposals
    tg = det.roi_head.bbox_targets
    out['roi_labels'] = tg[0].numpy().copy()
    out['roi_bbox_targets_sum'] = np.float64(tg[2].double().sum())
    out['n_params'] = np.int64(sum(p.numel() for p in det.parameters()))
    out['n_trainable'] = np.int64(sum(p.numel() for p in det.parameters() if p.requires_grad))
    name = name or f'model_step_{h}x{w}.npz'
    np.savez_compressed(os.path.join(HERE, name), **out)
    print('wrote', name, {k: v for k, v in out.items() if k.startswith('lv_') or k.startswith('n_')})


def main_test(h=256, w=512, n_img=2, seed=0):
    """Inference path: reference ``TwoStageDetector.simple_test`` (two_stage.py:224-266) -> ``simple_test_rpn`` ->
    ``StandardRoIHead.simple_test`` -> ``BBoxHead.get_bboxes`` -> ``multiclass_nms`` -> ``bbox2result`` on the same
    seeded batch and name-seeded weights, eval mode, rescale=True with a non-trivial scale factor."""
    det = build_reference_detector()
    load_named(det)
    det.eval()
    batch = model_batch(seed, n_img, h, w)
    img = torch.tensor(batch['img'])
    sf = np.array([1.25, 1.25, 1.25, 1.25], dtype=np.float32)
    metas = [dict(img_shape=(h, w, 3), pad_shape=(h, w, 3), ori_shape=(int(h / 1.25), int(w / 1.25), 3), scale_factor=sf,
                  flip=False, ori_filename=f'{i}.png') for i in range(n_img)]
    with torch.no_grad():
        x = det.extract_feat(img)
        props = det.rpn_head.simple_test_rpn(x, metas)
        results = det.simple_test(img, metas, rescale=True)
    out = dict(h=np.int64(h), w=np.int64(w), n_img=np.int64(n_img), seed=np.int64(seed), scale_factor=sf)
    for i in range(n_img):
        out[f'proposals{i}'] = props[i].numpy().copy()
        for c, arr in enumerate(results[i]):
            out[f'det{i}_c{c}'] = np.asarray(arr, dtype=np.float32)
    np.savez_compressed(os.path.join(HERE, f'model_test_{h}x{w}.npz'), **out)
    print('wrote', f'model_test_{h}x{w}.npz', [len(p) for p in props],
          [[len(a) for a in r] for r in results])


def aug_inputs(seed=0, h=256, w=512):
    """the four test-time augmentations of ONE seeded image, as MultiScaleFlipAug(img_scale=[(w, h), (1.25 w, 1.25 h)],
    flip=True) orders them: (scale 1, no flip), (scale 1, horizontal flip), (scale 1.25, no flip), (scale 1.25, flip).
    The larger view is torch's bilinear interpolation of the first (deterministic on the CPU) - aug_test's arithmetic does
    not care how the views were made, the fixture pins what it does with them."""
    import torch.nn.functional as F
    img = torch.tensor(model_batch(seed, 1, h, w)['img'])
    big = F.interpolate(img, size=(int(h * 1.25), int(w * 1.25)), mode='bilinear', align_corners=False)
    imgs, metas = [], []
    for t, sf in ((img, 1.0), (big, 1.25)):
        for flip in (False, True):
            imgs.append(t.flip(3) if flip else t)
            metas.append([dict(img_shape=tuple(t.shape[2:]) + (3,), pad_shape=tuple(t.shape[2:]) + (3,),
                               ori_shape=(h, w, 3), scale_factor=np.full(4, sf, dtype=np.float32), flip=flip,
                               flip_direction='horizontal' if flip else None, ori_filename='0.png')])
    return imgs, metas


def main_aug_test(h=256, w=512, seed=0):
    """Test-time augmentation: reference ``TwoStageDetector.aug_test`` (two_stage.py:268-277) -> ``aug_test_rpn`` +
    ``merge_aug_proposals`` (dense_test_mixins.py:135-167, merge_augs.py:13-83) -> ``aug_test_bboxes`` +
    ``merge_aug_bboxes`` (test_mixins.py:139-177, merge_augs.py:86-112) -> ``multiclass_nms`` -> ``bbox2result`` on four
    augmentations of one seeded image, name-seeded weights, eval mode, rescale=True."""
    det = build_reference_detector()
    load_named(det)
    det.eval()
    imgs, metas = aug_inputs(seed, h, w)
    with torch.no_grad():
        feats = det.extract_feats(imgs)
        props = det.rpn_head.aug_test_rpn(feats, metas)
        results = det.aug_test(imgs, metas, rescale=True)
    out = dict(h=np.int64(h), w=np.int64(w), seed=np.int64(seed), merged_proposals=props[0].numpy().copy())
    for c, arr in enumerate(results[0]):
        out[f'det_c{c}'] = np.asarray(arr, dtype=np.float32)
    np.savez_compressed(os.path.join(HERE, f'model_aug_test_{h}x{w}.npz'), **out)
    print('wrote', f'model_aug_test_{h}x{w}.npz', props[0].shape, [len(a) for a in results[0]])


if __name__ == '__main__':
    mode = sys.argv[1] if len(sys.argv) > 1 else 'step'
    if mode == 'test':
        main_test()
    elif mode == 'aug':
        main_aug_test()
    elif mode == 'full':       # SURVEY 8c G7 at BASELINE config 1's real shape (N=2, 1024x2048): ~2 min, 13.5 GB here
        main(1024, 2048, samples=False)
    elif mode == 'dc5':        # BASELINE configs[3] (R101-DC5 OA-DG): 17,280 anchors > nms_pre 12,000 > split_thr 10,000
        main(384, 768, config='dc5', name='model_step_dc5_384x768.npz', n_cls=7)
    else:
        main()
