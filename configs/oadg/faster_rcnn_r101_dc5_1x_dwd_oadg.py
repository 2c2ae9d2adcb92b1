# OA-DG on the Diverse-Weather dataset setting: Faster R-CNN R101-DC5 + OA-Mix ('augmix.all') + OA-Loss.
# BASELINE.json names this config; the reference ships only the plain baseline
# (configs/OA-DG/dwd/faster_rcnn_r101_dc5_1x_dwd.py) and the OA-Mix pipeline (configs/OA-DG/_base_/dwd_oamix.py),
# so it is composed here from those two and the ContrastiveRoIHead / OA-Loss block of the Cityscapes OA-DG config
# (configs/OA-DG/cityscapes/faster_rcnn_r50_fpn_1x_cityscapes_oadg.py:17-44), as SURVEY.md section 0 prescribes.
_base_ = ['../_base_/faster_rcnn_r50_caffe_dc5.py', '../_base_/runtime.py']
num_views = 2
_plus = dict(num_views=num_views)
model = dict(
    backbone=dict(depth=101, init_cfg=None),
    rpn_head=dict(
        loss_cls=dict(type='CrossEntropyLossPlus', use_sigmoid=True, loss_weight=1.0, additional_loss='jsdv1_3_2aug',
                      lambda_weight=0.1, wandb_name='rpn_cls', **_plus),
        loss_bbox=dict(type='L1LossPlus', loss_weight=1.0, additional_loss='None', lambda_weight=0.0,
                       wandb_name='rpn_bbox', **_plus)),
    roi_head=dict(
        type='ContrastiveRoIHead',
        bbox_head=dict(
            type='Shared2FCContrastiveHead', in_channels=2048, num_classes=7, with_cont=True, out_dim_cont=256,
            cont_predictor_cfg=dict(num_linear=2, feat_channels=256, return_relu=True),
            loss_cls=dict(type='CrossEntropyLossPlus', use_sigmoid=False, loss_weight=1.0,
                          additional_loss='jsdv1_3_2aug', lambda_weight=10, wandb_name='roi_cls', **_plus),
            loss_bbox=dict(type='SmoothL1LossPlus', beta=1.0, loss_weight=1.0, additional_loss='None',
                           lambda_weight=0.0, wandb_name='roi_bbox', **_plus),
            loss_cont=dict(type='ContrastiveLossPlus', loss_weight=0.01, temperature=0.06, **_plus))),
    train_cfg=dict(random_proposal_cfg=dict(bbox_from='oagrb', num_bboxes=10, scales=(0.01, 0.3),
                                            ratios=(0.3, 1 / 0.3), iou_max=0.7, iou_min=0.0)))
oamix_config = dict(type='OAMix', version='augmix.all', num_views=num_views, keep_orig=True,
                    use_mix=True, mixture_width=1, mixture_depth=-1, use_oa=True, oa_version='saliency_sparse',
                    use_mrange=False, use_multilevel=True)
img_norm_cfg = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
train_pipeline = [
    oamix_config,
    dict(type='Normalize', **img_norm_cfg),
    dict(type='Pad', size_divisor=32),
    dict(type='DefaultFormatBundle'),
    dict(type='Collect', keys=['img', 'img2', 'gt_bboxes', 'gt_bboxes2', 'gt_labels', 'multilevel_boxes',
                               'oamix_boxes']),
]
data = dict(samples_per_gpu=2, workers_per_gpu=8,
            train=dict(type='SyntheticCityscapes', img_shape=(720, 1280), num_boxes=12, num_classes=7, length=19395,
                       pipeline=train_pipeline))
optimizer = dict(type='SGD', lr=0.001, momentum=0.9, weight_decay=0.0001)
optimizer_config = dict(grad_clip=None)
lr_config = dict(policy='step', step=[4, 8])
runner = dict(type='EpochBasedRunner', max_epochs=10)
