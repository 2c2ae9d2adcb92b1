# The reference's complete train pipeline list (configs/OA-DG/cityscapes/faster_rcnn_r50_fpn_1x_cityscapes_oadg.py:62-74):
# multi-scale Resize [(2048, 800), (2048, 1024)] + RandomFlip(0.5) ahead of OA-Mix.  The bench config
# (faster_rcnn_r50_fpn_1x_cityscapes_oadg.py) feeds BASELINE.json's fixed 1024x2048 batches instead.
_base_ = ['./faster_rcnn_r50_fpn_1x_cityscapes_oadg.py']
img_norm_cfg = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
train_pipeline = [
    dict(type='LoadImageFromFile'),
    dict(type='LoadAnnotations', with_bbox=True),
    dict(type='Resize', img_scale=[(2048, 800), (2048, 1024)], keep_ratio=True),
    dict(type='RandomFlip', flip_ratio=0.5),
    dict(type='OAMix', version='augmix', num_views=2, keep_orig=True, severity=10,
         random_box_ratio=(3, 1 / 3), random_box_scale=(0.01, 0.1),
         oa_random_box_scale=(0.005, 0.1), oa_random_box_ratio=(3, 1 / 3), spatial_ratio=4, sigma_ratio=0.3),
    dict(type='Normalize', **img_norm_cfg),
    dict(type='Pad', size_divisor=32),
    dict(type='DefaultFormatBundle'),
    dict(type='Collect', keys=['img', 'img2', 'gt_bboxes', 'gt_bboxes2', 'gt_labels', 'multilevel_boxes',
                               'oamix_boxes']),
]
data = dict(train=dict(pipeline=train_pipeline))
