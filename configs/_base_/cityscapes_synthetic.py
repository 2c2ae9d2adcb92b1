# Cityscapes-shaped data: 1024x2048 images, 8 classes.  The dataset itself is out of scope:
# `SyntheticCityscapes` produces seeded low-pass-noise images and ~20 boxes per image on the device.
dataset_type = 'SyntheticCityscapes'
img_norm_cfg = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
train_pipeline = [
    dict(type='Normalize', **img_norm_cfg),
    dict(type='Pad', size_divisor=32),
    dict(type='DefaultFormatBundle'),
    dict(type='Collect', keys=['img', 'gt_bboxes', 'gt_labels']),
]
# the reference's test pipeline (configs/_base_/datasets/cityscapes_detection.py:19-33)
test_pipeline = [
    dict(type='LoadImageFromFile'),
    dict(type='MultiScaleFlipAug', img_scale=(2048, 1024), flip=False,
         transforms=[dict(type='Resize', keep_ratio=True), dict(type='RandomFlip'),
                     dict(type='Normalize', **img_norm_cfg), dict(type='Pad', size_divisor=32),
                     dict(type='ImageToTensor', keys=['img']), dict(type='Collect', keys=['img'])]),
]
data = dict(samples_per_gpu=1, workers_per_gpu=2,
            train=dict(type=dataset_type, img_shape=(1024, 2048), num_boxes=20, num_classes=8, length=2975,
                       pipeline=train_pipeline),
            val=dict(type=dataset_type, img_shape=(1024, 2048), num_boxes=20, num_classes=8, length=500,
                     pipeline=test_pipeline),
            test=dict(type=dataset_type, img_shape=(1024, 2048), num_boxes=20, num_classes=8, length=500,
                      pipeline=test_pipeline))
