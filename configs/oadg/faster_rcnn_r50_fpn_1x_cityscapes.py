# Baseline Faster R-CNN R50-FPN on (synthetic) Cityscapes: what the OA-DG config below extends.
_base_ = ['../_base_/faster_rcnn_r50_fpn.py', '../_base_/cityscapes_synthetic.py', '../_base_/runtime.py']
model = dict(backbone=dict(init_cfg=None),
             roi_head=dict(bbox_head=dict(num_classes=8,
                                          loss_bbox=dict(type='SmoothL1Loss', beta=1.0, loss_weight=1.0))),
             train_cfg=dict(rcnn=dict(dropout=False), wandb=dict(log=dict(features_list=[], vars=['log_vars']))))
data = dict(samples_per_gpu=2, workers_per_gpu=4)
optimizer = dict(type='SGD', lr=0.01, momentum=0.9, weight_decay=0.0001)   # lr for a total batch of 8
optimizer_config = dict(grad_clip=None)
lr_config = dict(policy='step', warmup='linear', warmup_iters=500, warmup_ratio=0.001, step=[1])
runner = dict(type='EpochBasedRunner', max_epochs=2)
log_config = dict(interval=100, hooks=[dict(type='TextLoggerHook')])
