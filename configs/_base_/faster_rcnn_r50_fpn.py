# Faster R-CNN R50-FPN, the model the OA-DG Cityscapes configs start from
# (same keys/values as the reference's configs/_base_/models/faster_rcnn_r50_fpn.py; 80 COCO classes here,
# overridden to 8 by the Cityscapes config).
_anchor = dict(type='AnchorGenerator', scales=[8], ratios=[0.5, 1.0, 2.0], strides=[4, 8, 16, 32, 64])
_assign = lambda pos, neg, mn, lq: dict(type='MaxIoUAssigner', pos_iou_thr=pos, neg_iou_thr=neg,  # noqa: E731
                                        min_pos_iou=mn, match_low_quality=lq, ignore_iof_thr=-1)
_sample = lambda n, frac, gt: dict(type='RandomSampler', num=n, pos_fraction=frac, neg_pos_ub=-1,  # noqa: E731
                                   add_gt_as_proposals=gt)
_coder = lambda stds: dict(type='DeltaXYWHBBoxCoder', target_means=[0., 0., 0., 0.], target_stds=stds)  # noqa: E731

model = dict(
    type='FasterRCNN',
    backbone=dict(type='ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                  norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, style='pytorch',
                  init_cfg=dict(type='Pretrained', checkpoint='torchvision://resnet50')),
    neck=dict(type='FPN', in_channels=[256, 512, 1024, 2048], out_channels=256, num_outs=5),
    rpn_head=dict(type='RPNHead', in_channels=256, feat_channels=256, anchor_generator=_anchor,
                  bbox_coder=_coder([1.0, 1.0, 1.0, 1.0]),
                  loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                  loss_bbox=dict(type='L1Loss', loss_weight=1.0)),
    roi_head=dict(
        type='StandardRoIHead',
        bbox_roi_extractor=dict(type='SingleRoIExtractor',
                                roi_layer=dict(type='RoIAlign', output_size=7, sampling_ratio=0),
                                out_channels=256, featmap_strides=[4, 8, 16, 32]),
        bbox_head=dict(type='Shared2FCBBoxHead', in_channels=256, fc_out_channels=1024, roi_feat_size=7,
                       num_classes=80, bbox_coder=_coder([0.1, 0.1, 0.2, 0.2]), reg_class_agnostic=False,
                       loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0),
                       loss_bbox=dict(type='L1Loss', loss_weight=1.0))),
    train_cfg=dict(
        rpn=dict(assigner=_assign(0.7, 0.3, 0.3, True), sampler=_sample(256, 0.5, False), allowed_border=-1,
                 pos_weight=-1, debug=False),
        rpn_proposal=dict(nms_pre=2000, max_per_img=1000, nms=dict(type='nms', iou_threshold=0.7),
                          min_bbox_size=0),
        rcnn=dict(assigner=_assign(0.5, 0.5, 0.5, False), sampler=_sample(512, 0.25, True), pos_weight=-1,
                  debug=False),
        wandb=dict(layer_list=[])),
    test_cfg=dict(
        rpn=dict(nms_pre=1000, max_per_img=1000, nms=dict(type='nms', iou_threshold=0.7), min_bbox_size=0),
        rcnn=dict(score_thr=0.05, nms=dict(type='nms', iou_threshold=0.5), max_per_img=100)))
