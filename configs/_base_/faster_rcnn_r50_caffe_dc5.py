# Faster R-CNN on a dilated C5 ResNet (caffe style, no FPN): the base of the DWD (S-DGOD) experiments
# (values follow the reference's configs/_base_/models/faster_rcnn_r50_caffe_dc5.py).
_norm = dict(type='BN', requires_grad=False)
_coder = lambda stds: dict(type='DeltaXYWHBBoxCoder', target_means=[0., 0., 0., 0.], target_stds=stds)  # noqa: E731
model = dict(
    type='FasterRCNN',
    backbone=dict(type='ResNet', depth=50, num_stages=4, strides=(1, 2, 2, 1), dilations=(1, 1, 1, 2),
                  out_indices=(3,), frozen_stages=1, norm_cfg=_norm, norm_eval=True, style='caffe',
                  init_cfg=dict(type='Pretrained', checkpoint='open-mmlab://detectron2/resnet50_caffe')),
    rpn_head=dict(type='RPNHead', in_channels=2048, feat_channels=2048,
                  anchor_generator=dict(type='AnchorGenerator', scales=[2, 4, 8, 16, 32], ratios=[0.5, 1.0, 2.0],
                                        strides=[16]),
                  bbox_coder=_coder([1.0, 1.0, 1.0, 1.0]),
                  loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                  loss_bbox=dict(type='L1Loss', loss_weight=1.0)),
    roi_head=dict(
        type='StandardRoIHead',
        bbox_roi_extractor=dict(type='SingleRoIExtractor',
                                roi_layer=dict(type='RoIAlign', output_size=7, sampling_ratio=0),
                                out_channels=2048, featmap_strides=[16]),
        bbox_head=dict(type='Shared2FCBBoxHead', in_channels=2048, fc_out_channels=1024, roi_feat_size=7,
                       num_classes=80, bbox_coder=_coder([0.1, 0.1, 0.2, 0.2]), reg_class_agnostic=False,
                       loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0),
                       loss_bbox=dict(type='L1Loss', loss_weight=1.0))),
    train_cfg=dict(
        rpn=dict(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3,
                               match_low_quality=True, ignore_iof_thr=-1),
                 sampler=dict(type='RandomSampler', num=256, pos_fraction=0.5, neg_pos_ub=-1,
                              add_gt_as_proposals=False),
                 allowed_border=0, pos_weight=-1, debug=False),
        rpn_proposal=dict(nms_pre=12000, max_per_img=2000, nms=dict(type='nms', iou_threshold=0.7), min_bbox_size=0),
        rcnn=dict(assigner=dict(type='MaxIoUAssigner', pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5,
                                match_low_quality=False, ignore_iof_thr=-1),
                  sampler=dict(type='RandomSampler', num=512, pos_fraction=0.25, neg_pos_ub=-1,
                               add_gt_as_proposals=True),
                  pos_weight=-1, debug=False)),
    test_cfg=dict(rpn=dict(nms=dict(type='nms', iou_threshold=0.7), nms_pre=6000, max_per_img=1000, min_bbox_size=0),
                  rcnn=dict(score_thr=0.05, nms=dict(type='nms', iou_threshold=0.5), max_per_img=100)))
