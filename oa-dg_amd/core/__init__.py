"""Host-side box/anchor logic of the hot path: plain torch tensor
plumbing on whatever device the tensors live on.  Mirrors mmdet/core/{anchor,bbox,utils}."""
from .anchor import AnchorGenerator, anchor_inside_flags, images_to_levels  # noqa: F401
from .bbox import (AssignResult, DeltaXYWHBBoxCoder, MaxIoUAssigner, RandomSampler, SamplingResult,  # noqa: F401
                   bbox2delta, bbox2roi, bbox_overlaps, bbox_overlaps_np, delta2bbox)
from .misc import multi_apply, select_single_mlvl, unmap  # noqa: F401
from .post_processing import (bbox2result, bbox_flip, bbox_mapping, bbox_mapping_back, merge_aug_bboxes,  # noqa: F401
                              merge_aug_proposals, multiclass_nms, nms_plain)
