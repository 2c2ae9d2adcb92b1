"""Inference post-processing: ``multiclass_nms`` (mmdet/core/post_processing/bbox_nms.py:8-98) and
``bbox2result`` (mmdet/core/bbox/transforms.py:118-139).

``mmcv.ops.batched_nms`` (class-offset trick + greedy NMS in descending-score order) runs through the same device
kernel as the RPN proposals (csrc/nms.hip via ``hip_ops.nms_sorted_batched``): boxes under the score threshold are
sorted behind the valid ones instead of being compacted with ``nonzero``, so the only host read is the final count.
"""
import numpy as np
import torch

from .. import hip_ops


def multiclass_nms(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1, score_factors=None, return_inds=False):
    """multi_bboxes [n, C*4] or [n, 4]; multi_scores [n, C+1] (last column = background, ignored).
    Returns (dets [k, 5], labels [k]) (+ flat indices into the n*C candidates), in descending-score order."""
    num_classes = multi_scores.size(1) - 1
    n = multi_scores.size(0)
    if multi_bboxes.shape[1] > 4:
        bboxes = multi_bboxes.view(n, -1, 4)
    else:
        bboxes = multi_bboxes[:, None].expand(n, num_classes, 4)
    scores = multi_scores[:, :-1]
    labels = torch.arange(num_classes, dtype=torch.long, device=scores.device).view(1, -1).expand_as(scores)
    bboxes, scores, labels = bboxes.reshape(-1, 4).float(), scores.reshape(-1).float(), labels.reshape(-1)
    valid = scores > score_thr
    if score_factors is not None:
        scores = scores * score_factors.view(-1, 1).expand(n, num_classes).reshape(-1)
    M = scores.numel()
    if M == 0:
        dets = torch.cat([bboxes, scores[:, None]], -1)
        inds = labels.new_zeros((0,))
        return (dets, labels, inds) if return_inds else (dets, labels)
    cfg = dict(nms_cfg)
    assert cfg.pop('type', 'nms') == 'nms' and not cfg.get('class_agnostic', False)
    thr = cfg.get('iou_threshold', cfg.get('iou_thr'))
    # batched_nms: every class shifted by (largest coordinate of the surviving boxes + 1)
    mx = torch.where(valid[:, None], bboxes, bboxes.new_full((), -float('inf'))).amax()
    offs = labels.to(bboxes) * (mx + 1)
    key = torch.where(valid, scores, scores.new_full((), -float('inf')))
    order = key.sort(descending=True, stable=True)[1]
    boxes_sorted = (bboxes + offs[:, None])[order]
    counts = valid.sum().int().view(1)
    keep, keep_cnt = hip_ops.nms_sorted_batched(boxes_sorted[None], counts, thr, max_num)
    k = int(keep_cnt.item())                      # the one host read
    if max_num > 0:
        k = min(k, max_num)
    sel = order[keep[0, :k].long()]
    dets = torch.cat([bboxes[sel], scores[sel, None]], -1)
    return (dets, labels[sel], sel) if return_inds else (dets, labels[sel])


def bbox2result(bboxes, labels, num_classes):
    """det boxes [k, 5] + labels [k] -> list (per class) of float32 numpy arrays [k_c, 5]."""
    if bboxes.shape[0] == 0:
        return [np.zeros((0, 5), dtype=np.float32) for _ in range(num_classes)]
    if isinstance(bboxes, torch.Tensor):
        bboxes = bboxes.detach().cpu().numpy()
        labels = labels.detach().cpu().numpy()
    return [bboxes[labels == i, :] for i in range(num_classes)]
