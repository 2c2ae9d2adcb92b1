"""Inference post-processing: ``multiclass_nms`` (mmdet/core/post_processing/bbox_nms.py:8-98),
``bbox2result`` (mmdet/core/bbox/transforms.py:118-139) and the test-time-augmentation merges
(mmdet/core/post_processing/merge_augs.py:13-112, mmdet/core/bbox/transforms.py:22-72).

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


# ------------------------------------------------------------------------------------ test-time augmentation merges
def bbox_flip(bboxes, img_shape, direction='horizontal'):
    """transforms.py:22-49: boxes [..., 4k] flipped inside an image of ``img_shape`` (h, w, ...)"""
    assert bboxes.shape[-1] % 4 == 0 and direction in ('horizontal', 'vertical', 'diagonal')
    flipped = bboxes.clone()
    if direction in ('horizontal', 'diagonal'):
        flipped[..., 0::4] = img_shape[1] - bboxes[..., 2::4]
        flipped[..., 2::4] = img_shape[1] - bboxes[..., 0::4]
    if direction in ('vertical', 'diagonal'):
        flipped[..., 1::4] = img_shape[0] - bboxes[..., 3::4]
        flipped[..., 3::4] = img_shape[0] - bboxes[..., 1::4]
    return flipped


def bbox_mapping(bboxes, img_shape, scale_factor, flip, flip_direction='horizontal'):
    """original image scale -> testing scale (transforms.py:51-60)"""
    new = bboxes * bboxes.new_tensor(scale_factor)
    return bbox_flip(new, img_shape, flip_direction) if flip else new


def bbox_mapping_back(bboxes, img_shape, scale_factor, flip, flip_direction='horizontal'):
    """testing scale -> original image scale (transforms.py:63-72)"""
    new = bbox_flip(bboxes, img_shape, flip_direction) if flip else bboxes
    new = new.view(-1, 4) / new.new_tensor(scale_factor)
    return new.view(bboxes.shape)


def nms_plain(dets, iou_thr):
    """mmcv.ops.nms on dets [n, 5]: the kept rows in descending-score order (greedy, IoU > thr suppressed)"""
    if dets.shape[0] == 0:
        return dets
    order = dets[:, 4].sort(descending=True, stable=True)[1]
    boxes = dets[order, :4].float().contiguous()
    keep, cnt = hip_ops.nms_sorted_batched(boxes[None], torch.tensor([boxes.shape[0]], dtype=torch.int32, device=dets.device),
                                           iou_thr, -1)
    k = int(cnt.item())
    return dets[order[keep[0, :k].long()]]


def merge_aug_proposals(aug_proposals, img_metas, cfg):
    """merge_augs.py:13-81: the proposals of every augmentation mapped back to the original image, one NMS over all of
    them, the best ``max_per_img``"""
    recovered = []
    for proposals, info in zip(aug_proposals, img_metas):
        p = proposals.clone()
        p[:, :4] = bbox_mapping_back(p[:, :4], info['img_shape'], info['scale_factor'], info['flip'],
                                     info.get('flip_direction') or 'horizontal')
        recovered.append(p)
    allp = torch.cat(recovered, dim=0)
    nms_cfg = dict(cfg.get('nms', {}))
    thr = nms_cfg.get('iou_threshold', nms_cfg.get('iou_thr', cfg.get('nms_thr')))
    merged = nms_plain(allp, thr)
    order = merged[:, 4].sort(0, descending=True)[1]
    num = min(cfg.get('max_per_img', cfg.get('max_num')), merged.shape[0])
    return merged[order[:num], :]


def merge_aug_bboxes(aug_bboxes, aug_scores, img_metas, rcnn_test_cfg):
    """merge_augs.py:84-112: boxes [n, 4 * classes] of every augmentation mapped back and averaged, scores averaged"""
    recovered = []
    for bboxes, info in zip(aug_bboxes, img_metas):
        i0 = info[0]
        recovered.append(bbox_mapping_back(bboxes, i0['img_shape'], i0['scale_factor'], i0['flip'],
                                           i0.get('flip_direction') or 'horizontal'))
    bboxes = torch.stack(recovered).mean(dim=0)
    if aug_scores is None:
        return bboxes
    return bboxes, torch.stack(aug_scores).mean(dim=0)
