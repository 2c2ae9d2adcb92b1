"""CPU restatement of mmcv.ops.nms / batched_nms.  Test infrastructure only (see oracle/__init__).

PARITY UNPINNED: mmcv-full (1.3.17..1.5.0, mmdet/__init__.py:19-27) is neither vendored nor installed.
Restated from the published algorithm (mmcv/ops/nms.py batched_nms + csrc nms kernels),
anchored on the call site mmdet/models/dense_heads/rpn_head.py:231 and the slice at :235.  The reference's
tests/test_utils/test_nms.py only checks input validation.
"""
import numpy as np
import torch

f32 = np.float32


def iou_matrix_f32(b):
    x1, y1, x2, y2 = (b[:, i].astype(np.float32) for i in range(4))
    area = (x2 - x1) * (y2 - y1)
    left = np.maximum(x1[:, None], x1[None]); right = np.minimum(x2[:, None], x2[None])
    top = np.maximum(y1[:, None], y1[None]); bottom = np.minimum(y2[:, None], y2[None])
    w = np.maximum(right - left, f32(0)); h = np.maximum(bottom - top, f32(0))
    inter = (w * h).astype(np.float32)
    with np.errstate(divide='ignore', invalid='ignore'):
        return inter / ((area[:, None] + area[None]).astype(np.float32) - inter)


def nms_sorted(boxes, iou_thr, max_keep=-1):
    """Greedy suppression over boxes already in descending-score order; IoU > thr suppresses (offset 0)."""
    M = boxes.shape[0]
    iou = iou_matrix_f32(np.asarray(boxes, np.float32)) if M else np.zeros((0, 0), np.float32)
    removed = np.zeros(M, bool)
    keep = []
    for i in range(M):
        if removed[i]:
            continue
        keep.append(i)
        if 0 < max_keep <= len(keep):
            break
        removed[i + 1:] |= iou[i, i + 1:] > f32(iou_thr)
    return np.asarray(keep, np.int64)


def nms(boxes, scores, iou_threshold):
    """mmcv.ops.nms: returns (dets [M',5], keep) with keep in descending-score order (stable sort)."""
    order = torch.sort(scores, descending=True, stable=True)[1]
    k = nms_sorted(boxes[order].detach().cpu().numpy(), iou_threshold)
    keep = order[torch.as_tensor(k, dtype=torch.long)]
    dets = torch.cat([boxes[keep], scores[keep].view(-1, 1)], dim=1)
    return dets, keep


def batched_nms(boxes, scores, idxs, nms_cfg, class_agnostic=False):
    """mmcv.ops.batched_nms: per-class offset trick; a per-id loop + re-sort when M >= split_thr."""
    cfg = dict(nms_cfg)
    class_agnostic = cfg.pop('class_agnostic', class_agnostic)
    thr = cfg.get('iou_threshold', cfg.get('iou_thr'))
    split_thr = cfg.get('split_thr', 10000)
    if boxes.numel() == 0:
        return torch.cat([boxes, scores.view(-1, 1)], 1), boxes.new_zeros((0,), dtype=torch.long)
    if class_agnostic:
        b = boxes
    else:
        mx = boxes.max()
        b = boxes + (idxs.to(boxes) * (mx + boxes.new_tensor(1)))[:, None]
    if b.shape[0] < split_thr:
        dets, keep = nms(b, scores, thr)
        return torch.cat([boxes[keep], dets[:, -1:]], 1), keep
    mask = scores.new_zeros(scores.size(), dtype=torch.bool)
    for i in torch.unique(idxs):
        m = (idxs == i).nonzero(as_tuple=False).view(-1)
        _, k = nms(b[m], scores[m], thr)
        mask[m[k]] = True
    keep = mask.nonzero(as_tuple=False).view(-1)
    # mmcv: keep[scores[keep].argsort(descending=True)] - an UNSTABLE argsort, so the order among exactly equal scores
    # (saturated sigmoids are common) is whatever the torch build's sort does.  The restatement fixes it: ties in
    # ascending input index, i.e. the same order the single-call branch above produces.
    keep = keep[torch.sort(scores[keep], descending=True, stable=True)[1]]
    return torch.cat([boxes[keep], scores[keep].view(-1, 1)], 1), keep
