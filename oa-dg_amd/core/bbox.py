"""IoU, assignment, sampling, box coding (mmdet/core/bbox/**, mmdet/core/evaluation/bbox_overlaps.py).

Sampling consumes the global CPU torch generator exactly like the reference (random_sampler.py:58:
``torch.randperm(n)`` with n = number of candidates), so seeded runs pick identical indices.
"""
import ctypes
import os

import numpy as np
import torch

from ..registry import BBOX_ASSIGNERS, BBOX_CODERS, BBOX_SAMPLERS, IOU_CALCULATORS


# ----------------------------------------------------------------------------------------------- IoU
def bbox_overlaps(b1, b2, mode='iou', is_aligned=False, eps=1e-6):
    """torch IoU/IoF/GIoU of [.., m, 4] x [.., n, 4] (iou_calculators/iou2d_calculator.py:78-...)."""
    assert mode in ('iou', 'iof', 'giou')
    rows, cols = b1.size(-2), b2.size(-2)
    batch = b1.shape[:-2]
    if is_aligned:
        assert rows == cols
    if rows * cols == 0:
        return b1.new(batch + ((rows,) if is_aligned else (rows, cols)))
    a1 = (b1[..., 2] - b1[..., 0]) * (b1[..., 3] - b1[..., 1])
    a2 = (b2[..., 2] - b2[..., 0]) * (b2[..., 3] - b2[..., 1])
    if is_aligned:
        lt = torch.max(b1[..., :2], b2[..., :2])
        rb = torch.min(b1[..., 2:], b2[..., 2:])
        wh = (rb - lt).clamp(min=0)
        overlap = wh[..., 0] * wh[..., 1]
        union = a1 + a2 - overlap if mode in ('iou', 'giou') else a1
        if mode == 'giou':
            elt = torch.min(b1[..., :2], b2[..., :2])
            erb = torch.max(b1[..., 2:], b2[..., 2:])
    else:
        lt = torch.max(b1[..., :, None, :2], b2[..., None, :, :2])
        rb = torch.min(b1[..., :, None, 2:], b2[..., None, :, 2:])
        wh = (rb - lt).clamp(min=0)
        overlap = wh[..., 0] * wh[..., 1]
        union = a1[..., None] + a2[..., None, :] - overlap if mode in ('iou', 'giou') else a1[..., None]
        if mode == 'giou':
            elt = torch.min(b1[..., :, None, :2], b2[..., None, :, :2])
            erb = torch.max(b1[..., :, None, 2:], b2[..., None, :, 2:])
    union = union.clamp(min=eps)
    ious = overlap / union
    if mode in ('iou', 'iof'):
        return ious
    ewh = (erb - elt).clamp(min=0)
    earea = (ewh[..., 0] * ewh[..., 1]).clamp(min=eps)
    return ious - (earea - union) / earea


@IOU_CALCULATORS.register_module()
class BboxOverlaps2D:

    def __init__(self, scale=1., dtype=None):
        self.scale, self.dtype = scale, dtype

    def __call__(self, b1, b2, mode='iou', is_aligned=False):
        if b2.size(-1) == 5:
            b2 = b2[..., :4]
        if b1.size(-1) == 5:
            b1 = b1[..., :4]
        return bbox_overlaps(b1, b2, mode, is_aligned)


def bbox_overlaps_np(bboxes1, bboxes2, mode='iou', eps=1e-6):
    """numpy fp32 IoU used by OA-Mix and the random-proposal generator
    (mmdet/core/evaluation/bbox_overlaps.py:5-65, use_legacy_coordinate=False)."""
    assert mode in ('iou', 'iof')
    b1 = np.asarray(bboxes1).astype(np.float32)
    b2 = np.asarray(bboxes2).astype(np.float32)
    rows, cols = b1.shape[0], b2.shape[0]
    if rows * cols == 0:
        return np.zeros((rows, cols), dtype=np.float32)
    swap = rows > cols
    if swap:
        b1, b2 = b2, b1
    area1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    area2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    out = np.zeros((b1.shape[0], b2.shape[0]), dtype=np.float32)
    for i in range(b1.shape[0]):
        w = np.maximum(np.minimum(b1[i, 2], b2[:, 2]) - np.maximum(b1[i, 0], b2[:, 0]), 0)
        h = np.maximum(np.minimum(b1[i, 3], b2[:, 3]) - np.maximum(b1[i, 1], b2[:, 1]), 0)
        inter = w * h
        if mode == 'iou':
            union = area1[i] + area2 - inter
        else:
            union = area1[i] if not swap else area2
        out[i, :] = inter / np.maximum(union, eps)
    return out.T if swap else out


# ----------------------------------------------------------------------------------------------- assign
class AssignResult:
    """assign_result.py: gt_inds 0 = negative, -1 = ignore, k>0 = gt k-1."""

    def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
        self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels

    @property
    def num_preds(self):
        return len(self.gt_inds)

    def add_gt_(self, gt_labels):
        """Prepend the gts as self-matched proposals (assign_result.py add_gt_)."""
        self_inds = torch.arange(1, len(gt_labels) + 1, dtype=torch.long, device=gt_labels.device)
        self.gt_inds = torch.cat([self_inds, self.gt_inds])
        self.max_overlaps = torch.cat([self.max_overlaps.new_ones(len(gt_labels)), self.max_overlaps])
        if self.labels is not None:
            self.labels = torch.cat([gt_labels, self.labels])


@BBOX_ASSIGNERS.register_module()
class MaxIoUAssigner:
    """max_iou_assigner.py:61-213.  The per-gt Python loop of low-quality matching (:195-201) is replaced by
    an equivalent "last matching gt wins" reduction; no host synchronisation."""

    def __init__(self, pos_iou_thr, neg_iou_thr, min_pos_iou=.0, gt_max_assign_all=True, ignore_iof_thr=-1,
                 ignore_wrt_candidates=True, match_low_quality=True, gpu_assign_thr=-1,
                 iou_calculator=dict(type='BboxOverlaps2D')):
        self.pos_iou_thr, self.neg_iou_thr, self.min_pos_iou = pos_iou_thr, neg_iou_thr, min_pos_iou
        self.gt_max_assign_all = gt_max_assign_all
        self.ignore_iof_thr, self.ignore_wrt_candidates = ignore_iof_thr, ignore_wrt_candidates
        self.match_low_quality = match_low_quality
        self.gpu_assign_thr = gpu_assign_thr
        self.iou_calculator = IOU_CALCULATORS.build(iou_calculator)

    def assign(self, bboxes, gt_bboxes, gt_bboxes_ignore=None, gt_labels=None):
        overlaps = self.iou_calculator(gt_bboxes, bboxes)
        if (self.ignore_iof_thr > 0 and gt_bboxes_ignore is not None and gt_bboxes_ignore.numel() > 0
                and bboxes.numel() > 0):
            if self.ignore_wrt_candidates:
                ig = self.iou_calculator(bboxes, gt_bboxes_ignore, mode='iof').max(dim=1)[0]
            else:
                ig = self.iou_calculator(gt_bboxes_ignore, bboxes, mode='iof').max(dim=0)[0]
            overlaps = overlaps.masked_fill((ig > self.ignore_iof_thr)[None, :], -1)
        return self.assign_wrt_overlaps(overlaps, gt_labels)

    def assign_masked(self, bboxes, valid, gt_bboxes, gt_labels=None):
        """assign() on the rows of ``bboxes`` flagged ``valid`` ([n] bool), without compacting them: invalid
        rows get gt_ind -1 and take no part in the per-gt maxima, exactly as if they had been filtered out."""
        overlaps = self.iou_calculator(gt_bboxes, bboxes)
        if overlaps.numel():
            overlaps = overlaps.masked_fill(~valid[None, :], -1)
        res = self.assign_wrt_overlaps(overlaps, gt_labels)
        res.gt_inds = res.gt_inds.masked_fill(~valid, -1)
        return res

    def assign_many(self, boxes, valids, gt_bboxes_list, gt_labels_list=None):
        """assign_masked() for a whole batch in one fused pass (csrc/assign.hip), bit-identical to the tensor
        path.  ``boxes``: one [N,4] tensor shared by all images, or a list of [N,4(+)] tensors of equal N;
        ``valids``: None, or a list of [N] bool masks (None entries = all valid).  Returns
        (list of AssignResult, counts [B,2] int32 on the device = #(gt_inds > 0), #(gt_inds == 0) per image),
        or None when the configuration / inputs are outside the kernel's domain (callers then loop)."""
        from .. import _lib
        B = len(gt_bboxes_list)
        shared = isinstance(boxes, torch.Tensor)
        first = boxes if shared else boxes[0]
        if not (first.is_cuda and first.dtype == torch.float32 and self.gt_max_assign_all and B > 0
                and isinstance(self.neg_iou_thr, (float, tuple))):
            return None
        if not shared and any(b.shape[0] != first.shape[0] for b in boxes):
            return None
        N = first.shape[0]
        counts_host = [int(g.shape[0]) for g in gt_bboxes_list]
        Gmax = max(counts_host)
        if Gmax > 1024 or N == 0:
            return None
        dev = first.device
        if shared:
            bx, stride = first[:, :4].contiguous(), 0
        else:
            bx, stride = torch.stack([b[:, :4] for b in boxes]).contiguous(), N * 4
        if all(c == Gmax for c in counts_host):
            gts = torch.stack([g[:, :4] for g in gt_bboxes_list]).float().contiguous() if Gmax else None
        else:
            gts = first.new_zeros((B, Gmax, 4))
            for i, g in enumerate(gt_bboxes_list):
                gts[i, :counts_host[i]] = g[:, :4]
        with_labels = gt_labels_list is not None and all(l is not None for l in gt_labels_list)
        gl = None
        if with_labels and Gmax:
            if all(c == Gmax for c in counts_host):
                gl = torch.stack(list(gt_labels_list)).long().contiguous()
            else:
                gl = torch.zeros((B, Gmax), dtype=torch.long, device=dev)
                for i, l in enumerate(gt_labels_list):
                    gl[i, :counts_host[i]] = l
        vd = None
        if valids is not None and any(v is not None for v in valids):
            vd = torch.stack([v if v is not None else torch.ones(N, dtype=torch.bool, device=dev)
                              for v in valids]).to(torch.uint8).contiguous()
        gcnt = _pinned_to(torch.tensor(counts_host, dtype=torch.int32), dev)
        gt_inds = torch.empty((B, N), dtype=torch.long, device=dev)
        max_ov = torch.empty((B, N), dtype=torch.float32, device=dev)
        labels = torch.empty((B, N), dtype=torch.long, device=dev) if (with_labels and Gmax) else None
        counts = torch.empty((B, 2), dtype=torch.int32, device=dev)
        L = _lib.lib()
        nbytes = L.oadg_max_iou_assign_workspace_bytes(B, Gmax)
        ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=dev)
        lo, hi = (0.0, self.neg_iou_thr) if isinstance(self.neg_iou_thr, float) else self.neg_iou_thr
        _lib.check(L.oadg_max_iou_assign(_lib.ptr(bx), stride, _lib.ptr(vd), _lib.ptr(gts), _lib.ptr(gcnt),
                                         _lib.ptr(gl), B, N, Gmax, float(self.pos_iou_thr), float(lo), float(hi),
                                         float(self.min_pos_iou), int(bool(self.match_low_quality)), _lib.ptr(ws),
                                         nbytes, _lib.ptr(gt_inds), _lib.ptr(max_ov), _lib.ptr(labels),
                                         _lib.ptr(counts), _lib.stream_ptr()), 'oadg_max_iou_assign')
        res = []
        batch = dict(gt_inds=gt_inds, gts=gts, gt_labels=gl, Gmax=Gmax, boxes=bx if shared else None)
        for i in range(B):
            lab = None
            if with_labels:
                lab = labels[i] if labels is not None else gt_inds.new_full((N,), -1)
            ar = AssignResult(counts_host[i], gt_inds[i], max_ov[i], labels=lab)
            ar.batch = batch            # the whole-batch tensors, for the fused target kernel
            res.append(ar)
        return res, counts

    def assign_wrt_overlaps(self, overlaps, gt_labels=None):
        num_gts, num_bboxes = overlaps.size(0), overlaps.size(1)
        gt_inds = overlaps.new_full((num_bboxes,), -1, dtype=torch.long)
        if num_gts == 0 or num_bboxes == 0:
            max_overlaps = overlaps.new_zeros((num_bboxes,))
            if num_gts == 0:
                gt_inds[:] = 0
            labels = None if gt_labels is None else overlaps.new_full((num_bboxes,), -1, dtype=torch.long)
            return AssignResult(num_gts, gt_inds, max_overlaps, labels=labels)
        max_overlaps, argmax_overlaps = overlaps.max(dim=0)
        gt_max_overlaps, gt_argmax_overlaps = overlaps.max(dim=1)
        if isinstance(self.neg_iou_thr, float):
            gt_inds = gt_inds.masked_fill((max_overlaps >= 0) & (max_overlaps < self.neg_iou_thr), 0)
        elif isinstance(self.neg_iou_thr, tuple):
            gt_inds = gt_inds.masked_fill((max_overlaps >= self.neg_iou_thr[0]) &
                                          (max_overlaps < self.neg_iou_thr[1]), 0)
        pos = max_overlaps >= self.pos_iou_thr
        gt_inds = torch.where(pos, argmax_overlaps + 1, gt_inds)
        if self.match_low_quality:
            ok = gt_max_overlaps >= self.min_pos_iou                               # [k]
            ids = torch.arange(1, num_gts + 1, device=overlaps.device)
            if self.gt_max_assign_all:
                hit = (overlaps == gt_max_overlaps[:, None]) & ok[:, None]         # [k, n]
                last = (hit * ids[:, None]).max(dim=0)[0]                          # later gts overwrite
                gt_inds = torch.where(last > 0, last, gt_inds)
            else:
                for i in range(num_gts):                                           # rare path, order matters
                    if gt_max_overlaps[i] >= self.min_pos_iou:
                        gt_inds[gt_argmax_overlaps[i]] = i + 1
        if gt_labels is not None:
            labels = gt_inds.new_full((num_bboxes,), -1)
            p = gt_inds > 0
            labels = torch.where(p, gt_labels[(gt_inds - 1).clamp(min=0)], labels)
        else:
            labels = None
        return AssignResult(num_gts, gt_inds, max_overlaps, labels=labels)


# ----------------------------------------------------------------------------------------------- sample
class SamplingResult:
    """sampling_result.py.  The gathered views (pos_bboxes, ...) are produced on first use: the RPN's fused
    target kernel never needs them."""

    def __init__(self, pos_inds, neg_inds, bboxes, gt_bboxes, assign_result, gt_flags):
        self.pos_inds, self.neg_inds = pos_inds, neg_inds
        self._src = (bboxes, gt_bboxes, assign_result, gt_flags)
        self.num_gts = gt_bboxes.shape[0]
        self._cache = {}

    def _get(self, name):
        c = self._cache
        if name not in c:
            bboxes, gt_bboxes, ar, gt_flags = self._src
            if name == 'pos_bboxes':
                c[name] = bboxes[self.pos_inds]
            elif name == 'neg_bboxes':
                c[name] = bboxes[self.neg_inds]
            elif name == 'pos_is_gt':
                # (an int: that many leading rows of ``bboxes`` are the gts added as proposals - no flag tensor was built)
                c[name] = (self.pos_inds < gt_flags).to(torch.uint8) if isinstance(gt_flags, int) \
                    else gt_flags[self.pos_inds]
            elif name == 'pos_assigned_gt_inds':
                c[name] = ar.gt_inds[self.pos_inds] - 1
            elif name == 'pos_gt_bboxes':
                if gt_bboxes.numel() == 0:
                    assert self.pos_assigned_gt_inds.numel() == 0
                    c[name] = torch.empty_like(gt_bboxes).view(-1, 4)
                else:
                    g = gt_bboxes.view(-1, 4) if len(gt_bboxes.shape) < 2 else gt_bboxes
                    c[name] = g[self.pos_assigned_gt_inds.long(), :]
            elif name == 'pos_gt_labels':
                c[name] = ar.labels[self.pos_inds] if ar.labels is not None else None
            else:
                raise AttributeError(name)
        return c[name]

    pos_bboxes = property(lambda self: self._get('pos_bboxes'))
    neg_bboxes = property(lambda self: self._get('neg_bboxes'))
    pos_is_gt = property(lambda self: self._get('pos_is_gt'))
    pos_assigned_gt_inds = property(lambda self: self._get('pos_assigned_gt_inds'))
    pos_gt_bboxes = property(lambda self: self._get('pos_gt_bboxes'))
    pos_gt_labels = property(lambda self: self._get('pos_gt_labels'))

    @property
    def bboxes(self):
        return torch.cat([self.pos_bboxes, self.neg_bboxes])


class DeviceSamplingResult:
    """The sampled rows of one image with the positive / negative split on the DEVICE (csrc/roi_sampler.hip): ``sel`` [num]
    holds the sorted positive indices, then the sorted negative indices; ``cnt`` [2] = (k_pos, k_neg).  The fused consumers
    (BBoxHead.rois_and_targets, the fused losses) never need the split on the host; the SamplingResult attributes exist for
    everything else and read the two counts back on first use (a device synchronisation)."""

    def __init__(self, sel, cnt, cap, bboxes, gt_bboxes, assign_result, gt_flags):
        self.sel, self.cnt, self.cap = sel, cnt, int(cap)
        self._src = (bboxes, gt_bboxes, assign_result, gt_flags)
        self.num_gts = gt_bboxes.shape[0]
        self._cache = {}
        self._host = None

    def _split(self):
        if self._host is None:
            kp, kn = (int(v) for v in self.cnt.tolist())
            self._host = (self.sel[:kp], self.sel[kp:kp + kn])
        return self._host

    pos_inds = property(lambda self: self._split()[0])
    neg_inds = property(lambda self: self._split()[1])
    _get = SamplingResult._get
    pos_bboxes = SamplingResult.pos_bboxes
    neg_bboxes = SamplingResult.neg_bboxes
    pos_is_gt = SamplingResult.pos_is_gt
    pos_assigned_gt_inds = SamplingResult.pos_assigned_gt_inds
    pos_gt_bboxes = SamplingResult.pos_gt_bboxes
    pos_gt_labels = SamplingResult.pos_gt_labels
    bboxes = SamplingResult.bboxes


# Device-side RoI sampling is SPECULATIVE about one thing: that every image yields the sampler's full ``num`` rows (1000
# proposals + gts against num = 512: always, in practice), which fixes every downstream shape without a host read.  The
# trainer opens a record per step (begin_speculation), every device sampling appends its flag buffer + generator, and the
# trainer checks the flags before the optimizer step - by then the copy has long landed - and repeats the step through the
# host path if an image came up short (apis.TrainEngine._step).  Outside such a record the host path runs.
DEVICE_SAMPLER = os.environ.get('OADG_DEVICE_SAMPLER', '1') == '1'
_SPEC = None


_SPEC_GROUP = None          # the process group whose ranks must take the same repeat decision (None: single process)


def begin_speculation(group=None, collective=False):
    """``collective``: more than one rank steps in lockstep - the flags of every sampler call are all-reduced (MAX) over
    ``group`` so that all ranks repeat a step together; EVERY rank must then issue that collective for every call,
    whatever its own data looks like (see _finish_device)."""
    global _SPEC, _SPEC_GROUP
    _SPEC = [] if DEVICE_SAMPLER else None
    _SPEC_GROUP = (group, True) if collective else None


def end_speculation():
    global _SPEC
    recs, _SPEC = _SPEC, None
    return recs or []


@BBOX_SAMPLERS.register_module()
class RandomSampler:
    """base_sampler.py:38-103 + random_sampler.py:32-82."""

    def __init__(self, num, pos_fraction, neg_pos_ub=-1, add_gt_as_proposals=True, **kwargs):
        self.num, self.pos_fraction = num, pos_fraction
        self.neg_pos_ub, self.add_gt_as_proposals = neg_pos_ub, add_gt_as_proposals

    @staticmethod
    def random_choice(gallery, num):
        assert len(gallery) >= num
        # CPU generator, then moved: the draw depends only on gallery.numel() (random_sampler.py:58)
        perm = randperm_prefix(gallery.numel(), num).to(device=gallery.device)
        return gallery[perm]

    def _sample(self, mask, num_expected):
        inds = torch.nonzero(mask, as_tuple=False)
        if inds.numel() != 0:
            inds = inds.squeeze(1)
        if inds.numel() <= num_expected:
            return inds
        return self.random_choice(inds, num_expected)

    def sample(self, assign_result, bboxes, gt_bboxes, gt_labels=None, **kwargs):
        if len(bboxes.shape) < 2:
            bboxes = bboxes[None, :]
        bboxes = bboxes[:, :4]
        gt_flags = bboxes.new_zeros((bboxes.shape[0],), dtype=torch.uint8)
        if self.add_gt_as_proposals and len(gt_bboxes) > 0:
            if gt_labels is None:
                raise ValueError('gt_labels must be given when add_gt_as_proposals is True')
            bboxes = torch.cat([gt_bboxes, bboxes], dim=0)
            assign_result.add_gt_(gt_labels)
            gt_flags = torch.cat([bboxes.new_ones(gt_bboxes.shape[0], dtype=torch.uint8), gt_flags])
        num_pos = int(self.num * self.pos_fraction)
        pos_inds = self._sample(assign_result.gt_inds > 0, num_pos).unique()
        num_neg = self.num - pos_inds.numel()
        if self.neg_pos_ub >= 0:
            num_neg = min(num_neg, int(self.neg_pos_ub * max(1, pos_inds.numel())))
        neg_inds = self._sample(assign_result.gt_inds == 0, num_neg).unique()
        return SamplingResult(pos_inds, neg_inds, bboxes, gt_bboxes, assign_result, gt_flags)


def _pinned_to(t_cpu, device):
    """async H2D of a small host tensor (pinned staging; never blocks the host on the stream)."""
    if device.type != 'cuda':
        return t_cpu.to(device)
    return t_cpu.pin_memory().to(device, non_blocking=True)


_JOB_DTYPE = np.dtype([('gt_inds', np.uint64), ('n', np.int64), ('mode', np.int32), ('k', np.int32),
                       ('all', np.int32), ('rank_off', np.int32), ('out_off', np.int64)])      # oadg_select_job


class PendingSampling:
    """RandomSampler.sample for a list of images, split in two so that the single device->host read never
    stalls the stream: :func:`sample_many_begin` enqueues the candidate masks and an asynchronous copy of the
    candidate counts (pinned buffer + event recorded right behind it); ``finish()`` waits for THAT event only -
    work enqueued in between keeps the device busy - draws the permutations on the host in image order and
    locates the chosen candidates on the device by their rank among the candidates.

    Equivalent, image by image, to ``sampler.sample(...)`` (base_sampler.py:38-103, random_sampler.py:32-82):
    same candidates, same ``torch.randperm(n)[:num]`` draws from the global CPU generator (positives first, then
    negatives), same sorted index lists."""

    def __init__(self, sampler, prepared, gt_bboxes_list, counts_dev, added_pos=None):
        self.sampler, self.prepared, self.gt_bboxes_list = sampler, prepared, gt_bboxes_list
        self.added_pos = added_pos
        self.results = None
        self.event = None
        if counts_dev is None:
            self.counts = []
        elif counts_dev.is_cuda:
            self.counts = torch.empty(counts_dev.shape, dtype=counts_dev.dtype, pin_memory=True)
            self.counts.copy_(counts_dev, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record()
        else:
            self.counts = counts_dev

    def _plan(self):
        """Host side of the draw, image by image in the reference's order (positives, then negatives):
        [(k_pos, ranks_pos | None, k_neg, ranks_neg | None)]; ``None`` = every candidate."""
        sampler = self.sampler
        counts = self.counts.tolist() if isinstance(self.counts, torch.Tensor) else self.counts
        if self.added_pos is not None:
            counts = [(c[0] + a, c[1]) for c, a in zip(counts, self.added_pos)]
        num_pos_exp = int(sampler.num * sampler.pos_fraction)
        plan = []
        for n_pos, n_neg in counts:
            if n_pos <= num_pos_exp:
                k_pos, r_pos = n_pos, None
            else:
                k_pos, r_pos = num_pos_exp, randperm_prefix(n_pos, num_pos_exp)      # random_sampler.py:58
            n_neg_exp = sampler.num - k_pos
            if sampler.neg_pos_ub >= 0:
                n_neg_exp = min(n_neg_exp, int(sampler.neg_pos_ub * max(1, k_pos)))
            if n_neg <= n_neg_exp:
                k_neg, r_neg = n_neg, None
            else:
                k_neg, r_neg = n_neg_exp, randperm_prefix(n_neg, n_neg_exp)
            plan.append((k_pos, r_pos, k_neg, r_neg))
        return plan

    def _select_device(self, plan, dev):
        """Locate the planned candidates with csrc/targets.hip (two launches for the whole batch)."""
        from .. import _lib
        nj = 2 * len(plan)
        jobs = np.zeros(nj, dtype=_JOB_DTYPE)
        ranks, out_off, max_n = [], 0, 0
        for i, (k_pos, r_pos, k_neg, r_neg) in enumerate(plan):
            gi = self.prepared[i][0].gt_inds
            assert gi.is_contiguous() and gi.dtype == torch.long
            for j, (k, r) in enumerate(((k_pos, r_pos), (k_neg, r_neg))):
                jb = jobs[2 * i + j]
                jb['gt_inds'], jb['n'], jb['mode'], jb['k'] = gi.data_ptr(), gi.numel(), j, k
                jb['all'], jb['rank_off'], jb['out_off'] = int(r is None), sum(len(x) for x in ranks), out_off
                if r is not None:
                    ranks.append(np.sort(r.numpy()).astype(np.int32))
                out_off += k
            max_n = max(max_n, gi.numel())
        rk = np.concatenate(ranks) if ranks else np.zeros(1, np.int32)
        blob = np.concatenate([jobs.view(np.uint8), rk.view(np.uint8)])
        blob_dev = _pinned_to(torch.from_numpy(blob), dev)
        jobs_dev = blob_dev[:jobs.nbytes]
        ranks_dev = blob_dev[jobs.nbytes:]
        sel = torch.empty(max(out_off, 1), dtype=torch.long, device=dev)
        L = _lib.lib()
        nbytes = L.oadg_sample_select_workspace_bytes(nj, max_n)
        ws = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=dev)
        _lib.check(L.oadg_sample_select(_lib.ptr(jobs_dev), nj, max_n, int(jobs['k'].max()) if nj else 0,
                                        _lib.ptr(ranks_dev), _lib.ptr(sel),
                                        _lib.ptr(ws), nbytes, _lib.stream_ptr()), 'oadg_sample_select')
        self.device_select = dict(jobs_dev=jobs_dev, sel=sel, jobs=jobs, blob=blob_dev,
                                  max_k=int(jobs['k'].max()) if nj else 0)
        return [(sel[int(jobs[2 * i]['out_off']):int(jobs[2 * i]['out_off']) + plan[i][0]],
                 sel[int(jobs[2 * i + 1]['out_off']):int(jobs[2 * i + 1]['out_off']) + plan[i][2]])
                for i in range(len(plan))]

    def _finish_device(self):
        """RandomSampler.sample of every image in ONE launch on the device (csrc/roi_sampler.hip), the CPU generator's
        engine state handed over and back (device_rng): no host read.  None when the inputs are outside the kernel's
        domain (the host path then runs)."""
        from .. import _lib, device_rng
        sampler, B = self.sampler, len(self.prepared)
        if not (type(sampler) is RandomSampler and 0 < B <= _lib.ROI_SAMPLE_MAX_IMAGES and 0 < sampler.num <= 512 and
                getattr(self, '_scratch', None) is not None):
            return None
        L = _lib.lib()
        images = (_lib.RoiSampleImage * B)()
        max_rows = L.oadg_roi_sample_max_rows()
        dev = self.prepared[0][1].device
        import torch.distributed as dist
        for i, prep in enumerate(self.prepared):
            gi = prep[0].gt_inds
            if not (gi.is_cuda and gi.dtype == torch.long and gi.is_contiguous() and gi.numel() <= max_rows):
                # data-dependent: the other ranks may be eligible and will issue the flags collective - take part in it
                # with "outside the domain" flags (every rank then repeats the step on the host path) and draw on the host
                if _SPEC_GROUP is not None:
                    meta = torch.zeros(3 * B, dtype=torch.int32, device=dev)
                    meta[2 * B:] = 2
                    dist.all_reduce(meta[2 * B:], op=dist.ReduceOp.MAX, group=_SPEC_GROUP[0])
                    host = torch.empty(3 * B, dtype=torch.int32).pin_memory()
                    host.copy_(meta, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record()
                    _SPEC.append(dict(meta=host, B=B, gen=None, event=ev))
                return None
            images[i].gt_inds, images[i].n = gi.data_ptr(), gi.numel()
        num = int(sampler.num)
        sel = torch.empty((B, num), dtype=torch.long, device=dev)
        meta = torch.empty(3 * B, dtype=torch.int32, device=dev)          # counts [B][2] | flags [B]
        counts, flags = meta[:2 * B].view(B, 2), meta[2 * B:]
        gen = device_rng.generator(dev)
        if gen.pending():
            gen.sync_host()
        state, state_out = gen.upload()
        _lib.check(L.oadg_roi_sample_device(ctypes.cast(images, ctypes.c_void_p), B, num, int(num * sampler.pos_fraction),
                                            float(sampler.neg_pos_ub), _lib.ptr(state), _lib.ptr(state_out), _lib.ptr(sel), _lib.ptr(counts),
                                            _lib.ptr(flags), _lib.stream_ptr()), 'oadg_roi_sample_device')
        if _SPEC_GROUP is not None:
            # every rank must take the same decision about repeating the step (its collectives): share the flags
            dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=_SPEC_GROUP[0])
        host = torch.empty(3 * B, dtype=torch.int32).pin_memory()
        host.copy_(meta, non_blocking=True)
        gen.download_async()                      # (its event also covers the copy above: same stream)
        _SPEC.append(dict(meta=host, B=B, gen=gen))
        self.results = [DeviceSamplingResult(sel[i], counts[i], num, prep[1], self.gt_bboxes_list[i], prep[0], prep[2])
                        for i, prep in enumerate(self.prepared)]
        return self.results

    def finish(self):
        if self.results is not None:
            return self.results
        if _SPEC is not None and self.prepared and self.prepared[0][1].is_cuda:
            res = self._finish_device()
            if res is not None:
                return res
        if self.event is not None:
            self.event.synchronize()
        plan = self._plan()
        dev = self.prepared[0][1].device if self.prepared else None
        if dev is not None and dev.type == 'cuda':
            picked = self._select_device(plan, dev)
        else:
            picked = []
            for (ar, bboxes, gt_flags, pos_mask, neg_mask), (k_pos, r_pos, k_neg, r_neg) in zip(self.prepared, plan):
                def choose(mask, k, perm):
                    if k == 0:
                        return torch.zeros((0,), dtype=torch.long, device=mask.device)
                    if perm is None:
                        return torch.nonzero(mask, as_tuple=False).squeeze(1)
                    cand = torch.nonzero(mask, as_tuple=False).squeeze(1)
                    return cand[perm.to(cand.device)].unique()
                picked.append((choose(pos_mask(), k_pos, r_pos), choose(neg_mask(), k_neg, r_neg)))
        self.results = [SamplingResult(p, n, prep[1], self.gt_bboxes_list[i], prep[0], prep[2])
                        for i, (prep, (p, n)) in enumerate(zip(self.prepared, picked))]
        return self.results


def sample_many_begin(sampler, assign_results, bboxes_list, gt_bboxes_list, gt_labels_list=None, counts=None):
    """``counts`` ([B,2] int32 on the device, from MaxIoUAssigner.assign_many): candidate counts of the
    assignment BEFORE the gts are added as proposals; without it they are reduced from the masks here."""
    prepared = []
    added = []
    for i in range(len(assign_results)):
        ar, bboxes, gtb = assign_results[i], bboxes_list[i], gt_bboxes_list[i]
        if len(bboxes.shape) < 2:
            bboxes = bboxes[None, :]
        bboxes = bboxes[:, :4]
        gt_flags = bboxes.new_zeros((bboxes.shape[0],), dtype=torch.uint8)
        if sampler.add_gt_as_proposals and len(gtb) > 0:
            if gt_labels_list is None or gt_labels_list[i] is None:
                raise ValueError('gt_labels must be given when add_gt_as_proposals is True')
            bboxes = torch.cat([gtb, bboxes], dim=0)
            ar.add_gt_(gt_labels_list[i])
            gt_flags = torch.cat([bboxes.new_ones(gtb.shape[0], dtype=torch.uint8), gt_flags])
            added.append(int(gtb.shape[0]))          # every added gt is a positive candidate (assign_result.py)
        else:
            added.append(0)
        prepared.append((ar, bboxes, gt_flags, (lambda a_=ar: a_.gt_inds > 0), (lambda a_=ar: a_.gt_inds == 0)))
    if not prepared:
        counts = None
    elif counts is None:
        counts = torch.stack([torch.stack([p[3]().sum(), p[4]().sum()]) for p in prepared])
        added = None                                  # the masks already contain the added gts
    return PendingSampling(sampler, prepared, gt_bboxes_list, counts, added)


ROI_ASSIGN_FUSED = os.environ.get('OADG_FUSED_ROI_ASSIGN', '1') == '1'


def roi_assign_sample_begin(assigner, sampler, proposals, gt_bboxes_list, gt_labels_list):
    """``assigner.assign`` over the proposals of every image + the head of ``sampler.sample`` (gts added as proposals:
    boxes, gt_inds, labels, overlaps with the gt rows in front) in THREE launches of csrc/assign.hip
    (``oadg_roi_assign_add_gt``) instead of ~48 (per image: validity mask, stack, assignment, arange / ones / five
    concatenations): every image's tensors are contiguous row ranges of four batch tensors.  Returns the
    :class:`PendingSampling` handle ``sample_many_begin`` would, or None when the inputs are outside the kernels' domain
    (standard_roi_head.py:88-101, base_sampler.py:38-78, assign_result.py add_gt_)."""
    from .. import _lib
    B = len(proposals)
    if not (ROI_ASSIGN_FUSED and 0 < B <= _lib.ROI_ASSIGN_MAX_IMAGES and sampler.add_gt_as_proposals and
            assigner.gt_max_assign_all and isinstance(assigner.neg_iou_thr, (float, tuple)) and
            gt_labels_list is not None and len(gt_bboxes_list) == B):
        return None
    first = proposals[0]
    N = first.shape[0]
    if not (first.is_cuda and N > 0):
        return None
    counts_host = [int(g.shape[0]) for g in gt_bboxes_list]
    Gmax = max(counts_host)
    if Gmax > 1024:
        return None
    images = (_lib.RoiAssignImage * B)()
    keep = []
    for i in range(B):
        p, g, l = proposals[i], gt_bboxes_list[i], gt_labels_list[i]
        if p.dtype != torch.float32 or p.dim() != 2 or p.shape[0] != N or p.stride(1) != 1 or \
                not (p.shape[1] == 5 or (p.shape[1] == 4 and p.stride(0) == 4)) or \
                l is None or g.dtype != torch.float32 or l.dtype != torch.long or g.device != p.device:
            return None         # ([N, 5]: column 4 is the score, negative on padding rows; [N, 4] dense: all rows valid)
        g = g.view(-1, 4) if g.dim() < 2 else g[:, :4]
        g, l = g.contiguous(), l.contiguous()
        keep.extend((g, l))
        im = images[i]
        im.proposals, im.stride, im.num_gts = p.data_ptr(), p.stride(0), counts_host[i]
        im.gt_bboxes, im.gt_labels = (g.data_ptr(), l.data_ptr()) if counts_host[i] else (None, None)
    dev, rows = first.device, Gmax + N
    boxes_full = torch.empty((B, rows, 4), dtype=torch.float32, device=dev)
    gt_inds_full = torch.empty((B, rows), dtype=torch.long, device=dev)
    labels_full = torch.empty((B, rows), dtype=torch.long, device=dev)
    max_ov_full = torch.empty((B, rows), dtype=torch.float32, device=dev)
    L = _lib.lib()
    nws = L.oadg_max_iou_assign_workspace_bytes(B, Gmax)
    # scratch in one allocation: valid [B][N] u8 | gts_pad [B][Gmax][4] f32 | gl_pad [B][Gmax] i64 | gt_counts [B] i32 |
    # workspace | counts [B][2] i32, each part 16-byte aligned
    sizes = [B * N, B * Gmax * 16, B * Gmax * 8, B * 4, nws, B * 8]
    offs, at = [], 0
    for n in sizes:
        offs.append(at)
        at += (n + 15) // 16 * 16
    scratch = torch.empty(max(at, 16), dtype=torch.uint8, device=dev)
    base = scratch.data_ptr()
    pp = [ctypes.c_void_p(base + o) for o in offs]
    counts = scratch[offs[5]:offs[5] + B * 8].view(torch.int32).view(B, 2)
    lo, hi = (0.0, assigner.neg_iou_thr) if isinstance(assigner.neg_iou_thr, float) else assigner.neg_iou_thr
    _lib.check(L.oadg_roi_assign_add_gt(ctypes.cast(images, ctypes.c_void_p), B, N, Gmax, float(assigner.pos_iou_thr),
                                        float(lo), float(hi), float(assigner.min_pos_iou),
                                        int(bool(assigner.match_low_quality)), _lib.ptr(boxes_full),
                                        _lib.ptr(gt_inds_full), _lib.ptr(labels_full), _lib.ptr(max_ov_full), pp[0],
                                        pp[1] if Gmax else None, pp[2] if Gmax else None, pp[3], pp[4], nws, pp[5],
                                        _lib.stream_ptr()), 'oadg_roi_assign_add_gt')
    del keep
    prepared, added = [], []
    for i in range(B):
        s0 = Gmax - counts_host[i]
        ar = AssignResult(counts_host[i], gt_inds_full[i, s0:], max_ov_full[i, s0:], labels=labels_full[i, s0:])
        prepared.append((ar, boxes_full[i, s0:], counts_host[i],
                         (lambda a_=ar: a_.gt_inds > 0), (lambda a_=ar: a_.gt_inds == 0)))
        added.append(counts_host[i])
    pend = PendingSampling(sampler, prepared, gt_bboxes_list, counts, added)
    pend._scratch = scratch          # (counts is a view of it; the asynchronous read-back has its own pinned copy)
    return pend


def sample_many(sampler, assign_results, bboxes_list, gt_bboxes_list, gt_labels_list=None):
    """begin + finish in one call."""
    return sample_many_begin(sampler, assign_results, bboxes_list, gt_bboxes_list, gt_labels_list).finish()


# ----------------------------------------------------------------------------------------------- coder
_CONSTS = {}


def _const(values, like):
    """small constant tensor on ``like``'s device, built once (new_tensor copies host->device every call)."""
    key = (tuple(float(v) for v in values), like.dtype, str(like.device))
    t = _CONSTS.get(key)
    if t is None:
        t = _CONSTS[key] = torch.tensor(key[0], dtype=like.dtype, device=like.device)
    return t


def bbox2delta(proposals, gt, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.)):
    """delta_xywh_bbox_coder.py:119-180, including the fork's zero-size guard (:152-160) and its
    mismatched mask ``gy[nan_x] = py[nan_y]``."""
    assert proposals.size() == gt.size()
    proposals, gt = proposals.float(), gt.float()
    px = (proposals[..., 0] + proposals[..., 2]) * 0.5
    py = (proposals[..., 1] + proposals[..., 3]) * 0.5
    pw = proposals[..., 2] - proposals[..., 0]
    ph = proposals[..., 3] - proposals[..., 1]
    gx = (gt[..., 0] + gt[..., 2]) * 0.5
    gy = (gt[..., 1] + gt[..., 3]) * 0.5
    gw = gt[..., 2] - gt[..., 0]
    gh = gt[..., 3] - gt[..., 1]
    # fork guard against zero-size proposals (:152-160), written without boolean indexing (which reads the
    # mask count back to the host).  Its last line is ``gy[nan_x] = py[nan_y]``: a POSITIONAL pairing - the k-th
    # zero-width row receives py of the k-th zero-height row (the reference raises when the two counts differ; here
    # the surplus rows read the next rows in order).  Reproduced through ranks instead of compaction.
    nan_x, nan_y = (pw == 0), (ph == 0)
    tiny = pw.new_full((), 1e-6)
    pw = torch.where(nan_x, tiny, pw)
    ph = torch.where(nan_y, tiny, ph)
    gw = torch.where(nan_x, tiny, gw)
    gh = torch.where(nan_y, tiny, gh)
    gx = torch.where(nan_x, px, gx)
    flat_x, flat_y = nan_x.reshape(-1), nan_y.reshape(-1)
    if flat_x.numel() > 0:
        rank = (torch.cumsum(flat_x, 0) - 1).clamp(min=0)
        order_y = torch.sort((~flat_y).to(torch.uint8), stable=True)[1]          # zero-height rows first, in order
        src = order_y[rank].reshape(nan_x.shape)
        gy = torch.where(nan_x, py.reshape(-1)[src], gy)
    deltas = torch.stack([(gx - px) / pw, (gy - py) / ph, torch.log(gw / pw), torch.log(gh / ph)], dim=-1)
    means = _const(means, deltas).unsqueeze(0)
    stds = _const(stds, deltas).unsqueeze(0)
    return deltas.sub_(means).div_(stds)


def delta2bbox(rois, deltas, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.), max_shape=None,
               wh_ratio_clip=16 / 1000, clip_border=True, add_ctr_clamp=False, ctr_clamp=32):
    """delta_xywh_bbox_coder.py:184-..."""
    num_bboxes, num_classes = deltas.size(0), deltas.size(1) // 4
    if num_bboxes == 0:
        return deltas
    deltas = deltas.reshape(-1, 4)
    means = _const(means, deltas).view(1, -1)
    stds = _const(stds, deltas).view(1, -1)
    d = deltas * stds + means
    dxy, dwh = d[:, :2], d[:, 2:]
    rois_ = rois.repeat(1, num_classes).reshape(-1, 4)
    pxy = (rois_[:, :2] + rois_[:, 2:]) * 0.5
    pwh = rois_[:, 2:] - rois_[:, :2]
    dxy_wh = pwh * dxy
    max_ratio = np.abs(np.log(wh_ratio_clip))
    if add_ctr_clamp:
        dxy_wh = torch.clamp(dxy_wh, max=ctr_clamp, min=-ctr_clamp)
        dwh = torch.clamp(dwh, max=max_ratio)
    else:
        dwh = dwh.clamp(min=-max_ratio, max=max_ratio)
    gxy = pxy + dxy_wh
    gwh = pwh * dwh.exp()
    bboxes = torch.cat([gxy - gwh * 0.5, gxy + gwh * 0.5], dim=-1)
    if clip_border and max_shape is not None:
        bboxes[..., 0::2].clamp_(min=0, max=max_shape[1])
        bboxes[..., 1::2].clamp_(min=0, max=max_shape[0])
    return bboxes.reshape(num_bboxes, -1)


@BBOX_CODERS.register_module()
class DeltaXYWHBBoxCoder:

    def __init__(self, target_means=(0., 0., 0., 0.), target_stds=(1., 1., 1., 1.), clip_border=True,
                 add_ctr_clamp=False, ctr_clamp=32):
        self.means, self.stds = target_means, target_stds
        self.clip_border, self.add_ctr_clamp, self.ctr_clamp = clip_border, add_ctr_clamp, ctr_clamp

    def encode(self, bboxes, gt_bboxes):
        assert bboxes.size(0) == gt_bboxes.size(0) and bboxes.size(-1) == gt_bboxes.size(-1) == 4
        return bbox2delta(bboxes, gt_bboxes, self.means, self.stds)

    def decode(self, bboxes, pred_bboxes, max_shape=None, wh_ratio_clip=16 / 1000):
        assert pred_bboxes.size(0) == bboxes.size(0)
        return delta2bbox(bboxes, pred_bboxes, self.means, self.stds, max_shape, wh_ratio_clip,
                          self.clip_border, self.add_ctr_clamp, self.ctr_clamp)


def bbox2roi(bbox_list):
    """list of [n_i, >=4] boxes -> [sum n_i, 5] with the list position as batch index (transforms.py:75-94)."""
    sizes = [int(b.size(0)) for b in bbox_list]
    if len(bbox_list) > 1 and sum(sizes) > 0 and bbox_list[0].is_cuda:
        # one index column for the whole batch (built on the host from the known sizes, uploaded through pinned memory)
        # and two concatenations, instead of a fill + a concatenation per image
        first = bbox_list[0]
        idx = _pinned_to(torch.from_numpy(np.repeat(np.arange(len(sizes), dtype=np.float32), sizes)), first.device)
        boxes = torch.cat([b[:, :4] for b in bbox_list], 0)
        return torch.cat([idx.to(boxes.dtype)[:, None], boxes], dim=1)
    rois = []
    for img_id, b in enumerate(bbox_list):
        if b.size(0) > 0:
            rois.append(torch.cat([b.new_full((b.size(0), 1), img_id), b[:, :4]], dim=-1))
        else:
            rois.append(b.new_zeros((0, 5)))
    return torch.cat(rois, 0)


def randperm_prefix(n, k):
    """``torch.randperm(n)[:k]`` on the global CPU generator - same values, same generator state afterwards -
    through the O(k) replay in csrc/host_rng.hip (ATen's shuffle touches all n entries: ~10-70 ms at the
    RPN's n ~ 520k candidates, per image)."""
    import ctypes
    from .. import _lib
    if n < 4096:
        return torch.randperm(n)[:k]
    L = _lib.lib()
    st = torch.get_rng_state()
    a = st.numpy()
    assert a.size == 5056, 'unexpected CPU generator state layout'
    left = ctypes.c_int(int(a[8:12].view(np.int32)[0]))
    nxt = ctypes.c_uint64(int(a[16:24].view(np.uint64)[0]))
    state = a[24:24 + 624 * 8].view(np.uint64).copy()
    k = min(k, n)
    out = np.empty(k, np.int64)
    rc = L.oadg_host_randperm_prefix(state.ctypes.data, ctypes.byref(left), ctypes.byref(nxt), n, k,
                                     out.ctypes.data)
    if rc != 0:
        raise RuntimeError('oadg_host_randperm_prefix failed')
    a[24:24 + 624 * 8] = state.view(np.uint8)
    a[8:12] = np.array([left.value], np.int32).view(np.uint8)
    a[16:24] = np.array([nxt.value], np.uint64).view(np.uint8)
    torch.set_rng_state(st)
    return torch.from_numpy(out)
