"""RoI extractor, bbox heads and RoI heads (mmdet/models/roi_heads/**).

RoI feature extraction is one HIP launch over the whole pyramid (level mapping in-kernel) instead of the
reference's per-level nonzero/gather/scatter loop; every pyramid level is therefore always in the autograd
graph (the reference needs a dummy-graph trick for that, single_level_roi_extractor.py:136-145)."""
import ctypes
import os

import numpy as np
import torch
from torch.profiler import record_function as _rf
import torch.nn as nn
import torch.nn.functional as F

from . import hip_ops
from .core import bbox2roi, multi_apply
from .core.bbox import DeviceSamplingResult, _pinned_to, roi_assign_sample_begin, sample_many, sample_many_begin
from .layers import normal_init, xavier_init
from .losses import accuracy
from .registry import (HEADS, ROI_EXTRACTORS, ROI_LAYERS, build_assigner, build_bbox_coder, build_head,
                       build_loss, build_roi_extractor, build_sampler)


def _autocast_dtype():
    """the CUDA autocast dtype (torch >= 2.4: get_autocast_dtype; older builds: get_autocast_gpu_dtype)"""
    f = getattr(torch, 'get_autocast_dtype', None)
    return f('cuda') if f is not None else torch.get_autocast_gpu_dtype()


def _pair(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


@ROI_LAYERS.register_module()
class RoIAlign(nn.Module):
    """mmcv.ops.RoIAlign signature: forward(feats [N,C,H,W], rois [K,5]) -> [K,C,oh,ow] (HIP, single map)."""

    def __init__(self, output_size, spatial_scale=1.0, sampling_ratio=0, pool_mode='avg', aligned=True,
                 use_torchvision=False):
        super().__init__()
        assert pool_mode == 'avg', 'only average pooling is on the OA-DG path'
        self.output_size = _pair(output_size)
        self.spatial_scale, self.sampling_ratio, self.aligned = float(spatial_scale), int(sampling_ratio), aligned

    def forward(self, input, rois):
        assert rois.dim() == 2 and rois.size(1) == 5
        return hip_ops.roi_align_fpn([input], rois, self.output_size, [self.spatial_scale],
                                     sampling_ratio=self.sampling_ratio, aligned=self.aligned)


@ROI_EXTRACTORS.register_module()
class SingleRoIExtractor(nn.Module):
    """base_roi_extractor.py:9-86 + single_level_roi_extractor.py:9-146."""

    def __init__(self, roi_layer, out_channels, featmap_strides, finest_scale=56, init_cfg=None):
        super().__init__()
        cfg = dict(roi_layer)
        layer_type = cfg.pop('type')
        layer_cls = ROI_LAYERS.get(layer_type)
        if layer_cls is None:
            raise KeyError(f'{layer_type} is not a known RoI layer')
        self.roi_layers = nn.ModuleList([layer_cls(spatial_scale=1 / s, **cfg) for s in featmap_strides])
        self.out_channels, self.featmap_strides, self.finest_scale = out_channels, featmap_strides, finest_scale

    @property
    def num_inputs(self):
        return len(self.featmap_strides)

    def forward(self, feats, rois, roi_scale_factor=None):
        assert roi_scale_factor is None, 'roi_scale_factor is not used on the OA-DG path'
        l0 = self.roi_layers[0]
        feats = feats[:self.num_inputs]
        if len(rois) == 0:
            return feats[0].new_zeros(0, self.out_channels, *l0.output_size)
        return hip_ops.roi_align_fpn(list(feats), rois, l0.output_size,
                                     [layer.spatial_scale for layer in self.roi_layers],
                                     finest_scale=self.finest_scale, sampling_ratio=l0.sampling_ratio,
                                     aligned=l0.aligned)


class BBoxHead(nn.Module):
    """bbox_head.py (targets + stock loss) - base of the conv-fc heads."""

    def __init__(self, with_avg_pool=False, with_cls=True, with_reg=True, roi_feat_size=7, in_channels=256,
                 num_classes=80,
                 bbox_coder=dict(type='DeltaXYWHBBoxCoder', clip_border=True, target_means=[0., 0., 0., 0.],
                                 target_stds=[0.1, 0.1, 0.2, 0.2]),
                 reg_class_agnostic=False, reg_decoded_bbox=False, reg_predictor_cfg=dict(type='Linear'),
                 cls_predictor_cfg=dict(type='Linear'),
                 loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0),
                 loss_bbox=dict(type='SmoothL1Loss', beta=1.0, loss_weight=1.0), init_cfg=None):
        super().__init__()
        assert with_cls or with_reg
        assert not with_avg_pool and not reg_decoded_bbox
        assert reg_predictor_cfg['type'] == 'Linear' and cls_predictor_cfg['type'] == 'Linear'
        self.with_avg_pool, self.with_cls, self.with_reg = with_avg_pool, with_cls, with_reg
        self.roi_feat_size = _pair(roi_feat_size)
        self.roi_feat_area = self.roi_feat_size[0] * self.roi_feat_size[1]
        self.in_channels, self.num_classes = in_channels, num_classes
        self.reg_class_agnostic, self.reg_decoded_bbox = reg_class_agnostic, reg_decoded_bbox
        self.bbox_coder = build_bbox_coder(bbox_coder)
        self.loss_cls = build_loss(loss_cls)
        self.loss_bbox = build_loss(loss_bbox)
        self.custom_activation = False

    def _make_predictors(self, cls_dim, reg_dim):
        if self.with_cls:
            self.fc_cls = nn.Linear(cls_dim, self.num_classes + 1)
        if self.with_reg:
            self.fc_reg = nn.Linear(reg_dim, 4 if self.reg_class_agnostic else 4 * self.num_classes)

    # -- targets (bbox_head.py:190-257,328-394) -------------------------------------------------------
    def _get_target_single(self, pos_bboxes, neg_bboxes, pos_gt_bboxes, pos_gt_labels, cfg):
        num_pos, num_neg = pos_bboxes.size(0), neg_bboxes.size(0)
        n = num_pos + num_neg
        labels = pos_bboxes.new_full((n,), self.num_classes, dtype=torch.long)
        label_weights = pos_bboxes.new_zeros(n)
        bbox_targets = pos_bboxes.new_zeros(n, 4)
        bbox_weights = pos_bboxes.new_zeros(n, 4)
        absolute = pos_bboxes.new_zeros(n, 4)
        if num_pos > 0:
            labels[:num_pos] = pos_gt_labels
            label_weights[:num_pos] = 1.0 if cfg.pos_weight <= 0 else cfg.pos_weight
            bbox_targets[:num_pos, :] = self.bbox_coder.encode(pos_bboxes, pos_gt_bboxes)
            bbox_weights[:num_pos, :] = 1
            absolute[:num_pos, :] = pos_gt_bboxes
        if num_neg > 0:
            label_weights[-num_neg:] = 1.0
        return labels, label_weights, bbox_targets, bbox_weights, absolute

    def get_targets_with_absolute(self, sampling_results, gt_bboxes, gt_labels, rcnn_train_cfg, concat=True):
        if concat and len(sampling_results) > 0 and sampling_results[0].pos_inds.is_cuda:
            return self._get_targets_batched(sampling_results, rcnn_train_cfg)
        out = multi_apply(self._get_target_single, [r.pos_bboxes for r in sampling_results],
                          [r.neg_bboxes for r in sampling_results],
                          [r.pos_gt_bboxes for r in sampling_results],
                          [r.pos_gt_labels for r in sampling_results], cfg=rcnn_train_cfg)
        return tuple(torch.cat(o, 0) for o in out) if concat else out

    def _get_targets_batched(self, sampling_results, cfg):
        """_get_target_single for every image in a dozen launches: one encode over all positives, rows placed by
        host-known offsets (positives first, then negatives, image by image - bbox_head.py:190-257).  Element for
        element the same arithmetic as the per-image form."""
        dev = sampling_results[0].pos_inds.device
        rows, total = [], 0
        for r in sampling_results:
            npos, nneg = r.pos_inds.numel(), r.neg_inds.numel()
            rows.append(torch.arange(total, total + npos))
            total += npos + nneg
        pos_rows = _pinned_to(torch.cat(rows), dev)
        like = sampling_results[0].pos_bboxes
        labels = like.new_full((total,), self.num_classes, dtype=torch.long)
        label_weights = like.new_ones(total)             # every sampled row carries weight 1 ...
        bbox_targets = like.new_zeros(total, 4)
        bbox_weights = like.new_zeros(total, 4)
        absolute = like.new_zeros(total, 4)
        if pos_rows.numel() > 0:
            pb = torch.cat([r.pos_bboxes for r in sampling_results])
            pg = torch.cat([r.pos_gt_bboxes for r in sampling_results])
            pl = torch.cat([r.pos_gt_labels for r in sampling_results])
            labels.index_copy_(0, pos_rows, pl)
            if cfg.pos_weight > 0:                       # ... unless the config re-weights the positives
                label_weights.index_fill_(0, pos_rows, float(cfg.pos_weight))
            bbox_targets.index_copy_(0, pos_rows, self.bbox_coder.encode(pb, pg))
            bbox_weights.index_fill_(0, pos_rows, 1.0)
            absolute.index_copy_(0, pos_rows, pg)
        return labels, label_weights, bbox_targets, bbox_weights, absolute

    FUSED_TARGETS = os.environ.get('OADG_FUSED_ROI_TARGETS', '1') != '0'

    def rois_and_targets(self, sampling_results, rcnn_train_cfg, extra=()):
        """``bbox2roi([r.bboxes ...])`` + ``get_targets_with_absolute`` (+ ``bbox2roi`` of the ``extra`` box lists,
        appended) in ONE launch of csrc/targets.hip (``oadg_roi_targets``) instead of ≈ 75 gathers, concatenations
        and the tensor form of the encoder.  Returns (rois [K + K_extra, 5], K, targets) or None when the inputs are
        outside the kernel's domain (callers then take the tensor path).  bbox_head.py:190-257,328-394,
        transforms.py:75-94."""
        from . import _lib
        n_t, n_all = len(sampling_results), len(sampling_results) + len(extra)
        if not self.FUSED_TARGETS or n_t == 0 or n_all > _lib.ROI_TARGET_MAX_ENTRIES:
            return None
        first = sampling_results[0]
        on_device = isinstance(first, DeviceSamplingResult)      # the positive / negative split lives on the device
        if not on_device and not first.pos_inds.is_cuda:
            return None
        if on_device and not all(isinstance(r, DeviceSamplingResult) for r in sampling_results):
            return None
        dev = first.sel.device if on_device else first.pos_inds.device
        entries = (_lib.RoiTargetEntry * n_all)()
        keep, K = [], 0
        n_src = 0
        if on_device:
            # the views of a batch share their sampling results (the list repeats them): n_src distinct images whose
            # count rows are consecutive in one tensor
            ids = []
            for r in sampling_results:
                if id(r) not in ids:
                    ids.append(id(r))
            n_src = len(ids)
            base_cnt = first.cnt
            if any(r is not sampling_results[i % n_src] or
                   r.cnt.data_ptr() != base_cnt.data_ptr() + (i % n_src) * 8 for i, r in enumerate(sampling_results)):
                return None
        for i, r in enumerate(sampling_results):
            bboxes, gtb, ar, _ = r._src
            if ar.labels is None or bboxes.dtype != torch.float32 or bboxes.dim() != 2 or bboxes.stride(1) != 1:
                return None
            gtb = gtb.view(-1, 4) if gtb.dim() < 2 else gtb
            if gtb.dtype != torch.float32 or not gtb.is_contiguous():
                gtb = gtb.float().contiguous()
            gi, lab = ar.gt_inds, ar.labels
            if gi.dtype != torch.long or lab.dtype != torch.long:
                return None
            gi, lab = gi.contiguous(), lab.contiguous()
            e = entries[i]
            e.bboxes, e.gt_bboxes, e.gt_inds, e.labels = bboxes.data_ptr(), gtb.data_ptr(), gi.data_ptr(), lab.data_ptr()
            if on_device:       # npos = the row capacity; the kernel reads the split from the count rows
                keep.extend((gtb, gi, lab))
                e.pos_inds, e.neg_inds, e.npos, e.nneg = r.sel.data_ptr(), None, r.cap, 0
                K += r.cap
            else:
                pi, ni = r.pos_inds.contiguous(), r.neg_inds.contiguous()
                keep.extend((gtb, gi, lab, pi, ni))
                e.pos_inds, e.neg_inds, e.npos, e.nneg = pi.data_ptr(), ni.data_ptr(), pi.numel(), ni.numel()
                K += pi.numel() + ni.numel()
            e.stride, e.batch = bboxes.stride(0), i
        K_all = K
        for j, b in enumerate(extra):
            if b.dtype != torch.float32 or b.dim() != 2 or b.size(1) < 4 or b.stride(1) != 1 or b.device != dev:
                return None
            e = entries[n_t + j]
            e.bboxes, e.npos, e.nneg, e.stride, e.batch = b.data_ptr(), 0, b.size(0), b.stride(0), j
            K_all += b.size(0)
        rois = torch.empty((K_all, 5), dtype=torch.float32, device=dev)
        labels = torch.empty((K,), dtype=torch.long, device=dev)
        label_weights = torch.empty((K,), dtype=torch.float32, device=dev)
        bbox_targets = torch.empty((K, 4), dtype=torch.float32, device=dev)
        bbox_weights = torch.empty((K, 4), dtype=torch.float32, device=dev)
        absolute = torch.empty((K, 4), dtype=torch.float32, device=dev)
        means = (ctypes.c_float * 4)(*[float(v) for v in self.bbox_coder.means])
        stds = (ctypes.c_float * 4)(*[float(v) for v in self.bbox_coder.stds])
        if on_device:
            _lib.check(_lib.lib().oadg_roi_targets_dev(
                ctypes.cast(entries, ctypes.c_void_p), n_all, n_t, n_src, _lib.ptr(first.cnt), int(self.num_classes),
                float(rcnn_train_cfg.pos_weight), ctypes.cast(means, ctypes.c_void_p), ctypes.cast(stds, ctypes.c_void_p),
                _lib.ptr(rois), _lib.ptr(labels), _lib.ptr(label_weights), _lib.ptr(bbox_targets), _lib.ptr(bbox_weights),
                _lib.ptr(absolute), _lib.stream_ptr()), 'oadg_roi_targets_dev')
        else:
            _lib.check(_lib.lib().oadg_roi_targets(
                ctypes.cast(entries, ctypes.c_void_p), n_all, n_t, int(self.num_classes), float(rcnn_train_cfg.pos_weight),
                ctypes.cast(means, ctypes.c_void_p), ctypes.cast(stds, ctypes.c_void_p), _lib.ptr(rois), _lib.ptr(labels),
                _lib.ptr(label_weights), _lib.ptr(bbox_targets), _lib.ptr(bbox_weights), _lib.ptr(absolute),
                _lib.stream_ptr()), 'oadg_roi_targets')
        del keep
        return rois, K, (labels, label_weights, bbox_targets, bbox_weights, absolute)

    def get_targets(self, sampling_results, gt_bboxes, gt_labels, rcnn_train_cfg, concat=True):
        return self.get_targets_with_absolute(sampling_results, gt_bboxes, gt_labels, rcnn_train_cfg,
                                              concat)[:4]

    # -- loss ------------------------------------------------------------------------------------------
    def get_bboxes(self, rois, cls_score, bbox_pred, img_shape, scale_factor, rescale=False, cfg=None):
        """bbox_head.py:468-530: softmax scores, decode per class, undo the test-time resize, multiclass NMS."""
        from .core import multiclass_nms
        scores = F.softmax(cls_score.float(), dim=-1) if cls_score is not None else None
        if bbox_pred is not None:
            bboxes = self.bbox_coder.decode(rois[..., 1:], bbox_pred.float(), max_shape=img_shape)
        else:
            bboxes = rois[:, 1:].clone()
            if img_shape is not None:
                bboxes[:, [0, 2]] = bboxes[:, [0, 2]].clamp(min=0, max=img_shape[1])
                bboxes[:, [1, 3]] = bboxes[:, [1, 3]].clamp(min=0, max=img_shape[0])
        if rescale and bboxes.size(0) > 0:
            sf = torch.as_tensor(np.asarray(scale_factor, dtype=np.float32)).to(bboxes.device)
            bboxes = (bboxes.view(bboxes.size(0), -1, 4) / sf).view(bboxes.size(0), -1)
        if cfg is None:
            return bboxes, scores
        return multiclass_nms(bboxes, scores, cfg.score_thr, cfg.nms, cfg.max_per_img)

    def _cls_reg_losses(self, cls_score, bbox_pred, labels, label_weights, bbox_targets, bbox_weights,
                        avg_factor, reduction_override=None, pos_rows=None, view1_rows=None):
        """``cls_score`` / ``bbox_pred`` as the head produced them (bf16 under autocast): the classification loss and the
        tensor path cast to fp32 like the reference's ``loss`` does, the fused box loss reads them as they are."""
        losses = dict()
        cls_f = cls_score.float() if cls_score is not None else None
        if cls_f is not None and cls_f.numel() > 0:
            losses['loss_cls'] = self.loss_cls(cls_f, labels, label_weights, avg_factor=avg_factor,
                                               reduction_override=reduction_override)
        fused = self._fused_reg_acc(cls_score, bbox_pred, labels, bbox_targets, bbox_weights, reduction_override, pos_rows,
                                    view1_rows)
        if fused is not None:
            losses['acc'], losses['loss_bbox'] = fused[1], fused[0]         # (the reference's key order: loss_cls, acc, loss_bbox)
            return losses
        if cls_f is not None and cls_f.numel() > 0:
            losses['acc'] = accuracy(cls_f, labels)
        bbox_pred = bbox_pred.float() if bbox_pred is not None else None
        if bbox_pred is not None and pos_rows is not None:
            # positives are the leading rows of every image block: their indices are known on the host
            if pos_rows.numel() > 0:
                from .core.bbox import _pinned_to
                pr = _pinned_to(pos_rows, bbox_pred.device)
                if self.reg_class_agnostic:
                    pos_pred = bbox_pred.view(bbox_pred.size(0), 4)[pr]
                else:
                    pos_pred = bbox_pred.view(bbox_pred.size(0), -1, 4)[pr, labels[pr]]
                losses['loss_bbox'] = self.loss_bbox(pos_pred, bbox_targets[pr], bbox_weights[pr],
                                                     avg_factor=bbox_targets.size(0),
                                                     reduction_override=reduction_override)
            else:
                losses['loss_bbox'] = bbox_pred[:0].sum()
        elif bbox_pred is not None:
            bg = self.num_classes
            pos = (labels >= 0) & (labels < bg)
            if pos.any():
                if self.reg_class_agnostic:
                    pos_pred = bbox_pred.view(bbox_pred.size(0), 4)[pos]
                else:
                    pos_pred = bbox_pred.view(bbox_pred.size(0), -1, 4)[pos, labels[pos]]
                losses['loss_bbox'] = self.loss_bbox(pos_pred, bbox_targets[pos], bbox_weights[pos],
                                                     avg_factor=bbox_targets.size(0),
                                                     reduction_override=reduction_override)
            else:
                losses['loss_bbox'] = bbox_pred[pos].sum()
        return losses

    def _fused_reg_acc(self, cls_score, bbox_pred, labels, bbox_targets, bbox_weights, reduction_override, pos_rows,
                       view1_rows=None):
        """(loss_bbox, acc) from one launch of csrc/cls_loss.hip (hip_ops.roi_reg_acc) - or None: the tensor path runs.
        ``pos_rows`` (host tensor, ascending) tells which rows are positives without a device read; the fork's
        ...LossPlus takes the leading chunk of them (view 1, smooth_l1_loss_plus.py: torch.chunk(pos_pred, num_views)[0]).
        ``view1_rows`` (device-side sampling, the split is not on the host): the rows of view 1 - the views share their
        sampling, so the leading chunk of the positives = the positives among those rows; every other row below that bound
        is a negative, whose background label the kernel skips: the same rows take part."""
        from .losses import L1Loss, L1LossPlus, SmoothL1Loss, SmoothL1LossPlus
        lb = self.loss_bbox
        if pos_rows is None and view1_rows is not None:
            pos_rows = ()
        if not (hip_ops.FUSED_ROI_LOSS and pos_rows is not None and reduction_override is None and cls_score is not None
                and bbox_pred is not None and bbox_pred.is_cuda and cls_score.numel() > 0 and bbox_pred.dim() == 2
                and type(lb) in (L1Loss, L1LossPlus, SmoothL1Loss, SmoothL1LossPlus) and lb.reduction == 'mean'
                and bbox_pred.dtype in (torch.float32, torch.bfloat16) and cls_score.dtype in (torch.float32, torch.bfloat16)
                and labels.dtype == torch.long and bbox_pred.shape[1] == (4 if self.reg_class_agnostic else 4 * self.num_classes)):
            return None
        K = bbox_pred.shape[0]
        if view1_rows is not None and isinstance(pos_rows, tuple):
            reg_limit = int(view1_rows) if isinstance(lb, (L1LossPlus, SmoothL1LossPlus)) else K
        elif isinstance(lb, (L1LossPlus, SmoothL1LossPlus)):
            P = int(pos_rows.numel())
            n1 = -(-P // lb.num_views)                          # rows of torch.chunk(pos_pred, num_views)[0]
            reg_limit = int(pos_rows[n1 - 1]) + 1 if n1 > 0 else 0
        else:
            reg_limit = K
        beta = float(getattr(lb, 'beta', 0.0))
        return hip_ops.roi_reg_acc(bbox_pred, cls_score, labels, bbox_targets, bbox_weights, self.num_classes, reg_limit,
                                   beta, float(max(bbox_targets.size(0), 1)), float(lb.loss_weight))

    def loss(self, cls_score, bbox_pred, rois, labels, label_weights, bbox_targets, bbox_weights,
             reduction_override=None, num_sampled=None, pos_rows=None, view1_rows=None, **kwargs):
        """bbox_head.py:397-...: avg_factor = #(label_weights > 0)."""
        avg = None
        if cls_score is not None:
            avg = max(float(num_sampled), 1.) if num_sampled is not None else \
                max(torch.sum(label_weights > 0).float().item(), 1.)
        return self._cls_reg_losses(cls_score, bbox_pred, labels, label_weights, bbox_targets, bbox_weights, avg,
                                    reduction_override, pos_rows=pos_rows, view1_rows=view1_rows)


class ConvFCBBoxHead(BBoxHead):
    """convfc_bbox_head.py restricted to fc branches (num_*_convs == 0 in every named config)."""

    def __init__(self, num_shared_convs=0, num_shared_fcs=0, num_cls_convs=0, num_cls_fcs=0,
                 num_reg_convs=0, num_reg_fcs=0, conv_out_channels=256, fc_out_channels=1024, conv_cfg=None,
                 norm_cfg=None, init_cfg=None, *args, **kwargs):
        super().__init__(*args, init_cfg=init_cfg, **kwargs)
        assert num_shared_convs == num_cls_convs == num_reg_convs == 0
        assert num_shared_fcs > 0
        self.num_shared_fcs, self.num_cls_fcs, self.num_reg_fcs = num_shared_fcs, num_cls_fcs, num_reg_fcs
        self.fc_out_channels = fc_out_channels
        last = self.in_channels * self.roi_feat_area
        self.shared_fcs = nn.ModuleList()
        for i in range(num_shared_fcs):
            self.shared_fcs.append(nn.Linear(last if i == 0 else fc_out_channels, fc_out_channels))
        self.shared_out_channels = fc_out_channels
        self.cls_fcs, self.reg_fcs = nn.ModuleList(), nn.ModuleList()
        for i in range(num_cls_fcs):
            self.cls_fcs.append(nn.Linear(fc_out_channels, fc_out_channels))
        for i in range(num_reg_fcs):
            self.reg_fcs.append(nn.Linear(fc_out_channels, fc_out_channels))
        self.cls_last_dim = self.reg_last_dim = fc_out_channels
        self.relu = nn.ReLU(inplace=True)
        self._make_predictors(self.cls_last_dim, self.reg_last_dim)

    def init_weights(self):
        """bbox_head.py:83-94 + convfc init_cfg: fc_cls N(0,.01), fc_reg N(0,.001), shared/cls/reg fcs Xavier."""
        if self.with_cls:
            normal_init(self.fc_cls, std=0.01)
        if self.with_reg:
            normal_init(self.fc_reg, std=0.001)
        for ml in (self.shared_fcs, self.cls_fcs, self.reg_fcs):
            for m in ml:
                xavier_init(m, distribution='uniform')

    _cast = {}

    def _cast_params(self, x, skip_weight_of=None):
        """{parameter: bf16 copy} for every Linear of the head, made in one multi-tensor pass (hip_ops.cast_all_bf16) when
        the head runs under bf16 autocast on the device - else {} and every layer casts for itself as autocast does"""
        self._cast = {}
        if not (hip_ops.FC_CAST_ONCE and x.is_cuda and torch.is_autocast_enabled() and torch.is_grad_enabled()
                and _autocast_dtype() == torch.bfloat16):
            return
        ps = []
        for m in self.modules():
            if isinstance(m, nn.Linear):
                for t in (m.weight, m.bias):
                    if t is not None and t is not skip_weight_of and t.dtype == torch.float32 and t.is_cuda:
                        ps.append(t)
        if ps:
            self._cast = dict(zip(ps, hip_ops.cast_all_bf16(ps)))

    def _lin(self, fc, x, relu=False):
        """``[relu](fc(x))`` on the pre-cast parameters when :meth:`_cast_params` made them (hip_ops.linear_bias_grad: the
        bias gradient / ReLU mask of the backward pass on the library's kernel)"""
        c = self._cast
        w = c.get(fc.weight) if c else None
        if w is None:
            return self.relu(fc(x)) if relu else fc(x)
        return hip_ops.linear_bias_grad(x, w, c.get(fc.bias) if fc.bias is not None else None, relu)

    def _seq(self, seq, x):
        mods = list(seq)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.Linear):
                fuse = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                x = self._lin(m, x, relu=fuse)
                i += 2 if fuse else 1
            else:
                x = m(x)
                i += 1
        return x

    def _trunk(self, x):
        fcs = list(self.shared_fcs)
        if hip_ops.FC_PERMUTE and x.dim() == 4 and x.is_cuda and x.dtype == torch.bfloat16 and \
                torch.is_autocast_enabled() and x.shape[1] % 64 == 0 and x.shape[2] * x.shape[3] <= 256 and \
                x.is_contiguous(memory_format=torch.channels_last) and fcs[0].weight.dtype == torch.float32 and \
                fcs[0].weight.is_contiguous():
            # RoIAlign wrote [K][ph][pw][C]: the features stay where they are (a view) and the first FC's weight columns
            # are permuted to that order inside autocast's cast pass (hip_ops._FcWeightPermute) - the same products as
            # the reference's NCHW flatten, summed in another order
            fc = fcs.pop(0)
            K, C, PH, PW = x.shape
            self._cast_params(x, skip_weight_of=fc.weight)
            w = hip_ops.fc_weight_permuted(fc.weight, C, PH * PW)
            x = hip_ops.linear_bias_grad(x.permute(0, 2, 3, 1).reshape(K, PH * PW * C), w, self._cast.get(fc.bias, fc.bias), True)
        else:
            self._cast_params(x)
            x = x.flatten(1)        # (c, ph, pw) order, as the reference's NCHW flatten
        for fc in fcs:
            x = self._lin(fc, x, relu=True)
        x_cls, x_reg = x, x
        for fc in self.cls_fcs:
            x_cls = self._lin(fc, x_cls, relu=True)
        for fc in self.reg_fcs:
            x_reg = self._lin(fc, x_reg, relu=True)
        return x, x_cls, x_reg

    def forward(self, x):
        _, x_cls, x_reg = self._trunk(x)
        out = (self._lin(self.fc_cls, x_cls) if self.with_cls else None,
               self._lin(self.fc_reg, x_reg) if self.with_reg else None)
        self._cast = {}
        return out


@HEADS.register_module()
class Shared2FCBBoxHead(ConvFCBBoxHead):

    def __init__(self, fc_out_channels=1024, *args, **kwargs):
        super().__init__(num_shared_fcs=2, fc_out_channels=fc_out_channels, *args, **kwargs)
        self.init_weights()


@HEADS.register_module()
class Shared2FCContrastiveHead(ConvFCBBoxHead):
    """contrastive_head.py:14-366: shared 2 FC + cls / reg / contrastive branches."""

    def __init__(self, fc_out_channels=1024, num_cls_convs=0, num_cls_fcs=0, with_cont=True,
                 cont_predictor_cfg=dict(num_linear=2, feat_channels=256, return_relu=True),
                 out_dim_cont=1024,
                 loss_cont=dict(type='ContrastiveLossPlus', version='r-cnn', loss_weight=0.01),
                 *args, **kwargs):
        super().__init__(num_shared_fcs=2, num_cls_convs=num_cls_convs, num_cls_fcs=num_cls_fcs,
                         fc_out_channels=fc_out_channels, *args, **kwargs)
        self.with_cont, self.out_dim_cont = with_cont, out_dim_cont
        self.loss_cont = build_loss(loss_cont)
        self.loss_cont.num_classes = self.num_classes
        if with_cont:
            self.fc_cont = self._linear_relu(in_channels=self.cls_last_dim, **cont_predictor_cfg)
        self.init_weights()

    @staticmethod
    def _linear_relu(num_linear, in_channels, feat_channels, return_relu=False):
        """contrastive_head.py:252-263 (note: ``return_relu`` only shifts where the ReLUs stop)."""
        layers = []
        num_relu = num_linear if return_relu else num_linear - 1
        for i in range(num_linear):
            layers.append(nn.Linear(in_channels if i == 0 else feat_channels, feat_channels))
            if i < num_relu - 1:
                layers.append(nn.ReLU(inplace=True))
        return nn.Sequential(*layers)

    def forward(self, x):
        x, x_cls, x_reg = self._trunk(x)
        self.cls_feats = x_cls
        out = (self._lin(self.fc_cls, x_cls) if self.with_cls else None,
               self._lin(self.fc_reg, x_reg) if self.with_reg else None,
               self._seq(self.fc_cont, x) if self.with_cont else None)
        self._cast = {}
        return out

    def loss(self, cls_score, bbox_pred, cont_feats, rois, labels, label_weights, bbox_targets, bbox_weights,
             bbox_absolute_targets=None, reduction_override=None, num_sampled=None, pos_rows=None, view1_rows=None,
             **kwargs):
        """contrastive_head.py:60-138.  ``num_sampled`` / ``pos_rows`` (host-side facts about the sampling
        results) replace the reference's two device reads: every sampled row has label weight 1."""
        avg = None
        if cls_score is not None:
            avg = max(float(num_sampled), 1.) if num_sampled is not None else \
                max(torch.sum(label_weights > 0).float().item(), 1.)
        losses = self._cls_reg_losses(cls_score, bbox_pred, labels, label_weights, bbox_targets, bbox_weights, avg,
                                      reduction_override, pos_rows=pos_rows, view1_rows=view1_rows)
        labels = labels.contiguous().view(-1, 1)
        if cont_feats is not None and cont_feats.numel() > 0:
            # The reference adds the key only when #foreground > min_samples (a host-side branch on device
            # data, :123-129); the kernel applies the same rule on the device and returns 0 otherwise, so
            # the key is always present and every rank logs the same set of variables.
            losses['loss_cont'] = self.loss_cont(cont_feats.float(), labels)
        self.roi_targets = (labels, label_weights, bbox_targets, bbox_weights)
        return losses


class BaseRoIHead(nn.Module):

    def __init__(self, bbox_roi_extractor=None, bbox_head=None, mask_roi_extractor=None, mask_head=None,
                 shared_head=None, train_cfg=None, test_cfg=None, pretrained=None, init_cfg=None, **kwargs):
        super().__init__()
        assert mask_head is None and shared_head is None, 'bbox-only detector'
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.bbox_roi_extractor = build_roi_extractor(bbox_roi_extractor)
        self.bbox_head = build_head(bbox_head)
        self.bbox_assigner = self.bbox_sampler = None
        if self.train_cfg:
            self.bbox_assigner = build_assigner(self.train_cfg.assigner)
            self.bbox_sampler = build_sampler(self.train_cfg.sampler, context=self)

    with_bbox, with_mask, with_shared_head = True, False, False


@HEADS.register_module()
class StandardRoIHead(BaseRoIHead):
    """standard_roi_head.py:11-200 (bbox branch)."""

    def _assign_and_sample(self, x, n, proposal_list, gt_bboxes, gt_labels, gt_bboxes_ignore, defer=False):
        """Per image: MaxIoU assign + RandomSampler (standard_roi_head.py:88-101).  Proposal lists may be
        padded to a fixed length with score -1 rows (RPNHead.get_bboxes(padded=True)); those rows are never
        candidates.  All images share one host read of the candidate counts."""
        if all(g is None for g in gt_bboxes_ignore[:n]) and hasattr(self.bbox_sampler, 'random_choice') and \
                hasattr(self.bbox_assigner, 'assign_masked'):
            props = proposal_list[:n]
            if all(l is not None for l in gt_labels[:n]):
                pend = roi_assign_sample_begin(self.bbox_assigner, self.bbox_sampler, props, gt_bboxes[:n], gt_labels[:n])
                if pend is not None:          # assignment + gts added as proposals: three launches for the batch
                    return pend if defer else pend.finish()
            valids = [p[:, 4] >= 0 if p.size(1) == 5 else None for p in props]
            out = self.bbox_assigner.assign_many(props, valids, gt_bboxes[:n], gt_labels[:n]) \
                if hasattr(self.bbox_assigner, 'assign_many') else None
            if out is not None:          # one fused assignment for the batch (csrc/assign.hip)
                ars, counts = out
            else:
                ars, counts = [], None
                for i in range(n):
                    p = props[i]
                    valid = valids[i] if valids[i] is not None else torch.ones_like(p[:, 0], dtype=torch.bool)
                    ars.append(self.bbox_assigner.assign_masked(p[:, :4], valid, gt_bboxes[i], gt_labels[i]))
            pend = sample_many_begin(self.bbox_sampler, ars, props, gt_bboxes[:n], gt_labels[:n], counts=counts)
            return pend if defer else pend.finish()
        out = []
        for i in range(n):
            ar = self.bbox_assigner.assign(proposal_list[i], gt_bboxes[i], gt_bboxes_ignore[i], gt_labels[i])
            out.append(self.bbox_sampler.sample(ar, proposal_list[i], gt_bboxes[i], gt_labels[i]))
        return out

    @staticmethod
    def _host_counts(sampling_results):
        """(#sampled rows, flat row indices of the positives) from shapes only - no device read."""
        if sampling_results and isinstance(sampling_results[0], DeviceSamplingResult):
            # device-side sampling: every image carries the sampler's full `num` rows (verified by the trainer before the
            # optimizer step); which of them are positives stays on the device
            return sum(r.cap for r in sampling_results), None
        total, rows = 0, []
        for r in sampling_results:
            npos, nneg = r.pos_inds.numel(), r.neg_inds.numel()
            rows.append(torch.arange(total, total + npos))
            total += npos + nneg
        return total, (torch.cat(rows) if rows else torch.zeros(0, dtype=torch.long))

    @staticmethod
    def _view1_rows(sampling_results, pos_rows, kwargs):
        """rows of the first view when the sampling results keep their positive / negative split on the device"""
        if pos_rows is not None or not sampling_results or not isinstance(sampling_results[0], DeviceSamplingResult):
            return None
        return sum(r.cap for r in sampling_results) // int(kwargs.get('num_views', 1))

    def begin_sampling(self, proposal_list, gt_bboxes, gt_labels, num_imgs, **kwargs):
        """Enqueue assignment + the asynchronous candidate-count read for the images that will be sampled; the
        detector calls this right after the NMS and passes the handle back as ``pending_sampling``."""
        n = kwargs['batch_size'] if 'num_views' in kwargs else num_imgs
        return self._assign_and_sample(None, n, proposal_list, gt_bboxes, gt_labels, [None] * num_imgs, defer=True)

    def forward_train(self, x, img_metas, proposal_list, gt_bboxes, gt_labels, gt_bboxes_ignore=None,
                      gt_masks=None, pending_sampling=None, **kwargs):
        num_imgs = len(img_metas)
        if gt_bboxes_ignore is None:
            gt_bboxes_ignore = [None] * num_imgs
        if pending_sampling is not None and hasattr(pending_sampling, 'finish'):
            with _rf('sec:roi_sampling_finish'):
                first = pending_sampling.finish()
            sampling_results = list(first)
            if 'num_views' in kwargs:
                sampling_results = []
                for _ in range(kwargs['num_views']):
                    sampling_results.extend(first)
        elif 'num_views' not in kwargs:
            sampling_results = self._assign_and_sample(x, num_imgs, proposal_list, gt_bboxes, gt_labels,
                                                       gt_bboxes_ignore)
        else:   # assign/sample on the first view only, replicate the lists (contrastive_roi_head.py:84-97)
            first = self._assign_and_sample(x, kwargs['batch_size'], proposal_list, gt_bboxes, gt_labels,
                                            gt_bboxes_ignore)
            sampling_results = []
            for _ in range(kwargs['num_views']):
                sampling_results.extend(first)
        bbox_results = self._bbox_forward_train(x, sampling_results, gt_bboxes, gt_labels, img_metas, **kwargs)
        return dict(bbox_results['loss_bbox'])

    @torch.no_grad()
    def simple_test_bboxes(self, x, img_metas, proposals, rcnn_test_cfg, rescale=False):
        """test_mixins.py:51-137"""
        rois = bbox2roi(proposals)
        if rois.shape[0] == 0:
            det_bbox, det_label = rois.new_zeros(0, 5), rois.new_zeros((0,), dtype=torch.long)
            if rcnn_test_cfg is None:
                det_bbox, det_label = det_bbox[:, :4], rois.new_zeros((0, self.bbox_head.fc_cls.out_features))
            return [det_bbox] * len(proposals), [det_label] * len(proposals)
        bbox_results = self._bbox_forward(x, rois)
        self.bbox_results = bbox_results
        num_per_img = tuple(len(p) for p in proposals)
        rois = rois.split(num_per_img, 0)
        cls_score = bbox_results['cls_score'].split(num_per_img, 0)
        bbox_pred = bbox_results['bbox_pred']
        bbox_pred = bbox_pred.split(num_per_img, 0) if bbox_pred is not None else (None,) * len(proposals)
        det_bboxes, det_labels = [], []
        for i in range(len(proposals)):
            if rois[i].shape[0] == 0:
                det_bbox, det_label = rois[i].new_zeros(0, 5), rois[i].new_zeros((0,), dtype=torch.long)
                if rcnn_test_cfg is None:
                    det_bbox = det_bbox[:, :4]
                    det_label = rois[i].new_zeros((0, self.bbox_head.fc_cls.out_features))
            else:
                det_bbox, det_label = self.bbox_head.get_bboxes(rois[i], cls_score[i], bbox_pred[i],
                                                                img_metas[i]['img_shape'], img_metas[i]['scale_factor'],
                                                                rescale=rescale, cfg=rcnn_test_cfg)
            det_bboxes.append(det_bbox)
            det_labels.append(det_label)
        return det_bboxes, det_labels

    @torch.no_grad()
    def aug_test_bboxes(self, feats, img_metas, proposal_list, rcnn_test_cfg):
        """test_mixins.py:139-177: the merged proposals (original frame) mapped into every augmentation, boxes / scores of
        the augmentations mapped back and averaged, one multiclass NMS"""
        from .core import bbox_mapping, merge_aug_bboxes, multiclass_nms
        aug_bboxes, aug_scores = [], []
        for x, meta in zip(feats, img_metas):
            m0 = meta[0]                                            # only one image in the batch
            proposals = bbox_mapping(proposal_list[0][:, :4], m0['img_shape'], m0['scale_factor'], m0['flip'],
                                     m0.get('flip_direction') or 'horizontal')
            rois = bbox2roi([proposals])
            res = self._bbox_forward(x, rois)
            bboxes, scores = self.bbox_head.get_bboxes(rois, res['cls_score'], res['bbox_pred'], m0['img_shape'],
                                                       m0['scale_factor'], rescale=False, cfg=None)
            aug_bboxes.append(bboxes)
            aug_scores.append(scores)
        merged_bboxes, merged_scores = merge_aug_bboxes(aug_bboxes, aug_scores, img_metas, rcnn_test_cfg)
        if merged_bboxes.shape[0] == 0:
            return merged_bboxes.new_zeros(0, 5), merged_bboxes.new_zeros((0,), dtype=torch.long)
        return multiclass_nms(merged_bboxes, merged_scores, rcnn_test_cfg.score_thr, rcnn_test_cfg.nms,
                              rcnn_test_cfg.max_per_img)

    def aug_test(self, x, proposal_list, img_metas, rescale=False):
        """standard_roi_head.py:438-462 (no mask branch): if ``rescale`` is False the boxes fit the scale of imgs[0]"""
        from .core import bbox2result
        det_bboxes, det_labels = self.aug_test_bboxes(x, img_metas, proposal_list, self.test_cfg)
        if not rescale:
            det_bboxes = det_bboxes.clone()
            det_bboxes[:, :4] *= det_bboxes.new_tensor(img_metas[0][0]['scale_factor'])
        return [bbox2result(det_bboxes, det_labels, self.bbox_head.num_classes)]

    def simple_test(self, x, proposal_list, img_metas, proposals=None, rescale=False):
        """standard_roi_head.py:392-436 (no mask branch): per image a list over classes of [k, 5] numpy arrays."""
        from .core import bbox2result
        det_bboxes, det_labels = self.simple_test_bboxes(x, img_metas, proposal_list, self.test_cfg, rescale=rescale)
        return [bbox2result(det_bboxes[i], det_labels[i], self.bbox_head.num_classes) for i in range(len(det_bboxes))]

    def _bbox_forward(self, x, rois):
        feats = self.bbox_roi_extractor(x[:self.bbox_roi_extractor.num_inputs], rois)
        cls_score, bbox_pred = self.bbox_head(feats)
        return dict(cls_score=cls_score, bbox_pred=bbox_pred, bbox_feats=feats)

    def _bbox_forward_train(self, x, sampling_results, gt_bboxes, gt_labels, img_metas, **kwargs):
        fused = self.bbox_head.rois_and_targets(sampling_results, self.train_cfg)
        if fused is not None:
            rois, _, targets = fused
            targets = targets[:4]
            res = self._bbox_forward(x, rois)
        else:
            rois = bbox2roi([r.bboxes for r in sampling_results])
            res = self._bbox_forward(x, rois)
            targets = self.bbox_head.get_targets(sampling_results, gt_bboxes, gt_labels, self.train_cfg)
        num_sampled, pos_rows = self._host_counts(sampling_results)
        res.update(loss_bbox=self.bbox_head.loss(res['cls_score'], res['bbox_pred'], rois, *targets,
                                                 num_sampled=num_sampled, pos_rows=pos_rows,
                                                 view1_rows=self._view1_rows(sampling_results, pos_rows, kwargs)))
        return res


@HEADS.register_module()
class ContrastiveRoIHead(StandardRoIHead):
    """contrastive_roi_head.py:10-157."""

    def _bbox_forward(self, x, rois):
        feats = self.bbox_roi_extractor(x[:self.bbox_roi_extractor.num_inputs], rois)
        cls_score, bbox_pred, cont_feats = self.bbox_head(feats)
        return dict(cls_score=cls_score, bbox_pred=bbox_pred, cont_feats=cont_feats, bbox_feats=feats)

    def _bbox_forward_train(self, x, sampling_results, gt_bboxes, gt_labels, img_metas,
                            gt_instance_inds=None, **kwargs):
        with _rf('sec:roi_bbox_forward'):
            extra = [r[:, :4] for r in kwargs['random_proposal_list']] if 'random_proposal_list' in kwargs else []
            fused = self.bbox_head.rois_and_targets(sampling_results, self.train_cfg, extra)
            if fused is not None:
                rois_all, K, targets = fused
                rois = rois_all[:K]
            else:
                rois = bbox2roi([r.bboxes for r in sampling_results])
                K, targets = rois.shape[0], None
                rois_all = torch.cat([rois, bbox2roi(extra)], dim=0) if extra else rois
            self._last_rois = [rois.detach()]
            if extra:
                # the reference runs extractor + head a second time on the random proposals and keeps only their
                # contrastive features (contrastive_roi_head.py:131-137); one pass over both RoI sets is the same
                # function row for row and halves the RoIAlign backward's full-pyramid gradient traffic (one fp32
                # scatter buffer, one cast, no accumulation add per level)
                both = self._bbox_forward(x, rois_all)
                res = dict(cls_score=both['cls_score'][:K], bbox_pred=both['bbox_pred'][:K],
                           cont_feats=both['cont_feats'], bbox_feats=both['bbox_feats'][:K])
                self._last_rois.append(rois_all[K:].detach())
            else:
                res = self._bbox_forward(x, rois)
        with _rf('sec:roi_targets'):
            if targets is None:
                targets = self.bbox_head.get_targets_with_absolute(sampling_results, gt_bboxes, gt_labels,
                                                                   self.train_cfg)
            self.bbox_targets = targets
            num_sampled, pos_rows = self._host_counts(sampling_results)
        with _rf('sec:roi_loss'):
            res.update(loss_bbox=self.bbox_head.loss(res['cls_score'], res['bbox_pred'], res['cont_feats'], rois,
                                                     *targets, num_sampled=num_sampled, pos_rows=pos_rows,
                                                     view1_rows=self._view1_rows(sampling_results, pos_rows, kwargs),
                                                     **kwargs))
        return res
