"""RPN head (mmdet/models/dense_heads/{base_dense_head,anchor_head,rpn_head}.py).

Host orchestration is torch tensor plumbing; the classification loss is the fused HIP CE+JSD kernel and the
proposal NMS is the batched HIP NMS (one launch for all images instead of one mmcv call per image)."""
import copy
import os

import numpy as np

import torch
from torch.profiler import record_function as _rf
import torch.nn as nn
import torch.nn.functional as F

from . import hip_ops
from .core import anchor_inside_flags, images_to_levels, multi_apply, unmap
from .core.bbox import sample_many, sample_many_begin
from .layers import Conv2d, conv2d, normal_init
from .registry import (HEADS, build_assigner, build_bbox_coder, build_loss, build_prior_generator,
                       build_sampler)


class AnchorHead(nn.Module):

    def __init__(self, num_classes, in_channels, feat_channels=256,
                 anchor_generator=dict(type='AnchorGenerator', scales=[8, 16, 32], ratios=[0.5, 1.0, 2.0],
                                       strides=[4, 8, 16, 32, 64]),
                 bbox_coder=dict(type='DeltaXYWHBBoxCoder', clip_border=True,
                                 target_means=(.0, .0, .0, .0), target_stds=(1.0, 1.0, 1.0, 1.0)),
                 reg_decoded_bbox=False,
                 loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                 loss_bbox=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0),
                 train_cfg=None, test_cfg=None, init_cfg=None):
        super().__init__()
        self.in_channels, self.num_classes, self.feat_channels = in_channels, num_classes, feat_channels
        self.use_sigmoid_cls = loss_cls.get('use_sigmoid', False)
        self.cls_out_channels = num_classes if self.use_sigmoid_cls else num_classes + 1
        if self.cls_out_channels <= 0:
            raise ValueError(f'num_classes={num_classes} is too small')
        self.reg_decoded_bbox = reg_decoded_bbox
        self.bbox_coder = build_bbox_coder(bbox_coder)
        self.loss_cls = build_loss(loss_cls)
        self.loss_bbox = build_loss(loss_bbox)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        if self.train_cfg:
            self.assigner = build_assigner(self.train_cfg.assigner)
            # anchor_head.py:96-101: sampling unless the cls loss is focal-style
            self.sampling = loss_cls['type'] not in ['FocalLoss', 'GHMC', 'QualityFocalLoss']
            sampler_cfg = self.train_cfg.sampler if self.sampling and hasattr(self.train_cfg, 'sampler') \
                else dict(type='PseudoSampler')
            self.sampler = build_sampler(sampler_cfg, context=self)
        self.prior_generator = build_prior_generator(anchor_generator)
        self.num_base_priors = self.prior_generator.num_base_priors[0]
        self._init_layers()

    # -- targets -------------------------------------------------------------------------------------
    def get_anchors(self, featmap_sizes, img_metas, device='cuda'):
        """anchor_head.py:171-199."""
        multi_level_anchors = self.prior_generator.grid_priors(featmap_sizes, device=device)
        anchor_list = [multi_level_anchors for _ in range(len(img_metas))]
        # valid_flags only depend on shapes: when the padded image covers every feature-map cell, every anchor
        # is valid and the boolean filtering of the reference (a device->host synchronisation) is a no-op
        self._all_anchors_valid = all(
            min(int(-(-m['pad_shape'][0] // s[1])), fs[0]) == fs[0] and
            min(int(-(-m['pad_shape'][1] // s[0])), fs[1]) == fs[1]
            for m in img_metas for fs, s in zip(featmap_sizes, self.prior_generator.strides))
        valid_flag_list = [self.prior_generator.valid_flags(featmap_sizes, m['pad_shape'], device)
                           for m in img_metas]
        return anchor_list, valid_flag_list

    def _get_targets_single(self, flat_anchors, valid_flags, gt_bboxes, gt_bboxes_ignore, gt_labels,
                            img_meta, sampling_result=None, label_channels=1, unmap_outputs=True):
        """anchor_head.py:201-297."""
        if sampling_result is None:     # reference order of operations: filter, assign, sample (synchronising)
            inside_flags = anchor_inside_flags(flat_anchors, valid_flags, img_meta['img_shape'][:2],
                                               self.train_cfg.allowed_border)
            if not inside_flags.any():
                return (None,) * 7
            anchors = flat_anchors[inside_flags, :]
            assign_result = self.assigner.assign(anchors, gt_bboxes, gt_bboxes_ignore,
                                                 None if self.sampling else gt_labels)
            sampling_result = self.sampler.sample(assign_result, anchors, gt_bboxes)
        else:                           # all anchors valid, already assigned + sampled for the whole batch
            inside_flags, anchors = None, flat_anchors
        n = anchors.shape[0]
        bbox_targets = torch.zeros_like(anchors)
        bbox_weights = torch.zeros_like(anchors)
        labels = anchors.new_full((n,), self.num_classes, dtype=torch.long)
        label_weights = anchors.new_zeros(n, dtype=torch.float)
        pos_inds, neg_inds = sampling_result.pos_inds, sampling_result.neg_inds
        if len(pos_inds) > 0:
            if not self.reg_decoded_bbox:
                pos_t = self.bbox_coder.encode(sampling_result.pos_bboxes, sampling_result.pos_gt_bboxes)
            else:
                pos_t = sampling_result.pos_gt_bboxes
            # index_fill_/index_copy_, not ``x[inds] = scalar``: the latter stages the scalar through a blocking
            # pageable host->device copy (6 ms of host stall per statement behind a busy stream, measured)
            bbox_targets.index_copy_(0, pos_inds, pos_t)
            bbox_weights.index_fill_(0, pos_inds, 1.0)
            if gt_labels is None:
                labels.index_fill_(0, pos_inds, 0)   # RPN: foreground is class 0
            else:
                labels.index_copy_(0, pos_inds, gt_labels[sampling_result.pos_assigned_gt_inds])
            label_weights.index_fill_(0, pos_inds,
                                      1.0 if self.train_cfg.pos_weight <= 0 else self.train_cfg.pos_weight)
        if len(neg_inds) > 0:
            label_weights.index_fill_(0, neg_inds, 1.0)
        if unmap_outputs and inside_flags is not None:
            total = flat_anchors.size(0)
            labels = unmap(labels, total, inside_flags, fill=self.num_classes)
            label_weights = unmap(label_weights, total, inside_flags)
            bbox_targets = unmap(bbox_targets, total, inside_flags)
            bbox_weights = unmap(bbox_weights, total, inside_flags)
        return labels, label_weights, bbox_targets, bbox_weights, pos_inds, neg_inds, sampling_result

    def get_targets(self, anchor_list, valid_flag_list, gt_bboxes_list, img_metas,
                    gt_bboxes_ignore_list=None, gt_labels_list=None, label_channels=1, unmap_outputs=True):
        """anchor_head.py:299-400."""
        num_imgs = len(img_metas)
        assert len(anchor_list) == len(valid_flag_list) == num_imgs
        num_level_anchors = [a.size(0) for a in anchor_list[0]]
        if gt_bboxes_ignore_list is None:
            gt_bboxes_ignore_list = [None] * num_imgs
        if gt_labels_list is None:
            gt_labels_list = [None] * num_imgs
        fast = (all(g is None for g in gt_bboxes_ignore_list) and hasattr(self.sampler, 'random_choice') and
                hasattr(self.assigner, 'assign_masked') and not getattr(self, 'reference_order_targets', False))
        pending = getattr(self, '_pending_targets', None)
        self._pending_targets = None
        if fast and pending is not None and pending[0] == (num_imgs, tuple(num_level_anchors)):
            srs = pending[1].finish()       # assigned at the start of the step (begin_targets)
            fused = self._fused_targets(pending[1], srs, num_level_anchors, gt_labels_list, unmap_outputs)
            if fused is not None:
                return fused
        # per-image concatenation over the levels; the images of a batch share their (cached) anchor / flag lists
        memo = {}

        def cat_once(lst):
            if id(lst) not in memo:
                memo[id(lst)] = torch.cat(lst)
            return memo[id(lst)]
        concat_anchors = [cat_once(a) for a in anchor_list]
        concat_flags = [cat_once(f) for f in valid_flag_list]
        if fast and pending is not None and pending[0] == (num_imgs, tuple(num_level_anchors)):
            pass
        elif fast:   # one host read for the whole batch instead of ~6 per image
            ars = [self._assign_inside(concat_anchors[i], concat_flags[i], gt_bboxes_list[i], img_metas[i],
                                       None if self.sampling else gt_labels_list[i]) for i in range(num_imgs)]
            srs = sample_many(self.sampler, ars, concat_anchors, gt_bboxes_list)
        else:
            srs = [None] * num_imgs
        results = multi_apply(self._get_targets_single, concat_anchors, concat_flags, gt_bboxes_list,
                              gt_bboxes_ignore_list, gt_labels_list, img_metas, srs,
                              label_channels=label_channels, unmap_outputs=unmap_outputs)
        all_labels, all_label_weights, all_bbox_targets, all_bbox_weights, pos_l, neg_l, _ = results[:7]
        if any(l is None for l in all_labels):
            return None
        num_total_pos = sum(max(i.numel(), 1) for i in pos_l)
        num_total_neg = sum(max(i.numel(), 1) for i in neg_l)
        return (images_to_levels(all_labels, num_level_anchors),
                images_to_levels(all_label_weights, num_level_anchors),
                images_to_levels(all_bbox_targets, num_level_anchors),
                images_to_levels(all_bbox_weights, num_level_anchors), num_total_pos, num_total_neg)

    def _fused_targets(self, pend, srs, num_level_anchors, gt_labels_list, unmap_outputs):
        """anchor_head.py:201-297 for the whole batch in two launches (csrc/targets.hip oadg_anchor_targets) when
        the assignment and the selection were done by the batch kernels over shared, all-inside anchors."""
        from . import _lib
        sel = getattr(pend, 'device_select', None)
        batch = getattr(pend.prepared[0][0], 'batch', None) if pend.prepared else None
        if sel is None or batch is None or batch['boxes'] is None or not unmap_outputs or self.reg_decoded_bbox or \
                not (self._all_anchors_valid and self.train_cfg.allowed_border < 0) or \
                not (gt_labels_list is None or all(l is None for l in gt_labels_list)):
            return None
        anchors, gt_inds = batch['boxes'], batch['gt_inds']
        B, A = gt_inds.shape
        dev = anchors.device
        labels = torch.empty((B, A), dtype=torch.long, device=dev)
        label_weights = torch.empty((B, A), dtype=torch.float32, device=dev)
        bbox_targets = torch.empty((B, A, 4), dtype=torch.float32, device=dev)
        bbox_weights = torch.empty((B, A, 4), dtype=torch.float32, device=dev)
        import ctypes
        f4 = ctypes.c_float * 4
        means, stds = f4(*self.bbox_coder.means), f4(*self.bbox_coder.stds)
        L = _lib.lib()
        _lib.check(L.oadg_anchor_targets(_lib.ptr(anchors), _lib.ptr(batch['gts']), _lib.ptr(gt_inds), None,
                                         _lib.ptr(sel['jobs_dev']), _lib.ptr(sel['sel']), B, A, batch['Gmax'],
                                         sel['max_k'], self.num_classes, float(self.train_cfg.pos_weight),
                                         ctypes.cast(means, ctypes.c_void_p), ctypes.cast(stds, ctypes.c_void_p),
                                         _lib.ptr(labels), _lib.ptr(label_weights), _lib.ptr(bbox_targets),
                                         _lib.ptr(bbox_weights), _lib.stream_ptr()), 'oadg_anchor_targets')
        num_total_pos = sum(max(r.pos_inds.numel(), 1) for r in srs)
        num_total_neg = sum(max(r.neg_inds.numel(), 1) for r in srs)
        self._whole_targets = (labels, label_weights, bbox_targets, bbox_weights)      # for the fused loss (all levels)
        out, start = [[], [], [], []], 0
        for n in num_level_anchors:              # images_to_levels: [B, n_level(, 4)] per level
            for lst, t in zip(out, (labels, label_weights, bbox_targets, bbox_weights)):
                lst.append(t[:, start:start + n])
            start += n
        return out[0], out[1], out[2], out[3], num_total_pos, num_total_neg

    def _assign_inside(self, flat_anchors, valid_flags, gt_bboxes, img_meta, gt_labels=None):
        """Assignment over the anchors the reference keeps (anchor_inside_flags, anchor/utils.py:19-43) without
        compacting them: anchors outside get gt_ind -1, so they are neither candidates nor part of any maximum,
        and the targets come out already 'unmapped' (anchor_head.py:285-295)."""
        if self._all_anchors_valid and self.train_cfg.allowed_border < 0:
            return self.assigner.assign(flat_anchors, gt_bboxes, None, gt_labels)
        inside = anchor_inside_flags(flat_anchors, valid_flags, img_meta['img_shape'][:2],
                                     self.train_cfg.allowed_border)
        return self.assigner.assign_masked(flat_anchors, inside, gt_bboxes, gt_labels)

    def _assign_batch(self, flat, flag_list, gt_bboxes, img_metas, gt_labels=None):
        """_assign_inside for all images: one fused assignment (csrc/assign.hip) when the assigner offers it."""
        n = len(img_metas)
        if hasattr(self.assigner, 'assign_many'):
            if self._all_anchors_valid and self.train_cfg.allowed_border < 0:
                valids = None
            else:
                valids = [anchor_inside_flags(flat, torch.cat(flag_list[i]), img_metas[i]['img_shape'][:2],
                                              self.train_cfg.allowed_border) for i in range(n)]
            out = self.assigner.assign_many(flat, valids, gt_bboxes, None if self.sampling else gt_labels)
            if out is not None:
                return out
        return [self._assign_inside(flat, torch.cat(flag_list[i]), gt_bboxes[i], img_metas[i],
                                    None if (self.sampling or gt_labels is None) else gt_labels[i])
                for i in range(n)], None

    def begin_targets(self, pad_hw, gt_bboxes, img_metas, device):
        """Anchor targets depend on anchors and gts only, not on the network: enqueue the IoU / assignment of all
        images and the asynchronous read of the sampler's candidate counts BEFORE the backbone runs, so the
        read has long completed when :meth:`loss` needs it (the reference computes them inside loss(),
        anchor_head.py:455-544, and stalls the device there)."""
        import math
        sizes = [(math.ceil(pad_hw[0] / s[1]), math.ceil(pad_hw[1] / s[0])) for s in self.prior_generator.strides]
        anchor_list, flag_list = self.get_anchors(sizes, img_metas, device=device)
        if not (hasattr(self.sampler, 'random_choice') and hasattr(self.assigner, 'assign_masked')):
            return
        flat = torch.cat(anchor_list[0])
        ars, counts = self._assign_batch(flat, flag_list, gt_bboxes, img_metas)
        pend = sample_many_begin(self.sampler, ars, [flat] * len(img_metas), gt_bboxes, counts=counts)
        self._pending_targets = ((len(img_metas), tuple(a.size(0) for a in anchor_list[0])), pend)

    # -- loss ----------------------------------------------------------------------------------------
    def loss_single(self, cls_score, bbox_pred, labels, label_weights, bbox_targets, bbox_weights,
                    num_total_samples):
        """anchor_head.py:402-452; NHWC-contiguous maps make the permute+reshape a free view."""
        labels = labels.reshape(-1)
        label_weights = label_weights.reshape(-1)
        cls_score = cls_score.float().permute(0, 2, 3, 1).reshape(-1, self.cls_out_channels)
        loss_cls = self.loss_cls(cls_score, labels, label_weights, avg_factor=num_total_samples)
        bbox_targets = bbox_targets.reshape(-1, 4)
        bbox_weights = bbox_weights.reshape(-1, 4)
        bbox_pred = bbox_pred.float().permute(0, 2, 3, 1).reshape(-1, 4)
        loss_bbox = self.loss_bbox(bbox_pred, bbox_targets, bbox_weights, avg_factor=num_total_samples)
        return loss_cls, loss_bbox

    FUSED_LOSS = os.environ.get('OADG_FUSED_RPN_LOSS', '1') == '1'

    def _fused_loss(self, cls_scores, bbox_preds, num_total_samples):
        """anchor_head.py:402-452,530-544 for all levels in one launch each way (hip_ops.rpn_loss) when the head wrote its
        fused channel-padded maps, the targets came from the batch kernel and the losses are the OA-DG pair (sigmoid
        CrossEntropyLossPlus with / without JSD, L1LossPlus, 2 views); None otherwise."""
        from .losses import CrossEntropyLossPlus, L1LossPlus
        whole = getattr(self, '_whole_targets', None)
        ys = [getattr(c, '_oadg_y', None) for c in cls_scores]
        lc, lb = self.loss_cls, self.loss_bbox
        if not self.FUSED_LOSS or whole is None or any(y is None for y in ys) or not ys[0].is_cuda or \
                type(lc) is not CrossEntropyLossPlus or type(lb) is not L1LossPlus or not lc.use_sigmoid or \
                lc.num_views != 2 or lb.num_views != 2 or lb.reduction != 'mean' or lc.reduction != 'mean' or \
                self.cls_out_channels != 1 or self.reg_decoded_bbox or not torch.is_grad_enabled():
            return None
        A = cls_scores[0].shape[1]
        B = ys[0].shape[0]
        if whole[0].shape[0] != B or (B & 1) or whole[0].shape[1] != sum(c.shape[2] * c.shape[3] * A for c in cls_scores) or \
                any(y.shape[1] < 5 * A or (y.shape[1] & 7) or not y.is_contiguous(memory_format=torch.channels_last)
                    for y in ys):
            return None
        lam = lc.lambda_weight if lc.with_jsd else 0.0
        loss_cls, loss_bbox, parts = hip_ops.rpn_loss(ys, A, whole, float(num_total_samples), lc.loss_weight, lam,
                                                       lb.loss_weight)
        lc.wandb_features[f'ce_loss({lc.wandb_name})'] = parts[1]
        lc.wandb_features[f'lam_additional_loss({lc.wandb_name})'] = parts[2]
        return dict(loss_cls=[loss_cls], loss_bbox=[loss_bbox])

    def loss(self, cls_scores, bbox_preds, gt_bboxes, gt_labels, img_metas, gt_bboxes_ignore=None):
        """anchor_head.py:455-544."""
        featmap_sizes = [f.size()[-2:] for f in cls_scores]
        assert len(featmap_sizes) == self.prior_generator.num_levels
        device = cls_scores[0].device
        anchor_list, valid_flag_list = self.get_anchors(featmap_sizes, img_metas, device=device)
        label_channels = self.cls_out_channels if self.use_sigmoid_cls else 1
        self._whole_targets = None
        with _rf('sec:rpn_get_targets'):
            targets = self.get_targets(anchor_list, valid_flag_list, gt_bboxes, img_metas,
                                       gt_bboxes_ignore_list=gt_bboxes_ignore, gt_labels_list=gt_labels,
                                       label_channels=label_channels)
        if targets is None:
            return None
        self.rpn_targets = targets
        labels_l, lw_l, bt_l, bw_l, num_pos, num_neg = targets
        num_total_samples = num_pos + num_neg if self.sampling else num_pos
        fused = self._fused_loss(cls_scores, bbox_preds, num_total_samples)
        if fused is not None:
            return fused
        losses_cls, losses_bbox = multi_apply(self.loss_single, cls_scores, bbox_preds, labels_l, lw_l, bt_l,
                                              bw_l, num_total_samples=num_total_samples)
        return dict(loss_cls=losses_cls, loss_bbox=losses_bbox)

    def forward(self, feats):
        return multi_apply(self.forward_single, feats)

    def forward_train(self, x, img_metas, gt_bboxes, gt_labels=None, gt_bboxes_ignore=None,
                      proposal_cfg=None, num_proposal_imgs=None, padded_proposals=False, after_proposals=None,
                      **kwargs):
        """base_dense_head.py:302-342.  ``num_proposal_imgs`` limits proposal generation to the first images
        (the contrastive RoI head only consumes the view-1 lists, contrastive_roi_head.py:85-95)."""
        with _rf('sec:rpn_head_convs'):
            outs = self(x)
        # proposals are enqueued before the loss so that the RoI head's host read (candidate counts) only
        # waits for the NMS, while the device is still busy with the RPN loss kernels
        proposal_list = None
        if proposal_cfg is not None:
            with _rf('sec:rpn_get_bboxes'):
                proposal_list = self.get_bboxes(*outs, img_metas=img_metas, cfg=proposal_cfg,
                                                num_imgs=num_proposal_imgs, padded=padded_proposals)
        if after_proposals is not None:
            after_proposals(proposal_list)      # e.g. the RoI head's assignment + asynchronous count read
        with _rf('sec:rpn_loss'):
            losses = self.loss(*outs, gt_bboxes, gt_labels, img_metas, gt_bboxes_ignore=gt_bboxes_ignore)
        if proposal_cfg is None:
            return losses
        return losses, proposal_list


class _SplitHeads(torch.autograd.Function):
    """(y[:, :a], y[:, a:a+b]) of a channel-padded NHWC map; the backward writes the two gradients into one zeroed
    padded map (3 launches instead of autograd's two zero-fills, two copies and an add)."""

    @staticmethod
    def forward(ctx, y, a, b):
        ctx.meta = (tuple(y.shape), y.dtype, a, b)
        return y[:, :a], y[:, a:a + b]

    @staticmethod
    def backward(ctx, ga, gb):
        shape, dt, a, b = ctx.meta
        g = torch.empty(shape, dtype=dt, device=(ga if ga is not None else gb).device,
                        memory_format=torch.channels_last).zero_()
        if ga is not None:
            g[:, :a].copy_(ga)
        if gb is not None:
            g[:, a:a + b].copy_(gb)
        return g, None, None


@HEADS.register_module()
class RPNHead(AnchorHead):
    """rpn_head.py:17-235."""

    def __init__(self, in_channels, init_cfg=dict(type='Normal', layer='Conv2d', std=0.01), num_convs=1,
                 **kwargs):
        assert num_convs == 1, 'the named configs use a single 3x3 RPN conv'
        self.num_convs = num_convs
        super().__init__(1, in_channels, init_cfg=init_cfg, **kwargs)

    def _init_layers(self):
        self.rpn_conv = Conv2d(self.in_channels, self.feat_channels, 3, padding=1)
        self.rpn_cls = Conv2d(self.feat_channels, self.num_base_priors * self.cls_out_channels, 1)
        self.rpn_reg = Conv2d(self.feat_channels, self.num_base_priors * 4, 1)
        self.init_weights()

    def init_weights(self):
        for m in (self.rpn_conv, self.rpn_cls, self.rpn_reg):
            normal_init(m, std=0.01)

    def forward_single(self, x):
        c = self.rpn_conv
        from . import hip_conv
        n_cls, n_reg = self.rpn_cls.out_channels, self.rpn_reg.out_channels
        if hip_conv.ENABLED and x.is_cuda and (x.dtype == torch.bfloat16 or torch.is_autocast_enabled()) and \
                n_cls + n_reg <= 128 and self.feat_channels % 64 == 0 and torch.is_grad_enabled():
            # rpn_cls and rpn_reg (3 + 12 output channels, rpn_head.py:56-59) as ONE 1x1 convolution on the MFMA
            # kernel, output channels zero-padded to its 128-channel tile: one pass over the 256-channel feature map
            # instead of two library convolutions forward and four backward, and - through the GradToken - the ReLU
            # mask and the bias gradient of rpn_conv come out of that convolution's data-gradient epilogue.
            tok = hip_conv.GradToken()
            # (in_token: an FPN level whose gradient this convolution finishes - necks.FPN._fpn_conv)
            x = conv2d(x, c.weight, c.bias, c.stride, c.padding, c.dilation, relu=True, out_token=tok,
                       in_token=getattr(x, '_oadg_token', None), owner=c)
            if getattr(x.grad_fn, 'name', lambda: '')().startswith('_Conv2dMFMA'):
                if hip_conv.NARROW_HEAD and n_cls + n_reg <= 16 and self.feat_channels in (128, 256) and \
                        self.rpn_cls.bias is not None and self.rpn_reg.bias is not None:
                    # round 4: 16-channel-wide head maps (csrc/narrow_head.hip) instead of a 128-channel tile with 15 live
                    # channels - the head's output and its gradient are written and re-read by five passes per step
                    y = hip_conv.narrow_head(x, *self._narrow_head_params(), in_token=tok)
                else:
                    w, b = self._fused_head_params()
                    y = conv2d(x, w, b, 1, 0, 1, in_token=tok)
                cls, reg = _SplitHeads.apply(y, n_cls, n_reg)
                cls._oadg_y = y          # (AnchorHead._fused_loss reads the head's map in place)
                return cls, reg
            return self.rpn_cls(x), self.rpn_reg(x)
        x = conv2d(x, c.weight, c.bias, c.stride, c.padding, c.dilation, relu=True)   # relu(rpn_conv(x))
        return self.rpn_cls(x), self.rpn_reg(x)

    def _fused_head_params(self):
        """[rpn_cls; rpn_reg; zero rows] weight / bias of the fused 1x1 head, built once per forward pass (the
        pyramid levels share it: one concatenation and one gradient split per step instead of five)."""
        c = getattr(self, '_fused_cache', None)
        if c is not None:
            return c
        wc, wr = self.rpn_cls.weight, self.rpn_reg.weight
        pad = 128 - wc.shape[0] - wr.shape[0]
        z = getattr(self, '_fused_pad', None)
        if z is None or z[0].device != wc.device or z[0].dtype != wc.dtype:
            z = self._fused_pad = (wc.new_zeros((pad,) + tuple(wc.shape[1:])), self.rpn_cls.bias.new_zeros(pad))
        c = (torch.cat([wc, wr, z[0]]), torch.cat([self.rpn_cls.bias, self.rpn_reg.bias, z[1]]))
        if torch.is_grad_enabled():
            self._fused_cache = c
        return c

    def _narrow_head_params(self):
        """[rpn_cls; rpn_reg] weight / bias and their 16-row bf16 forms for csrc/narrow_head.hip, built once per forward pass"""
        c = getattr(self, '_narrow_cache', None)
        if c is not None:
            return c
        from . import hip_conv
        w = torch.cat([self.rpn_cls.weight, self.rpn_reg.weight])
        b = torch.cat([self.rpn_cls.bias, self.rpn_reg.bias])
        c = (w, b) + hip_conv.narrow_params(w, b)
        if torch.is_grad_enabled():
            self._narrow_cache = c
        return c

    def _narrow_ok(self, x):
        from . import hip_conv
        n = self.rpn_cls.out_channels + self.rpn_reg.out_channels
        return hip_conv.NARROW_HEAD and hip_conv.ENABLED and x.is_cuda and n <= 16 and self.feat_channels in (128, 256) and \
            (x.dtype == torch.bfloat16 or torch.is_autocast_enabled()) and torch.is_grad_enabled() and \
            self.rpn_cls.bias is not None and self.rpn_reg.bias is not None

    def _forward_narrow(self, feats):
        """all pyramid levels through the 16-channel head as ONE autograd node (hip_conv._NarrowHead): rpn_conv per level,
        then the shared rpn_cls + rpn_reg on every level whose rpn_conv ran on the MFMA kernels"""
        from . import hip_conv
        c = self.rpn_conv
        n_cls, n_reg = self.rpn_cls.out_channels, self.rpn_reg.out_channels
        hs, toks = [], []
        for x in feats:
            tok = hip_conv.GradToken()
            h = conv2d(x, c.weight, c.bias, c.stride, c.padding, c.dilation, relu=True, out_token=tok,
                       in_token=getattr(x, '_oadg_token', None), owner=c)
            hs.append(h)
            toks.append(tok if getattr(h.grad_fn, 'name', lambda: '')().startswith('_Conv2dMFMA') else None)
        sel = [i for i, t in enumerate(toks) if t is not None]
        ys = dict(zip(sel, hip_conv.narrow_head_levels([hs[i] for i in sel], *self._narrow_head_params(),
                                                        tokens=[toks[i] for i in sel]))) if sel else {}
        cls_scores, bbox_preds = [], []
        for i, h in enumerate(hs):
            if i in ys:
                cls, reg = _SplitHeads.apply(ys[i], n_cls, n_reg)
                cls._oadg_y = ys[i]      # (AnchorHead._fused_loss reads the head's map in place)
            else:
                cls, reg = self.rpn_cls(h), self.rpn_reg(h)
            cls_scores.append(cls)
            bbox_preds.append(reg)
        return cls_scores, bbox_preds

    def forward(self, feats):
        self._fused_cache = self._narrow_cache = None
        try:
            if len(feats) and self._narrow_ok(feats[0]):
                return self._forward_narrow(feats)
            return super().forward(feats)
        finally:
            self._fused_cache = self._narrow_cache = None

    def loss(self, cls_scores, bbox_preds, gt_bboxes, gt_labels, img_metas, gt_bboxes_ignore=None):
        losses = super().loss(cls_scores, bbox_preds, gt_bboxes, None, img_metas,
                              gt_bboxes_ignore=gt_bboxes_ignore)
        return dict(loss_rpn_cls=losses['loss_cls'], loss_rpn_bbox=losses['loss_bbox'])

    def simple_test_rpn(self, x, img_metas):
        """dense_test_mixins.py:118-133"""
        return self.get_bboxes(*self(x), img_metas=img_metas)

    FUSED_PROPOSALS = os.environ.get('OADG_FUSED_PROPOSALS', '1') == '1'
    FUSED_TOPK = os.environ.get('OADG_FUSED_TOPK', '1') == '1'

    def _fused_proposals(self, cls_scores, bbox_preds, mlvl_anchors, img_metas, cfg, n_img, nms_pre, padded):
        """The part of get_bboxes after the per-level top-k on csrc/proposals.hip (decode + clip + size test, merge of
        the sorted levels into the global descending order, class-offset boxes, final gather): 3 launches + the NMS pair
        instead of ~120 element-wise / gather / sort launches.  Bit-identical to the tensor path below
        (tests/test_hip_proposals.py).  None: a configuration the kernels do not cover (softmax scores, centre clamp,
        CPU tensors) - the tensor path runs."""
        import ctypes
        from . import _lib
        from .core.bbox import DeltaXYWHBBoxCoder, _pinned_to
        coder = self.bbox_coder
        dev = cls_scores[0].device
        if not (self.FUSED_PROPOSALS and dev.type == 'cuda' and self.use_sigmoid_cls and type(coder) is DeltaXYWHBBoxCoder
                and not coder.add_ctr_clamp and len(cls_scores) <= 8 and n_img >= 1):
            return None
        nms_cfg = dict(cfg.nms)
        if nms_cfg.pop('type', 'nms') != 'nms':
            return None
        thr = nms_cfg.get('iou_threshold', nms_cfg.get('iou_thr'))
        L = _lib.lib()
        nl = len(cls_scores)
        levels = (_lib.RpnLevel * nl)()
        keep_alive, first = [], 0
        if any(bp.dtype not in (torch.float32, torch.bfloat16) for bp in bbox_preds):
            return None
        dims = [(int(c.shape[2]), int(c.shape[3]), int(c.shape[1])) for c in cls_scores]
        ks = [min(nms_pre, h * w * a) if nms_pre > 0 else h * w * a for h, w, a in dims]
        topk = None
        if self.FUSED_TOPK and nms_pre > 0 and n_img * nl <= 48 and max(ks) <= 16384 and \
                len({c.dtype for c in cls_scores}) == 1 and cls_scores[0].dtype in (torch.float32, torch.bfloat16):
            # radix select + ordered compaction + one LDS sort per (image, level) row: six launches for all rows
            csd = [c.detach() for c in cls_scores]
            sc = [torch.empty((n_img, k), dtype=torch.float32, device=dev) for k in ks]
            ix = [torch.empty((n_img, k), dtype=torch.int64, device=dev) for k in ks]
            level_n = (ctypes.c_int * nl)(*[h * w * a for h, w, a in dims])
            nbytes = L.oadg_rpn_topk_workspace_bytes(level_n, nl, n_img, int(nms_pre))
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            strides = (ctypes.c_long * (4 * nl))(*[int(v) for c in csd for v in c.stride()])
            dims_c = (ctypes.c_int * (3 * nl))(*[v for d in dims for v in d])
            _lib.check(L.oadg_rpn_topk((ctypes.c_void_p * nl)(*[c.data_ptr() for c in csd]), strides, dims_c,
                                       0 if csd[0].dtype == torch.float32 else 1, nl, n_img, int(nms_pre),
                                       (ctypes.c_void_p * nl)(*[t.data_ptr() for t in sc]),
                                       (ctypes.c_void_p * nl)(*[t.data_ptr() for t in ix]), _lib.ptr(ws), nbytes,
                                       _lib.stream_ptr()), 'oadg_rpn_topk')
            topk = (sc, ix)
            keep_alive += [csd, ws]
        for lvl, (cs, bp) in enumerate(zip(cls_scores, bbox_preds)):
            H, W, A = dims[lvl]
            k = ks[lvl]
            if topk is not None:
                sc_k, idx_k = topk[0][lvl], topk[1][lvl]
            else:
                scores = cs[:n_img].detach().float().permute(0, 2, 3, 1).reshape(n_img, -1).sigmoid()
                # every level arrives in stable descending order (the merge of the kernel relies on it; the reference leaves
                # a level shorter than nms_pre unsorted, which only matters for the order of exactly tied invalid boxes)
                ranked, rank_inds = scores.sort(dim=1, descending=True, stable=True)
                sc_k, idx_k = ranked[:, :k].contiguous(), rank_inds[:, :k].contiguous()
            bpd = bp.detach()
            an = mlvl_anchors[lvl].contiguous()
            keep_alive += [sc_k, idx_k, bpd, an]
            lv = levels[lvl]
            lv.deltas, lv.anchors, lv.scores, lv.index = bpd.data_ptr(), an.data_ptr(), sc_k.data_ptr(), idx_k.data_ptr()
            lv.sN, lv.sC, lv.sH, lv.sW = (int(v) for v in bpd.stride())
            lv.H, lv.W, lv.A, lv.k, lv.dtype, lv.first = H, W, int(A), int(k), 0 if bp.dtype == torch.float32 else 1, first
            first += int(k)
        M = first
        if M == 0 or (2 * M + 1) * 4 > 150 * 1024:
            return None
        clip = bool(getattr(coder, 'clip_border', True))
        lim_host = torch.tensor([[m['img_shape'][1], m['img_shape'][0]] for m in img_metas[:n_img]], dtype=torch.float32)
        lim = _pinned_to(lim_host, dev)
        means = (ctypes.c_float * 4)(*[float(v) for v in coder.means])
        stds = (ctypes.c_float * 4)(*[float(v) for v in coder.stds])
        max_ratio = float(np.float32(np.abs(np.log(16 / 1000))))
        props = torch.empty((n_img, M, 4), dtype=torch.float32, device=dev)
        scores_cat = torch.empty((n_img, M), dtype=torch.float32, device=dev)
        valid = torch.empty((n_img, M), dtype=torch.uint8, device=dev)
        st = _lib.stream_ptr()
        _lib.check(L.oadg_rpn_decode(levels, len(cls_scores), n_img, means, stds, max_ratio, _lib.ptr(lim), int(clip),
                                     float(cfg.min_bbox_size), _lib.ptr(props), _lib.ptr(scores_cat), _lib.ptr(valid), st),
                   'oadg_rpn_decode')
        boxes_sorted = torch.empty((n_img, M, 4), dtype=torch.float32, device=dev)
        order = torch.empty((n_img, M), dtype=torch.int32, device=dev)
        counts = torch.empty((n_img,), dtype=torch.int32, device=dev)
        mx = torch.empty((n_img,), dtype=torch.float32, device=dev)
        _lib.check(L.oadg_rpn_order(levels, len(cls_scores), n_img, _lib.ptr(props), _lib.ptr(scores_cat), _lib.ptr(valid),
                                    _lib.ptr(boxes_sorted), _lib.ptr(order), _lib.ptr(counts), _lib.ptr(mx), st),
                   'oadg_rpn_order')
        keep, keep_cnt = hip_ops.nms_sorted_batched(boxes_sorted, counts, thr, cfg.max_per_img)
        P = min(cfg.max_per_img, M) if cfg.max_per_img > 0 else M
        dets = torch.empty((n_img, P, 5), dtype=torch.float32, device=dev)
        _lib.check(L.oadg_rpn_gather(n_img, M, P, _lib.ptr(props), _lib.ptr(scores_cat), _lib.ptr(order), _lib.ptr(keep),
                                     _lib.ptr(keep_cnt), _lib.ptr(dets), st), 'oadg_rpn_gather')
        del keep_alive
        if padded:
            return list(dets.unbind(0))
        cnt = keep_cnt.tolist()                      # one host read for the whole batch
        return [dets[i, :min(cnt[i], P)] for i in range(n_img)]

    @torch.no_grad()
    def aug_test_rpn(self, feats, img_metas):
        """dense_test_mixins.py:135-167: proposals of every test-time augmentation, merged per image in the ORIGINAL image
        frame (merge_aug_proposals)"""
        from .core import merge_aug_proposals
        n = len(img_metas[0])
        aug_proposals = [[] for _ in range(n)]
        for x, metas in zip(feats, img_metas):
            for i, proposals in enumerate(self.simple_test_rpn(x, metas)):
                aug_proposals[i].append(proposals)
        aug_metas = [[img_metas[j][i] for j in range(len(img_metas))] for i in range(n)]
        return [merge_aug_proposals(p, m, self.test_cfg) for p, m in zip(aug_proposals, aug_metas)]

    @torch.no_grad()
    def get_bboxes(self, cls_scores, bbox_preds, img_metas=None, cfg=None, num_imgs=None, padded=False,
                   **kwargs):
        """base_dense_head.py:31-106 + rpn_head.py:103-235, batched over images.

        Per image and level: stable descending sort, top nms_pre, decode against the anchors, drop
        w/h <= min_bbox_size, then class(level)-aware NMS and the first max_per_img survivors."""
        cfg = copy.deepcopy(self.test_cfg if cfg is None else cfg)
        assert len(cls_scores) == len(bbox_preds)
        n_img = cls_scores[0].shape[0] if num_imgs is None else min(num_imgs, cls_scores[0].shape[0])
        device = cls_scores[0].device
        featmap_sizes = [c.shape[-2:] for c in cls_scores]
        mlvl_anchors = self.prior_generator.grid_priors(featmap_sizes, device=device)
        nms_pre = cfg.get('nms_pre', -1)
        fused = self._fused_proposals(cls_scores, bbox_preds, mlvl_anchors, img_metas, cfg, n_img, nms_pre, padded)
        if fused is not None:
            return fused
        sc_l, dl_l, an_l, id_l = [], [], [], []
        for lvl, (cs, bp) in enumerate(zip(cls_scores, bbox_preds)):
            cs = cs[:n_img].detach().float().permute(0, 2, 3, 1)
            if self.use_sigmoid_cls:
                scores = cs.reshape(n_img, -1).sigmoid()
            else:
                scores = cs.reshape(n_img, -1, 2).softmax(dim=-1)[..., 0]
            deltas = bp[:n_img].detach().float().permute(0, 2, 3, 1).reshape(n_img, -1, 4)
            anchors = mlvl_anchors[lvl][None].expand(n_img, -1, -1)
            if 0 < nms_pre < scores.shape[1]:
                ranked, rank_inds = scores.sort(dim=1, descending=True, stable=True)
                top = rank_inds[:, :nms_pre]
                scores = ranked[:, :nms_pre]
                deltas = torch.gather(deltas, 1, top[..., None].expand(-1, -1, 4))
                anchors = torch.gather(anchors, 1, top[..., None].expand(-1, -1, 4))
            sc_l.append(scores)
            dl_l.append(deltas)
            an_l.append(anchors)
            id_l.append(scores.new_full((scores.shape[1],), lvl, dtype=torch.long))
        scores = torch.cat(sc_l, 1)                 # [I, M]
        deltas = torch.cat(dl_l, 1)
        anchors = torch.cat(an_l, 1)
        ids = torch.cat(id_l)[None].expand(n_img, -1)
        M = scores.shape[1]
        img_shape = img_metas[0]['img_shape']
        if all(tuple(m['img_shape'][:2]) == tuple(img_shape[:2]) for m in img_metas[:n_img]):
            props = self.bbox_coder.decode(anchors.reshape(-1, 4), deltas.reshape(-1, 4),
                                           max_shape=img_shape).view(n_img, M, 4)
        else:
            # per-sample image shapes inside one padded batch (multi-scale Resize, transforms.py:177-243): every image
            # clips its proposals to ITS img_shape (rpn_head.py:168-171 passes img_meta['img_shape'] per image)
            props = self.bbox_coder.decode(anchors.reshape(-1, 4), deltas.reshape(-1, 4),
                                           max_shape=None).view(n_img, M, 4)
            if getattr(self.bbox_coder, 'clip_border', True):
                lim = torch.tensor([[m['img_shape'][1], m['img_shape'][0]] for m in img_metas[:n_img]],
                                   dtype=props.dtype).pin_memory().to(device, non_blocking=True)      # [I, (W, H)]
                lim = lim.repeat(1, 2)[:, None, :]                                                   # x1 y1 x2 y2 limits
                props = torch.minimum(props.clamp(min=0), lim)
        valid = torch.ones_like(scores, dtype=torch.bool)
        if cfg.min_bbox_size >= 0:
            w = props[..., 2] - props[..., 0]
            h = props[..., 3] - props[..., 1]
            valid = (w > cfg.min_bbox_size) & (h > cfg.min_bbox_size)
        nms_cfg = dict(cfg.nms)
        assert nms_cfg.pop('type', 'nms') == 'nms'
        thr = nms_cfg.get('iou_threshold', nms_cfg.get('iou_thr'))
        # batched_nms (mmcv): offset every class by (max coordinate + 1); invalid boxes sort last
        mx = torch.where(valid[..., None], props, props.new_zeros(())).amax(dim=(1, 2))   # [I]
        offs = ids.to(props) * (mx + 1)[:, None]
        key = torch.where(valid, scores, scores.new_full((), -1.0))
        order = key.sort(dim=1, descending=True, stable=True)[1]
        boxes_sorted = torch.gather(props + offs[..., None], 1, order[..., None].expand(-1, -1, 4))
        counts = valid.sum(dim=1).int()
        keep, keep_cnt = hip_ops.nms_sorted_batched(boxes_sorted, counts, thr, cfg.max_per_img)
        P = min(cfg.max_per_img, M) if cfg.max_per_img > 0 else M
        if not padded:
            cnt = keep_cnt.tolist()                  # one host read for the whole batch
            out = []
            for i in range(n_img):
                sel = order[i, keep[i, :cnt[i]].long()]
                out.append(torch.cat([props[i, sel], scores[i, sel, None]], dim=1))
            return out
        # fixed-size lists without any host read: rows past the kept count are zero boxes with score -1
        kidx = keep[:, :P].long().clamp(0, M - 1)
        sel = torch.gather(order, 1, kidx)
        live = torch.arange(P, device=device)[None, :] < keep_cnt[:, None]
        pb = torch.gather(props, 1, sel[..., None].expand(-1, -1, 4))
        ps = torch.gather(scores, 1, sel)
        dets = torch.cat([torch.where(live[..., None], pb, pb.new_zeros(())),
                          torch.where(live, ps, ps.new_full((), -1.0))[..., None]], dim=2)
        return list(dets.unbind(0))
