"""Detectors (mmdet/models/detectors/{base,two_stage,faster_rcnn}.py)."""
import os
from collections import OrderedDict

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn
from torch.profiler import record_function as _rf

from .core import bbox_overlaps_np
from .core.bbox import _pinned_to
from .registry import DETECTORS, build_backbone, build_head, build_neck


def _adjacent_views(views):
    """the tensor the views are consecutive batch slices of (the device pipeline writes all views into one allocation
    and tags the first with it), or None: then the caller concatenates"""
    parent = getattr(views[0], '_oadg_batch', None)
    if parent is None or parent.shape[0] != sum(v.shape[0] for v in views):
        return None
    at = parent.data_ptr()
    for v in views:
        if v.data_ptr() != at or v.dtype != parent.dtype or v.shape[1:] != parent.shape[1:] or \
                v.stride() != parent.stride():
            return None
        at += v.shape[0] * v.stride(0) * v.element_size()
    return parent


_LOG_PAD = {}


def _log_pad(like, n_names, slots=31):
    """[zeros(max(0, slots - n_names)), float(n_names)] on ``like``'s device (cached constants: no fill launch per step)"""
    key = (like.device, like.dtype, n_names)
    t = _LOG_PAD.get(key)
    if t is None:
        t = _LOG_PAD[key] = torch.cat([torch.zeros(max(0, slots - n_names), dtype=like.dtype),
                                       torch.full((1,), float(n_names), dtype=like.dtype)]).to(like.device)
    return t


def integrate_data(data, train_cfg):
    """base.py:22-48: concatenate the views along the batch axis (all originals first, then all OA-Mix
    images) and extend/duplicate the per-image lists to match; inject ``num_views`` / ``batch_size``."""
    batch_size = len(data['img'])
    if 'inv' in train_cfg.keys():
        if train_cfg['inv']:
            data['img'] = torch.cat([data['img2'], data['img']], dim=0)
    else:
        views = [v for k, v in data.items() if ('img' in k) and ('img_metas' not in k)]
        data['img'] = views[0] if len(views) == 1 else _adjacent_views(views)
        if data['img'] is None:
            data['img'] = torch.cat(views, dim=0)
    num_views = int(len(data['img']) / batch_size)
    for i in range(2, num_views + 1):
        for key in ['img', 'gt_bboxes', 'gt_labels', 'gt_instance_inds', 'img_metas', 'multilevel_boxes',
                    'oamix_boxes']:
            if f'{key}{i}' in data:
                if key != 'img':
                    data[key] += data[f'{key}{i}']
                del data[f'{key}{i}']
            elif key in data:
                for b in range(batch_size):
                    data[key].append(data[key][b])
    data['num_views'] = num_views
    data['batch_size'] = batch_size
    return data


_NP_F32_COMPARE = not bool(np.float32(0.1) > 0.1)     # numpy >= 2 compares a float32 with a python float in float32
NATIVE_RANDOM_BBOXES = os.environ.get('OADG_NATIVE_RANDOM_BBOXES', '1') == '1'


def _random_bboxes_native(img_width, img_height, num_bboxes, bboxes_xy, scales, ratios, max_iters, iou_max, iou_min):
    """the trial loop below in csrc/host_rng.hip (oadg_np_random_bboxes), drawing IN PLACE from numpy's global generator:
    the same boxes and the same generator state afterwards (tests/test_random_bboxes_native.py), ~50x less interpreter
    time on the training thread.  None when the case is not covered (another bit generator, no gt boxes)."""
    from . import _lib
    rs = np.random.mtrand._rand
    bg = rs._bit_generator
    if type(bg).__name__ != 'MT19937' or (bboxes_xy is not None and len(bboxes_xy) == 0) or num_bboxes < 0:
        return None
    try:
        L = _lib.lib()
    except Exception:            # (the library is not built: CPU-only unit tests of the host logic)
        return None
    if bboxes_xy is None:
        gts, n_gt = None, -1
    else:
        gts = np.ascontiguousarray(np.asarray(bboxes_xy)[:, :4], dtype=np.float32)
        n_gt = gts.shape[0]
    thr = (lambda v: float(np.float32(v))) if _NP_F32_COMPARE else float
    out = np.zeros((num_bboxes, 5))
    with bg.lock:
        n = L.oadg_np_random_bboxes(bg.ctypes.state_address, int(img_width), int(img_height), int(num_bboxes),
                                    gts.ctypes.data if gts is not None else None, n_gt, float(scales[0]), float(scales[1]),
                                    float(ratios[0]), float(ratios[1]), int(max_iters), thr(iou_max), thr(iou_min),
                                    out.ctypes.data)
    if n < 0:
        raise RuntimeError(f'oadg_np_random_bboxes failed ({n})')
    return out[:n, :]


def generate_random_bboxes_xy(img_size, num_bboxes, bboxes_xy=None, scales=(0.01, 0.2), ratios=(0.3, 1 / 0.3),
                              max_iters=500, iou_max=1.0, iou_min=0.0, **kwargs):
    """two_stage.py:389-419 (global numpy RNG, same draw order: x1, y1, scale, ratio per trial)."""
    if isinstance(num_bboxes, (tuple, list)):
        num_bboxes = np.random.randint(num_bboxes[0], num_bboxes[1] + 1)
    img_width, img_height = img_size
    if NATIVE_RANDOM_BBOXES:
        out = _random_bboxes_native(img_width, img_height, num_bboxes, bboxes_xy, scales, ratios, max_iters, iou_max, iou_min)
        if out is not None:
            return out
    out = np.zeros((num_bboxes, 5))
    total = 0
    for _ in range(max_iters):
        if total >= num_bboxes:
            break
        x1, y1 = np.random.randint(0, img_width), np.random.randint(0, img_height)
        scale = np.random.uniform(*scales) * img_height * img_width
        ratio = np.random.uniform(*ratios)
        w, h = int(np.sqrt(scale / ratio)), int(np.sqrt(scale * ratio))
        box = np.array([[x1, y1, min(x1 + w, img_width), min(y1 + h, img_height), 1]])
        if bboxes_xy is not None:
            ious = bbox_overlaps_np(box, bboxes_xy)
            if np.max(ious) > iou_max or np.max(ious) < iou_min:
                continue
        out[total, :] = box[0]
        total += 1
    return out[:total, :]


class BaseDetector(nn.Module):

    def __init__(self, init_cfg=None):
        super().__init__()
        self.features, self.wandb_features = dict(), dict()

    def forward(self, img, img_metas, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(img, img_metas, **kwargs)
        return self.forward_test(img, img_metas, **kwargs)

    def forward_test(self, imgs, img_metas, **kwargs):
        """base.py:170-212: ``imgs`` / ``img_metas`` are lists with one entry per test-time augmentation."""
        if isinstance(imgs, torch.Tensor):
            imgs, img_metas = [imgs], [img_metas]
        if len(imgs) != len(img_metas):
            raise ValueError(f'num of augmentations ({len(imgs)}) != num of image meta ({len(img_metas)})')
        for img, metas in zip(imgs, img_metas):
            for m in metas:
                m['batch_input_shape'] = tuple(img.size()[-2:])
        if len(imgs) == 1:
            if 'proposals' in kwargs:
                kwargs['proposals'] = kwargs['proposals'][0]
            return self.simple_test(imgs[0], img_metas[0], **kwargs)
        # aug test: one image per batch only (base.py:203-212)
        assert imgs[0].size(0) == 1, f'aug test does not support inference with batch size {imgs[0].size(0)}'
        assert 'proposals' not in kwargs
        return self.aug_test(imgs, img_metas, **kwargs)

    def _parse_losses(self, losses):
        """base.py:234-277.  Same keys and values; the per-variable all-reduces + .item() of the reference
        become one packed all-reduce and one host read."""
        fused = self._parse_losses_fused(losses)
        if fused is not None:
            loss, packed, names = fused
        else:
            log_vars = OrderedDict()
            for name, value in losses.items():
                if isinstance(value, torch.Tensor):
                    log_vars[name] = value.mean()
                elif isinstance(value, list):
                    log_vars[name] = sum(v.mean() for v in value)
                else:
                    raise TypeError(f'{name} is not a tensor or list of tensors')
            loss = sum(v for k, v in log_vars.items() if 'loss' in k)
            log_vars['loss'] = loss
            names = list(log_vars.keys())
            packed = torch.stack([log_vars[k].detach().float().reshape(()) for k in names])
        distributed = dist.is_available() and dist.is_initialized() and not getattr(self, 'local_log_vars', False)
        if distributed:
            # base.py:258-265 checks that every rank logs the same variables; the count rides in the same
            # all-reduce and is verified without a per-step host synchronisation (see _check_log_count)
            # The vector has a FIXED length (values, zero padding to 31 slots, count): should the ranks ever disagree on
            # their keys, the collective still matches in size and the assertion below reports it - the reference
            # reduces the count in a collective of its own first for the same reason (base.py:258-265)
            packed = torch.cat([packed, _log_pad(packed, len(names))])
            dist.all_reduce(packed)
            expect = len(names) * dist.get_world_size()
            count, packed = packed[-1], packed[:len(names)] / dist.get_world_size()
        if getattr(self, 'log_vars_on_host', True):
            if distributed:
                vals = torch.cat([packed, count.view(1)]).tolist()          # one host read
                assert int(round(vals.pop())) == expect, \
                    f'loss log variables are different across GPUs! rank {dist.get_rank()} keys: {names}'
            else:
                vals = packed.tolist()
            log_vars = OrderedDict(zip(names, vals))
        else:   # benchmark mode: keep the packed device tensor, no host synchronisation per step
            if distributed:
                self._check_log_count(count, expect, names)
            log_vars = OrderedDict(zip(names, packed.unbind(0)))
        return loss, log_vars

    @staticmethod
    def _parse_losses_fused(losses):
        """(loss, packed log vector, names) from one launch (hip_ops.parse_losses) when every value is a one-element fp32
        CUDA tensor (a mean of one element is the element; the sums are python's left-to-right adds) - else None"""
        from . import hip_ops
        if not hip_ops.FUSED_PARSE_LOSSES or 'loss' in losses:
            return None
        vals, name_of, names, mask = [], [], [], 0
        for name, value in losses.items():
            items = [value] if isinstance(value, torch.Tensor) else value
            if not isinstance(items, list) or not items:
                return None
            for v in items:
                if not (isinstance(v, torch.Tensor) and v.is_cuda and v.dtype == torch.float32 and v.numel() == 1):
                    return None
                vals.append(v)
                name_of.append(len(names))
            if 'loss' in name:
                mask |= 1 << len(names)
            names.append(name)
        if not vals or len(vals) > 32 or len(names) > 30:
            return None
        loss, packed = hip_ops.parse_losses(vals, name_of, len(names), mask)
        return loss, packed, names + ['loss']

    def _check_log_count(self, count, expect, names):
        """Deferred form of the reference's cross-rank assertion: this step's all-reduced count is copied to a
        pinned buffer asynchronously and the PREVIOUS step's copy (long complete) is the one asserted on."""
        prev = getattr(self, '_log_count_pending', None)
        if prev is not None:
            buf, ev, exp, nm = prev
            ev.synchronize()
            assert int(round(buf.item())) == exp, \
                f'loss log variables are different across GPUs! rank {dist.get_rank()} keys: {nm}'
        if count.is_cuda:
            buf = torch.empty((), dtype=count.dtype, pin_memory=True)
            buf.copy_(count, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._log_count_pending = (buf, ev, expect, names)
        else:
            assert int(round(count.item())) == expect, \
                f'loss log variables are different across GPUs! rank {dist.get_rank()} keys: {names}'

    def train_step(self, data, optimizer):
        """base.py:413-455."""
        self.features.clear()
        self.wandb_features.clear()
        data = integrate_data(data, self.train_cfg)
        losses = self(**data)
        loss, log_vars = self._parse_losses(losses)
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data['img_metas']))


class TwoStageDetector(BaseDetector):
    """two_stage.py:13-204."""

    def __init__(self, backbone, neck=None, rpn_head=None, roi_head=None, train_cfg=None, test_cfg=None,
                 pretrained=None, init_cfg=None):
        super().__init__(init_cfg)
        self.backbone = build_backbone(backbone)
        self.neck = build_neck(neck) if neck is not None else None
        self.rpn_head = self.roi_head = None
        if rpn_head is not None:
            rpn_train_cfg = train_cfg.rpn if train_cfg is not None else None
            rpn_head_ = dict(rpn_head)
            rpn_head_.update(train_cfg=rpn_train_cfg, test_cfg=test_cfg.rpn if test_cfg is not None else None)
            self.rpn_head = build_head(rpn_head_)
        if roi_head is not None:
            roi_head_ = dict(roi_head)
            roi_head_.update(train_cfg=train_cfg.rcnn if train_cfg is not None else None,
                             test_cfg=test_cfg.rcnn if test_cfg is not None else None)
            self.roi_head = build_head(roi_head_)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg

    with_rpn = property(lambda self: self.rpn_head is not None)
    with_neck = property(lambda self: self.neck is not None)

    def init_weights(self, allow_missing_pretrained=None):
        """``allow_missing_pretrained=True``: a configured but unavailable backbone checkpoint (no network here) only
        warns and the backbone stays randomly initialised - the synthetic benchmark / smoke / test contract."""
        self.backbone.init_weights(allow_missing_pretrained)

    def extract_feat(self, img):
        x = self.backbone(img)
        if self.with_neck:
            x = self.neck(x)
        return x

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None, gt_masks=None,
                      proposals=None, **kwargs):
        # Host/device choreography: the two host reads of the step (sampler candidate counts
        # of the RPN and of the RoI head) are asynchronous copies followed by an event; each is waited for only
        # after more device work has been enqueued behind it, so the stream never drains.
        if self.with_rpn and gt_bboxes_ignore is None and hasattr(self.rpn_head, 'begin_targets'):
            with _rf('sec:rpn_begin_targets'):
                self.rpn_head.begin_targets(img.shape[2:], gt_bboxes, img_metas, img.device)
        # The random proposals (two_stage.py:162-204) depend on the gt / OA-Mix boxes only and draw from the numpy stream,
        # which nothing else in the step touches (the samplers use torch's CPU generator): they are generated HERE, while
        # the device still works on the previous step, instead of between the RPN and the RoI head where the stream has
        # drained to the sampler's host read (~2 ms of host-only numpy work in the most latency-sensitive window).
        random_proposals = None
        if 'random_proposal_cfg' in self.train_cfg.keys():
            host_gts = [m.get('gt_bboxes_np') for m in img_metas]
            if all(g is not None for g in host_gts):
                kwargs['img_metas_host'] = host_gts
            with _rf('sec:random_proposals'):
                random_proposals = self.get_random_proposal_list(img, gt_bboxes, kwargs)
            kwargs.pop('img_metas_host', None)
        with _rf('sec:backbone_fpn'):
            x = self.extract_feat(img)
        losses = dict()
        pending = {}
        if self.with_rpn:
            proposal_cfg = self.train_cfg.get('rpn_proposal', self.test_cfg.rpn if self.test_cfg else None)
            # the contrastive RoI head reads the view-1 proposal lists only (contrastive_roi_head.py:85-95)
            n_prop = kwargs['batch_size'] if 'num_views' in kwargs else None

            def after_proposals(props):
                if gt_bboxes_ignore is None and hasattr(self.roi_head, 'begin_sampling'):
                    with _rf('sec:roi_begin_sampling'):
                        pending['roi'] = self.roi_head.begin_sampling(props, gt_bboxes, gt_labels, len(img_metas),
                                                                      **kwargs)
            rpn_losses, proposal_list = self.rpn_head.forward_train(
                x, img_metas, gt_bboxes, gt_labels=None, gt_bboxes_ignore=gt_bboxes_ignore,
                proposal_cfg=proposal_cfg, num_proposal_imgs=n_prop, padded_proposals=True,
                after_proposals=after_proposals)
            losses.update(rpn_losses)
        else:
            proposal_list = proposals
        if random_proposals is not None:
            kwargs['random_proposal_list'] = random_proposals
        losses.update(self.roi_head.forward_train(x, img_metas, proposal_list, gt_bboxes, gt_labels,
                                                  gt_bboxes_ignore, gt_masks,
                                                  pending_sampling=pending.get('roi'), **kwargs))
        return losses

    @torch.no_grad()
    def simple_test(self, img, img_metas, proposals=None, rescale=False, **kwargs):
        """two_stage.py:224-266 without the debug / visualisation branches."""
        x = self.extract_feat(img)
        self.fpn_features = x
        proposal_list = self.rpn_head.simple_test_rpn(x, img_metas) if proposals is None else proposals
        return self.roi_head.simple_test(x, proposal_list, img_metas, rescale=rescale)

    @torch.no_grad()
    def aug_test(self, imgs, img_metas, rescale=False):
        """two_stage.py:268-277: test-time augmentation (MultiScaleFlipAug with several scales and / or flips).  If
        ``rescale`` is False the returned boxes fit the scale of imgs[0]."""
        x = [self.extract_feat(img) for img in imgs]                 # base.py extract_feats
        proposal_list = self.rpn_head.aug_test_rpn(x, img_metas)
        return self.roi_head.aug_test(x, proposal_list, img_metas, rescale=rescale)

    def get_random_proposal_list(self, img, gt_bboxes, kwargs):
        """two_stage.py:162-204, quirks kept: OA-Mix boxes are filtered against image 0's gts (:178,186);
        ``img.shape[2:]`` = (H, W) is consumed as (width, height) (:164,193 vs :395); new boxes use the gts
        of image ``i % num_views`` (:195)."""
        cfg = self.train_cfg['random_proposal_cfg']
        img_shape = img.shape[2:]
        B = img.shape[0]
        assert 'num_views' in kwargs, 'num_view is required'
        assert cfg['bbox_from'] == 'oagrb', 'oagrb is required'
        assert 'multilevel_bboxes' in kwargs or 'oamix_boxes' in kwargs, 'boxes are required'
        device = img.device
        metas = kwargs.get('img_metas_host')
        if metas is not None:     # the data pipeline kept the host copies of the gt boxes: no device read
            gts = [np.asarray(g, dtype=np.float32) for g in metas]
        else:
            gts = [g.detach().cpu().numpy() for g in gt_bboxes]
        out = []
        for b in kwargs.get('multilevel_boxes', []):
            b = b.to(torch.float32).cpu().numpy()
            out.append(b[np.max(bbox_overlaps_np(b, gts[0]), axis=1) < cfg['iou_max']])
        for i, b in enumerate(kwargs.get('oamix_boxes', [])):
            b = b.to(torch.float32).cpu().numpy()
            b = b[np.max(bbox_overlaps_np(b, gts[0]), axis=1) < cfg['iou_max']]
            out[i] = np.concatenate([out[i], b], axis=0)
        for i in range(B):
            new = generate_random_bboxes_xy(img_shape, num_bboxes=cfg['num_bboxes'],
                                            bboxes_xy=gts[i % kwargs['num_views']], scales=cfg['scales'],
                                            ratios=cfg['ratios'], iou_max=cfg['iou_max'], iou_min=cfg['iou_min'])
            out[i] = np.concatenate([out[i], new[:, :4].astype(np.float32)], axis=0)
        return [_pinned_to(torch.from_numpy(np.ascontiguousarray(o, dtype=np.float32)), device) for o in out]


@DETECTORS.register_module()
class FasterRCNN(TwoStageDetector):
    pass
