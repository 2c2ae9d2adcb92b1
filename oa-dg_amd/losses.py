"""Loss modules under the reference's registry names.

``CrossEntropyLossPlus`` (2 views + jsdv1_3_2aug) and ``ContrastiveLossPlus`` run entirely in the HIP library
(hip_ops.ce_jsd_loss / hip_ops.supcon_loss); the small regression losses and the stock baselines are plain
torch tensor expressions, as in the reference.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hip_ops
from .registry import LOSSES


def reduce_loss(loss, reduction):
    return {'none': lambda: loss, 'mean': loss.mean, 'sum': loss.sum}[reduction]()


def weight_reduce_loss(loss, weight=None, reduction='mean', avg_factor=None):
    """mmdet/models/losses/utils.py:30-56."""
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        return reduce_loss(loss, reduction)
    if reduction == 'mean':
        return loss.sum() / avg_factor
    if reduction != 'none':
        raise ValueError('avg_factor can not be used with reduction="sum"')
    return loss


def accuracy(pred, target, topk=1, thresh=None):
    """mmdet/models/losses/accuracy.py accuracy (top-k, percent)."""
    single = isinstance(topk, int)
    ks = (topk,) if single else tuple(topk)
    if pred.size(0) == 0:
        out = [pred.new_tensor(0.) for _ in ks]
        return out[0] if single else out
    assert pred.ndim == 2 and target.ndim == 1 and pred.size(0) == target.size(0)
    maxk = max(ks)
    assert maxk <= pred.size(1)
    val, lab = pred.topk(maxk, dim=1)
    lab = lab.t()
    correct = lab.eq(target.view(1, -1).expand_as(lab))
    if thresh is not None:
        correct = correct & (val > thresh).t()
    out = [correct[:k].reshape(-1).float().sum(0, keepdim=True).mul_(100.0 / pred.size(0)) for k in ks]
    return out[0] if single else out


def _view1(t, num_views):
    return None if t is None else torch.chunk(t, num_views)[0]


# ------------------------------------------------------------------------------------------------- OA-Loss
@LOSSES.register_module()
class CrossEntropyLossPlus(nn.Module):
    """mmdet/models/losses/oadg/cross_entropy_loss_plus.py:322-500.

    CE/BCE on the view-1 rows + lambda * JSD between the two views, one fused HIP forward and one fused HIP
    backward.  avg_factor is not divided by the number of views (avg='1.0')."""

    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean', class_weight=None,
                 ignore_index=None, loss_weight=1.0, additional_loss='jsd',
                 additional_loss_weight_reduce=False, lambda_weight=0.0001, wandb_name=None, num_views=3,
                 avg='1.0', **kwargs):
        super().__init__()
        assert not use_mask, 'mask cross entropy is not on the OA-DG detection path'
        if class_weight is not None or additional_loss_weight_reduce or avg != '1.0':
            raise NotImplementedError('only the settings used by the OA-DG configs are implemented')
        if ignore_index not in (None, -100):
            raise NotImplementedError('ignore_index other than -100')
        self.use_sigmoid, self.reduction, self.loss_weight = use_sigmoid, reduction, loss_weight
        self.additional_loss, self.lambda_weight = additional_loss, lambda_weight
        self.num_views, self.wandb_name, self.kwargs = num_views, wandb_name, kwargs
        self.wandb_features = dict()
        if additional_loss in ('jsdv1_3_2aug',):
            if num_views != 2:
                raise NotImplementedError('jsdv1_3_2aug is defined for exactly 2 views')
        elif additional_loss not in (None, 'None'):
            raise NotImplementedError(f'additional_loss={additional_loss!r} is not used by the named configs')

    @property
    def with_jsd(self):
        return self.additional_loss == 'jsdv1_3_2aug'

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None,
                ignore_index=None, **kwargs):
        assert reduction_override in (None, 'mean'), 'the HIP path implements reduction="mean"'
        if avg_factor is None:
            raise NotImplementedError('avg_factor is always given on the Faster R-CNN path')
        if cls_score.dim() == 1:
            cls_score = cls_score.view(-1, 1)
        if self.num_views != 2:
            raise NotImplementedError('num_views != 2')
        lam = self.lambda_weight if self.with_jsd else 0.0
        total, parts = hip_ops.ce_jsd_loss(cls_score, label, weight, self.use_sigmoid, float(avg_factor),
                                           self.loss_weight, lam)
        self.wandb_features[f'ce_loss({self.wandb_name})'] = parts[1]
        self.wandb_features[f'lam_additional_loss({self.wandb_name})'] = parts[2]
        return total


@LOSSES.register_module()
class ContrastiveLossPlus(nn.Module):
    """mmdet/models/losses/oadg/contrastive_loss_plus.py:12-50 over oadg/contrastive_loss.py:147-232.

    The reference hard-codes ori_size = 512 * num_views (valid for samples_per_gpu = 2 only).  Here the
    sampled block is len(labels) rows, so ori_size = len(labels) / num_views: identical at samples_per_gpu = 2,
    and the cross-view pairing stays right at any batch size.  ``ori_size`` may also be forced."""

    def __init__(self, loss_weight=1, temperature=0.07, num_views=2, normalized_input=True, min_samples=10,
                 ori_size=None, **kwargs):
        super().__init__()
        assert num_views == 2, 'Only num_views 2 is supported (contrastive_loss.py:189)'
        assert normalized_input, 'the kernel always L2-normalises its input'
        self.loss_weight, self.temperature, self.num_views = loss_weight, temperature, num_views
        self.normalized_input, self.min_samples, self.ori_size = normalized_input, min_samples, ori_size
        self.kwargs = kwargs

    def forward(self, cont_feats, labels):
        if len(cont_feats) == 0:
            return torch.zeros(1, device=cont_feats.device)
        K, B = labels.shape[0], cont_feats.shape[0]
        if self.ori_size is None:
            ori, rp = K // self.num_views, (B - K) // self.num_views
        else:
            ori = self.ori_size
            rp = (B % ori) // self.num_views
        return hip_ops.supcon_loss(cont_feats, labels, ori, rp, self.temperature, self.min_samples,
                                   self.loss_weight)


class _RegLossPlus(nn.Module):
    """L1LossPlus / SmoothL1LossPlus (oadg/smooth_l1_loss_plus.py:350-552): the element-wise loss of the
    view-1 chunk, weights chunked the same way, avg_factor untouched (utils.py:138-149)."""

    def __init__(self, reduction='mean', loss_weight=1.0, additional_loss='jsd', lambda_weight=0.0001,
                 wandb_name=None, analysis=False, num_views=3, **kwargs):
        super().__init__()
        if additional_loss not in (None, 'None'):
            raise NotImplementedError(f'additional_loss={additional_loss!r} is not used by the named configs')
        assert not analysis
        self.reduction, self.loss_weight, self.num_views = reduction, loss_weight, num_views
        self.wandb_name, self.lambda_weight, self.kwargs = wandb_name, lambda_weight, kwargs
        self.wandb_features = dict()

    def elementwise(self, pred, target):
        raise NotImplementedError

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        assert reduction_override in (None, 'none', 'mean', 'sum')
        reduction = reduction_override or self.reduction
        p, t = _view1(pred, self.num_views), _view1(target, self.num_views)
        loss = p.sum() * 0 if t.numel() == 0 else self.elementwise(p, t)
        w = torch.chunk(weight, self.num_views)[0]
        return self.loss_weight * weight_reduce_loss(loss, w, reduction, avg_factor)


@LOSSES.register_module()
class L1LossPlus(_RegLossPlus):

    def elementwise(self, pred, target):
        return torch.abs(pred - target)


@LOSSES.register_module()
class SmoothL1LossPlus(_RegLossPlus):

    def __init__(self, beta=1.0, **kwargs):
        super().__init__(**kwargs)
        assert beta > 0
        self.beta = beta

    def elementwise(self, pred, target):
        d = torch.abs(pred - target)
        return torch.where(d < self.beta, 0.5 * d * d / self.beta, d - 0.5 * self.beta)


# ------------------------------------------------------------------------------------------------- stock
@LOSSES.register_module()
class CrossEntropyLoss(nn.Module):
    """Stock mmdet CE/BCE (baseline configs without OA-Loss); plain torch, not part of the hot path."""

    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean', class_weight=None,
                 ignore_index=None, loss_weight=1.0):
        super().__init__()
        assert not use_mask and class_weight is None
        self.use_sigmoid, self.reduction, self.loss_weight = use_sigmoid, reduction, loss_weight
        self.ignore_index = -100 if ignore_index is None else ignore_index

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        reduction = reduction_override or self.reduction
        if self.use_sigmoid:
            if cls_score.dim() != label.dim():
                C = cls_score.size(-1)
                valid = (label >= 0) & (label != self.ignore_index)
                onehot = label.new_zeros((label.size(0), C))
                sel = valid & (label < C)
                onehot[sel.nonzero(as_tuple=True)[0], label[sel]] = 1
                vm = valid.view(-1, 1).expand(label.size(0), C).float()
                weight = vm if weight is None else weight.view(-1, 1).repeat(1, C) * vm
                label = onehot
            loss = F.binary_cross_entropy_with_logits(cls_score, label.float(), reduction='none')
        else:
            loss = F.cross_entropy(cls_score, label, reduction='none', ignore_index=self.ignore_index)
        if weight is not None:
            weight = weight.float()
        return self.loss_weight * weight_reduce_loss(loss, weight, reduction, avg_factor)


@LOSSES.register_module()
class L1Loss(nn.Module):

    def __init__(self, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.reduction, self.loss_weight = reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None):
        loss = pred.sum() * 0 if target.numel() == 0 else torch.abs(pred - target)
        return self.loss_weight * weight_reduce_loss(loss, weight, reduction_override or self.reduction,
                                                     avg_factor)


@LOSSES.register_module()
class SmoothL1Loss(nn.Module):

    def __init__(self, beta=1.0, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.beta, self.reduction, self.loss_weight = beta, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        if target.numel() == 0:
            loss = pred.sum() * 0
        else:
            d = torch.abs(pred - target)
            loss = torch.where(d < self.beta, 0.5 * d * d / self.beta, d - 0.5 * self.beta)
        return self.loss_weight * weight_reduce_loss(loss, weight, reduction_override or self.reduction,
                                                     avg_factor)
