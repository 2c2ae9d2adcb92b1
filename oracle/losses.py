"""CPU (torch, fp32) restatement of the OA-Loss family.  Test infrastructure only (see oracle/__init__).

All paths are relative to /root/reference/mmdet/models/losses/.
"""
import torch
import torch.nn.functional as F


def reduce_weighted(loss, weight=None, reduction='mean', avg_factor=None):
    """utils.py:30-56 weight_reduce_loss."""
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        return {'none': loss, 'mean': loss.mean(), 'sum': loss.sum()}[reduction]
    if reduction == 'mean':
        return loss.sum() / avg_factor
    if reduction == 'none':
        return loss
    raise ValueError('avg_factor can not be used with reduction="sum"')


def view1(t, num_views):
    return None if t is None else torch.chunk(t, num_views)[0]


def ce_view1(pred, label, weight, avg_factor, num_views=2, ignore_index=-100):
    """oadg/cross_entropy_loss_plus.py:11-58: softmax CE on the first view's rows only; avg_factor is NOT
    divided by the number of views (avg='1.0')."""
    loss = F.cross_entropy(view1(pred, num_views), view1(label, num_views), reduction='none',
                           ignore_index=ignore_index)
    return reduce_weighted(loss, view1(weight, num_views).float(), 'mean', avg_factor)


def bce_view1(pred, label, weight, avg_factor, num_views=2, ignore_index=-100):
    """oadg/cross_entropy_loss_plus.py:61-130: labels -> one-hot over pred.size(-1) channels, invalid labels
    get weight 0, then BCE-with-logits on the first view's rows."""
    if pred.dim() != label.dim():
        C = pred.size(-1)
        valid = (label >= 0) & (label != ignore_index)
        onehot = label.new_zeros((label.size(0), C))
        sel = valid & (label < C)
        onehot[sel.nonzero(as_tuple=True)[0], label[sel]] = 1
        vm = valid.view(-1, 1).expand(label.size(0), C).float()
        weight = vm if weight is None else weight.view(-1, 1).repeat(1, C) * vm
        label = onehot
    loss = F.binary_cross_entropy_with_logits(view1(pred, num_views), view1(label, num_views).float(),
                                              reduction='none')
    return reduce_weighted(loss, view1(weight, num_views).float(), 'mean', avg_factor)


def jsd_2views(pred, avg_factor):
    """oadg/cross_entropy_loss_plus.py:264-319 jsdv1_3_2aug (weight is None at the call site because
    additional_loss_weight_reduce=False, :454-455; the '/ len(p_aug1)' is a division by 1, :299-310)."""
    a, b = torch.chunk(pred, 2)
    if a.shape[-1] == 1:
        sa, sb = torch.sigmoid(a), torch.sigmoid(b)
        pa, pb = torch.cat((sa, 1 - sa), 1), torch.cat((sb, 1 - sb), 1)
    else:
        pa, pb = F.softmax(a, 1), F.softmax(b, 1)
    logm = torch.clamp((pa + pb) / 2., 1e-7, 1).log()
    kl = (F.kl_div(logm, pa, reduction='none') + F.kl_div(logm, pb, reduction='none')) / 2.
    return kl.sum() / avg_factor


def ce_jsd(pred, label, weight, avg_factor, use_sigmoid, loss_weight=1.0, lambda_weight=0.0, num_views=2):
    """CrossEntropyLossPlus.forward, oadg/cross_entropy_loss_plus.py:418-500, additional_loss='jsdv1_3_2aug'.
    Returns (total, ce_part, lambda*jsd_part)."""
    crit = bce_view1 if use_sigmoid else ce_view1
    ce = loss_weight * crit(pred, label, weight, avg_factor, num_views)
    js = lambda_weight * jsd_2views(pred, avg_factor)
    return ce + js, ce, js


def l1_view1(pred, target, weight, avg_factor, num_views=2, loss_weight=1.0):
    """L1LossPlus, oadg/smooth_l1_loss_plus.py:43-62,506-552 via weighted_loss2 (utils.py:138-149)."""
    p, t = view1(pred, num_views), view1(target, num_views)
    loss = p.sum() * 0 if t.numel() == 0 else (p - t).abs()
    return loss_weight * reduce_weighted(loss, view1(weight, num_views), 'mean', avg_factor)


def smooth_l1_view1(pred, target, weight, avg_factor, beta=1.0, num_views=2, loss_weight=1.0):
    """SmoothL1LossPlus, oadg/smooth_l1_loss_plus.py:12-39,350-440."""
    p, t = view1(pred, num_views), view1(target, num_views)
    if t.numel() == 0:
        loss = p.sum() * 0
    else:
        d = (p - t).abs()
        loss = torch.where(d < beta, 0.5 * d * d / beta, d - 0.5 * beta)
    return loss_weight * reduce_weighted(loss, view1(weight, num_views), 'mean', avg_factor)


def twin_index(B, ori, rp):
    """oadg/contrastive_loss.py:199-206: cross-view partner of every row (-1 = none)."""
    t = torch.full((B,), -1, dtype=torch.long)
    i = torch.arange(B)
    t[i < ori] = i[i < ori] + ori
    m = (i >= ori) & (i < 2 * ori)
    t[m] = i[m] - ori
    j = i - 2 * ori
    m = (j >= 0) & (j < rp)
    t[m] = i[m] + rp
    m = (j >= rp) & (j < 2 * rp)
    t[m] = i[m] - rp
    return t


def supcon(feats, labels, ori_size=None, rp_size=None, num_views=2, temper=0.07, min_samples=10,
           loss_weight=1.0):
    """ContrastiveLossPlus.forward (oadg/contrastive_loss_plus.py:31-50) -> supcontrast
    (oadg/contrastive_loss.py:170-232) -> supcontrast_mask (:147-167), without the B x B masks.

    ori_size/rp_size default to the reference's literal values (512*num_views, (B % ori)//num_views)."""
    B = feats.shape[0]
    labels = labels.view(-1)
    if labels.numel() != B:  # random-proposal rows take the last label
        labels = torch.cat([labels, labels[-1:].repeat(B - labels.numel())])
    ori = 512 * num_views if ori_size is None else ori_size
    rp = (B % ori) // num_views if rp_size is None else rp_size
    bg = labels.max()
    if int((labels != bg).sum()) <= min_samples:
        return feats.sum() * 0
    f = F.normalize(F.normalize(feats, dim=1), dim=1)  # normalised twice: plus.py:41 and loss.py:155
    S = f @ f.t() / temper
    L = S - S.max(dim=1, keepdim=True)[0].detach()
    eye = torch.eye(B, dtype=torch.bool)
    fg = labels != bg
    same = labels.view(-1, 1) == labels.view(1, -1)
    P = same & fg.view(-1, 1) & fg.view(1, -1) & ~eye
    tw = twin_index(B, ori, rp)
    has = tw >= 0
    rows = torch.arange(B)[has]
    twin = torch.zeros(B, B, dtype=torch.bool)
    twin[rows, tw[has]] = True
    P = (P | (twin & (~fg).view(-1, 1) & (~fg).view(1, -1))).float()
    logZ = torch.log((torch.exp(L) * (~eye).float()).sum(1, keepdim=True))
    lp = L - logZ
    per_row = (P * lp).sum(1) / (P.sum(1) + 1e-8)
    return loss_weight * (-per_row).mean()
