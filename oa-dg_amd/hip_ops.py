"""torch.autograd glue over the C ABI of liboadg_hip.so.

Each function here validates shapes, allocates outputs/workspaces as torch tensors (device memory is
torch's, plumbing only) and enqueues the HIP kernels on torch's current stream.  No arithmetic of the
hot path happens in Python and nothing here falls back to torch ops.
"""
import ctypes
import os

import torch
import torch.nn.functional as F

from . import _lib
from ._lib import check, ptr, require_cuda, stream_ptr


# kernel name -> list of (start, end) torch.cuda.Event pairs recorded around that C-ABI call on the current
# stream; filled only while bench.py sets TIMERS to a dict (live HIP-event timing of the timed region)
TIMERS = None


def _timed(name, fn, *args, work=0.0):
    """``fn(*args)`` between two events on the calling thread's current stream when bench.py asked for ``name``;
    ``work`` = the launch's algorithmic FLOPs or bytes (0: the caller of the timers derives it)."""
    if TIMERS is None or name not in TIMERS:
        return fn(*args)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    rc = fn(*args)
    b.record()
    TIMERS[name].append((a, b, work))
    return rc


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 8), dtype=torch.uint8, device=device)


# --------------------------------------------------------------------------------------- OA-Loss: supcon
class _SupCon(torch.autograd.Function):

    @staticmethod
    def forward(ctx, feats, labels, n_labels, ori_size, rp_size, temper, min_samples, loss_weight):
        require_cuda(feats, labels)
        L = _lib.lib()
        feats = feats.contiguous().float()
        labels = labels.contiguous().view(-1).long()
        B, D = feats.shape
        nbytes = L.oadg_supcon_workspace_bytes(B, D)
        ws = _ws(nbytes, feats.device)
        out = torch.empty(1, dtype=torch.float32, device=feats.device)
        check(_timed('supcon_fwd', L.oadg_supcon_fwd, ptr(feats), ptr(labels), B, D, n_labels, ori_size, rp_size,
                     float(temper), int(min_samples), float(loss_weight), ptr(ws), nbytes,
                     ptr(out), stream_ptr(), work=2.0 * B * B * D), 'oadg_supcon_fwd')
        ctx.save_for_backward(labels, ws)
        ctx.args = (B, D, n_labels, ori_size, rp_size, float(temper), float(loss_weight), nbytes)
        return out[0]

    @staticmethod
    def backward(ctx, gout):
        labels, ws = ctx.saved_tensors
        B, D, n_labels, ori_size, rp_size, temper, loss_weight, nbytes = ctx.args
        L = _lib.lib()
        g = gout.contiguous().float().view(1)
        dfeats = torch.empty(B, D, dtype=torch.float32, device=ws.device)
        check(_timed('supcon_bwd', L.oadg_supcon_bwd, ptr(labels), B, D, n_labels, ori_size, rp_size, temper,
                     loss_weight, ptr(g), ptr(ws), nbytes, ptr(dfeats), stream_ptr(), work=4.0 * B * B * D),
              'oadg_supcon_bwd')
        return dfeats, None, None, None, None, None, None, None


def supcon_loss(feats, labels, ori_size, rp_size, temper=0.07, min_samples=10, loss_weight=1.0):
    """loss_weight * supcontrast(normalize(feats), labels) -- contrastive_loss_plus.py:31-50.

    feats [B, D]; labels [n_labels] (rows n_labels..B-1 inherit the last label)."""
    n_labels = labels.numel()
    return _SupCon.apply(feats, labels, n_labels, int(ori_size), int(rp_size), temper, min_samples,
                         loss_weight)


# --------------------------------------------------------------------------------------- OA-Loss: CE + JSD
class _CeJsd(torch.autograd.Function):

    @staticmethod
    def forward(ctx, logits, labels, weights, mode, avg_factor, loss_weight, lambda_jsd):
        require_cuda(logits, labels, weights)
        L = _lib.lib()
        logits = logits.contiguous().float()
        R = logits.shape[0]
        C = logits.numel() // max(R, 1)
        labels = labels.contiguous().view(-1).long()
        if weights is not None:
            weights = weights.contiguous().view(-1).float()
        nbytes = L.oadg_cls_loss_workspace_bytes()
        ws = _ws(nbytes, logits.device)
        out = torch.empty(3, dtype=torch.float32, device=logits.device)
        check(L.oadg_ce_jsd_fwd(ptr(logits), ptr(labels), ptr(weights), R, C, mode, float(avg_factor),
                                float(loss_weight), float(lambda_jsd), ptr(ws), nbytes, ptr(out),
                                stream_ptr()), 'oadg_ce_jsd_fwd')
        ctx.save_for_backward(logits, labels, weights)
        ctx.args = (R, C, mode, float(avg_factor), float(loss_weight), float(lambda_jsd))
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out

    @staticmethod
    def backward(ctx, gout, _gparts):
        logits, labels, weights = ctx.saved_tensors
        R, C, mode, avg_factor, loss_weight, lambda_jsd = ctx.args
        L = _lib.lib()
        g = gout.contiguous().float().view(1)
        dlogits = torch.empty_like(logits)
        check(L.oadg_ce_jsd_bwd(ptr(logits), ptr(labels), ptr(weights), R, C, mode, avg_factor,
                                loss_weight, lambda_jsd, ptr(g), ptr(dlogits), stream_ptr()),
              'oadg_ce_jsd_bwd')
        return dlogits, None, None, None, None, None, None


def ce_jsd_loss(logits, labels, weights, use_sigmoid, avg_factor, loss_weight=1.0, lambda_jsd=0.0):
    """CrossEntropyLossPlus(additional_loss='jsdv1_3_2aug') for 2 views.

    Returns (total, parts) with parts = [total, ce, lambda*jsd] (detached, for logging)."""
    return _CeJsd.apply(logits, labels, weights, 0 if use_sigmoid else 1, avg_factor, loss_weight,
                        lambda_jsd)


# --------------------------------------------------------------------------------------- fused RPN loss
class _RpnLoss(torch.autograd.Function):
    """AnchorHead.loss for all pyramid levels on the RPN head's channel-padded NHWC outputs (csrc/cls_loss.hip
    oadg_rpn_loss_fwd / _bwd): (loss_cls, loss_bbox, parts[4]) - sums over the levels."""

    @staticmethod
    def forward(ctx, A, targets, avg_factor, w_cls, lam, w_box, *ys):
        L = _lib.lib()
        labels, label_w, bbox_t, bbox_w = targets
        B, At = labels.shape
        dt = ys[0].dtype
        assert dt in (torch.float32, torch.bfloat16) and all(y.dtype == dt and y.shape[0] == B for y in ys)
        nbytes = L.oadg_rpn_loss_workspace_bytes()
        ws = _ws(nbytes, labels.device)
        out = torch.empty(4, dtype=torch.float32, device=labels.device)
        levels = _rpn_loss_levels(ys, None, A)
        check(L.oadg_rpn_loss_fwd(levels, len(ys), B, A, At, 0 if dt == torch.float32 else 1, ptr(labels), ptr(label_w),
                                  ptr(bbox_t), ptr(bbox_w), float(avg_factor), float(w_cls), float(lam), float(w_box),
                                  ptr(ws), nbytes, ptr(out), stream_ptr()), 'oadg_rpn_loss_fwd')
        ctx.save_for_backward(labels, label_w, bbox_t, bbox_w, *ys)
        ctx.args = (A, float(avg_factor), float(w_cls), float(lam), float(w_box))
        ctx.mark_non_differentiable(out)
        return out[0].clone(), out[3].clone(), out

    @staticmethod
    def backward(ctx, g_cls, g_box, _g):
        labels, label_w, bbox_t, bbox_w, *ys = ctx.saved_tensors
        A, avg, w_cls, lam, w_box = ctx.args
        L = _lib.lib()
        B, At = labels.shape
        dt = ys[0].dtype
        gys = [torch.empty(y.shape, dtype=dt, device=y.device, memory_format=torch.channels_last) for y in ys]
        levels = _rpn_loss_levels(ys, gys, A)
        gc = g_cls.contiguous().float().view(1) if g_cls is not None else None
        gb = g_box.contiguous().float().view(1) if g_box is not None else None
        if gc is None:
            gc = torch.zeros(1, device=labels.device)
        if gb is None:
            gb = torch.zeros(1, device=labels.device)
        check(L.oadg_rpn_loss_bwd(levels, len(ys), B, A, At, 0 if dt == torch.float32 else 1, ptr(labels), ptr(label_w),
                                  ptr(bbox_t), ptr(bbox_w), avg, w_cls, lam, w_box, ptr(gc), ptr(gb), stream_ptr()),
              'oadg_rpn_loss_bwd')
        return (None, None, None, None, None, None, *gys)


def _rpn_loss_levels(ys, gys, A):
    levels = (_lib.RpnLossLevel * len(ys))()
    first = pix = 0
    for i, y in enumerate(ys):
        N, Cy, H, W = y.shape
        lv = levels[i]
        lv.y = y.data_ptr()
        lv.gy = gys[i].data_ptr() if gys is not None else None
        lv.sN, lv.sC, lv.sH, lv.sW = (int(v) for v in y.stride())
        lv.H, lv.W, lv.Cy, lv.first, lv.pix0 = int(H), int(W), int(Cy), first, pix
        first += H * W * A
        pix += H * W
    return levels


def rpn_loss(ys, num_anchors, targets, avg_factor, w_cls, lambda_jsd, w_box):
    """(loss_cls, loss_bbox, parts) of AnchorHead.loss over all levels; ``ys``: the fused RPN head outputs [B, Cy, H, W]
    (channels [0, A) logits, [A, 5A) deltas), ``targets`` = (labels [B, At] int64, label_weights [B, At], bbox_targets
    [B, At, 4], bbox_weights [B, At, 4]), images [0, B/2) = view 1."""
    require_cuda(*ys, *targets)
    return _RpnLoss.apply(int(num_anchors), tuple(targets), avg_factor, w_cls, lambda_jsd, w_box, *ys)


# --------------------------------------------------------------------------------------- RoI head: box loss + accuracy
class _RoiRegAcc(torch.autograd.Function):
    """(loss_bbox, acc) of BBoxHead.loss on the head's raw outputs (csrc/cls_loss.hip oadg_roi_reg_acc_fwd / _bwd): the
    positives' class-specific deltas against their targets (L1 / SmoothL1, rows below ``reg_limit``) and the top-1
    accuracy, one launch forward, one backward (the whole [K, 4 C] gradient in bbox_pred's dtype)."""

    @staticmethod
    def forward(ctx, bbox_pred, cls_score, labels, targets, weights, num_classes, reg_limit, beta, avg_factor, loss_weight):
        require_cuda(bbox_pred, labels, targets, weights)
        dts = {torch.float32: 0, torch.bfloat16: 1}
        bp = bbox_pred.contiguous()
        cs = cls_score.detach().contiguous() if cls_score is not None else None
        K, n_reg = bp.shape
        out = torch.empty(2, dtype=torch.float32, device=bp.device)
        labels, targets, weights = labels.contiguous(), targets.float().contiguous(), weights.float().contiguous()
        check(_lib.lib().oadg_roi_reg_acc_fwd(ptr(bp), dts[bp.dtype], ptr(cs), dts[cs.dtype] if cs is not None else 0,
                                              ptr(labels), ptr(targets), ptr(weights), K, int(num_classes), n_reg,
                                              cs.shape[1] if cs is not None else 0, int(reg_limit), float(beta),
                                              float(avg_factor), float(loss_weight), ptr(out), stream_ptr()),
              'oadg_roi_reg_acc_fwd')
        ctx.save_for_backward(bp, labels, targets, weights)
        ctx.cfg = (int(num_classes), int(reg_limit), float(beta), float(avg_factor), float(loss_weight), dts[bp.dtype])
        loss, acc = out[0], out[1:2]
        ctx.mark_non_differentiable(acc)
        return loss, acc

    @staticmethod
    def backward(ctx, g, _gacc):
        bp, labels, targets, weights = ctx.saved_tensors
        C, reg_limit, beta, avg, lw, dt = ctx.cfg
        K, n_reg = bp.shape
        grad = torch.empty_like(bp)
        g = g.contiguous().float().view(1)
        check(_lib.lib().oadg_roi_reg_bwd(ptr(bp), dt, ptr(labels), ptr(targets), ptr(weights), K, C, n_reg, reg_limit, beta,
                                          avg, lw, ptr(g), ptr(grad), stream_ptr()), 'oadg_roi_reg_bwd')
        return grad, None, None, None, None, None, None, None, None, None


def roi_reg_acc(bbox_pred, cls_score, labels, bbox_targets, bbox_weights, num_classes, reg_limit, beta, avg_factor,
                loss_weight):
    """see :class:`_RoiRegAcc`; ``beta`` 0 = L1"""
    return _RoiRegAcc.apply(bbox_pred, cls_score, labels, bbox_targets, bbox_weights, num_classes, reg_limit, beta,
                            avg_factor, loss_weight)


FUSED_ROI_LOSS = os.environ.get('OADG_FUSED_ROI_LOSS', '1') == '1'


# --------------------------------------------------------------------------------------- _parse_losses
class _ParseLosses(torch.autograd.Function):
    """BaseDetector._parse_losses for one-element loss values (csrc/cls_loss.hip oadg_parse_losses): the per-variable sums,
    the total and the packed log vector in ONE launch instead of a mean / add chain of ~14; the backward hands the
    total's gradient to every value that is part of it as a view (no launch)."""

    @staticmethod
    def forward(ctx, name_of, n_names, mask, *vals):
        n = len(vals)
        dev = vals[0].device
        packed = torch.empty(n_names + 1, dtype=torch.float32, device=dev)
        total = torch.empty((), dtype=torch.float32, device=dev)
        P = (ctypes.c_void_p * n)(*[v.data_ptr() for v in vals])
        N = (ctypes.c_int * n)(*name_of)
        check(_lib.lib().oadg_parse_losses(ctypes.cast(P, ctypes.c_void_p), ctypes.cast(N, ctypes.c_void_p), n, n_names,
                                           int(mask), ptr(packed), ptr(total), stream_ptr()), 'oadg_parse_losses')
        ctx.meta = (tuple(name_of), int(mask), [tuple(v.shape) for v in vals])
        ctx.mark_non_differentiable(packed)
        return total, packed

    @staticmethod
    def backward(ctx, g, _gp):
        name_of, mask, shapes = ctx.meta
        return (None, None, None) + tuple(g.expand(shp) if (mask >> nm) & 1 else None for nm, shp in zip(name_of, shapes))


def parse_losses(values, name_of, n_names, mask):
    """(total, packed [n_names + 1]) - see :class:`_ParseLosses`; ``values``: one-element fp32 CUDA tensors"""
    return _ParseLosses.apply(tuple(name_of), int(n_names), int(mask), *values)


FUSED_PARSE_LOSSES = os.environ.get('OADG_FUSED_PARSE_LOSSES', '1') == '1'


# --------------------------------------------------------------------------------------- gradients straight into DDP buckets
# apis.FlatGradReducer sets GRAD_SINK = {parameter: its slice of the flat fp32 bucket buffer, in the parameter's layout}.  The
# functions that PRODUCE a parameter's gradient with a kernel of their own (convolution weights / BN scales:
# hip_conv._PrepWeights; the RoI head's linears: _CastAll / _FcWeightPermute below) then write it there instead of into a new
# tensor: AccumulateGrad adopts a gradient it holds the only reference to, so ``param.grad`` IS the bucket slice and the
# reducer has nothing to pack (the round-4 form copied every gradient into its bucket: 166 MB read + 166 MB written per step).
GRAD_SINK = None
_SINK_CLAIMED = set()       # ids of the parameters whose slice has been handed out in this backward pass


def sink_reset():
    """a new backward pass begins (FlatGradReducer.finish / hip_conv.begin_step): every slice may be handed out once again"""
    _SINK_CLAIMED.clear()


def grad_dest(param, like=None):
    """the tensor a producer should write ``param``'s gradient into: a fresh alias of the parameter's bucket slice when a
    reducer is active, the parameter holds no gradient yet AND nobody has been handed the slice in this backward pass
    (``param.grad`` is still None for every use of a leaf until AccumulateGrad runs, so a second contribution of the same
    pass is recognised by the claim set: it gets a tensor of its own and autograd SUMS the two), else a new fp32 tensor
    shaped and laid out like ``like`` (default: the parameter)"""
    sink = GRAD_SINK
    if sink is not None and param is not None and param.grad is None and id(param) not in _SINK_CLAIMED:
        v = sink.get(param)
        if v is not None:
            _SINK_CLAIMED.add(id(param))
            return v.detach()
    like = param if like is None else like
    return torch.empty_like(like, dtype=torch.float32)


def _foreach_copy(dst, src):
    f = getattr(torch, '_foreach_copy_', None)
    if f is not None:
        f(dst, src)
    else:
        for d, s_ in zip(dst, src):
            d.copy_(s_)


class _CastAll(torch.autograd.Function):
    """fp32 parameters -> bf16 copies in ONE multi-tensor pass, their gradients back to fp32 in one pass: what autocast
    does per ``F.linear`` call (a weight cast and a bias cast forward, two gradient casts backward: 24 tiny launches for
    the RoI head's five remaining linears, in the stretch of the step where the device waits for the host)."""

    @staticmethod
    def forward(ctx, *params):
        outs = [torch.empty_like(p, dtype=torch.bfloat16) for p in params]
        _foreach_copy(outs, [p.detach() for p in params])
        ctx.params = params if all(p.is_leaf for p in params) else None
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        live = [g for g in grads if g is not None]
        ps = [p for p, g in zip(ctx.params, grads) if g is not None] if ctx.params is not None else [None] * len(live)
        outs = [grad_dest(p, g) if (p is not None and g.is_contiguous()) else torch.empty_like(g, dtype=torch.float32)
                for p, g in zip(ps, live)]
        if live:
            _foreach_copy(outs, live)
        it = iter(outs)
        return tuple(next(it) if g is not None else None for g in grads)


def cast_all_bf16(params):
    """bf16 copies of ``params`` (a list of fp32 CUDA tensors), differentiable - see :class:`_CastAll`"""
    return _CastAll.apply(*params)


FC_CAST_ONCE = True


# --------------------------------------------------------------------------------------- FC weight on NHWC features
class _FcWeightPermute(torch.autograd.Function):
    """W fp32 [O, C*P] (column c*P + p, the reference's NCHW flatten) -> bf16 [O, P*C] (column p*C + c, the order of
    RoIAlign's NHWC output); backward: the bf16 gradient back as fp32 in the parameter's layout.  These are the two
    passes autocast makes over the weight anyway (cast forward, cast of the gradient), so the 105 MB feature permutation
    of ``x.flatten(1)`` and its backward disappear (csrc/eltwise.hip oadg_fc_weight_permute)."""

    @staticmethod
    def forward(ctx, w, C, P):
        require_cuda(w)
        O = w.shape[0]
        assert w.dtype == torch.float32 and w.is_contiguous() and w.shape[1] == C * P
        out = torch.empty((O, P * C), dtype=torch.bfloat16, device=w.device)
        check(_lib.lib().oadg_fc_weight_permute(ptr(w), ptr(out), O, C, P, 0, stream_ptr()), 'oadg_fc_weight_permute')
        ctx.meta = (O, C, P)
        ctx.param = w if w.is_leaf else None
        return out

    @staticmethod
    def backward(ctx, g):
        O, C, P = ctx.meta
        g = g.contiguous()
        if g.dtype != torch.bfloat16:
            g = g.to(torch.bfloat16)
        out = grad_dest(ctx.param) if ctx.param is not None else torch.empty((O, C * P), dtype=torch.float32, device=g.device)
        check(_lib.lib().oadg_fc_weight_permute(ptr(g), ptr(out), O, C, P, 1, stream_ptr()), 'oadg_fc_weight_permute')
        return out, None, None


def fc_weight_permuted(weight, C, P):
    """bf16 copy of an FC weight with its input columns in (p, c) order - see :class:`_FcWeightPermute`"""
    return _FcWeightPermute.apply(weight, int(C), int(P))


FC_PERMUTE = True


class _LinearBiasGrad(torch.autograd.Function):
    """y = [relu](x @ W^T + b) on bf16 operands (the RoI head's linears under autocast) with the backward pass of the bias /
    ReLU part on the library's own kernel: g = dy * (y > 0) and db = column sums of g in ONE pass with fixed-order partials
    (csrc/eltwise.hip oadg_relu_bias_bwd - what the convolutions use), then dx = g @ W, dW = g^T @ x through the library
    GEMMs as before.  Replaces torch's threshold_backward + sum(0) pair per layer; round 6: that column reduction was the one
    piece of a training step that was not bit-reproducible beside a second process on the device
    (profiles/r06_packed_fp32_hazard.txt item 8).  Out-feature counts that are not a multiple of 8 (fc_cls: 9) stay on
    F.linear."""

    @staticmethod
    def forward(ctx, x, w, b, relu):
        y = F.linear(x, w, b)
        if relu:
            y = torch.relu_(y)
        ctx.save_for_backward(x, w, y if relu else None)
        ctx.has_b = b is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        from . import hip_conv
        x, w, y = ctx.saved_tensors
        gy = gy.contiguous()
        if gy.dtype != torch.bfloat16:
            gy = gy.to(torch.bfloat16)
        M, K = gy.shape
        g, db = hip_conv.relu_bias_bwd(gy.view(M, K, 1, 1), y.view(M, K, 1, 1) if y is not None else None, ctx.has_b)
        g = g.reshape(M, K)
        gx = g @ w if ctx.needs_input_grad[0] else None
        gw = g.t() @ x if ctx.needs_input_grad[1] else None
        return gx, gw, (db if ctx.needs_input_grad[2] else None), None


def linear_bias_grad(x, w, b, relu=False):
    """``[relu](F.linear(x, w, b))`` for bf16 CUDA operands - see :class:`_LinearBiasGrad`; anything else goes to F.linear"""
    if LINEAR_BIAS_GRAD and x.is_cuda and x.dim() == 2 and x.dtype == w.dtype == torch.bfloat16 and w.shape[0] % 8 == 0 and \
            (b is None or b.dtype == torch.bfloat16) and x.is_contiguous() and torch.is_grad_enabled():
        return _LinearBiasGrad.apply(x, w, b, bool(relu))
    y = F.linear(x, w, b)
    return torch.relu_(y) if relu else y


LINEAR_BIAS_GRAD = True


# --------------------------------------------------------------------------------------- RoIAlign
def _pyramid_args(maps, scales):
    n = len(maps)
    P = (ctypes.c_void_p * n)(*[m.data_ptr() for m in maps])
    Hs = (ctypes.c_int * n)(*[m.shape[2] for m in maps])
    Ws = (ctypes.c_int * n)(*[m.shape[3] for m in maps])
    Ss = (ctypes.c_float * n)(*[float(s) for s in scales])
    return P, Hs, Ws, Ss


def _as_nhwc(x):
    # logical [N,C,H,W] with channels_last strides == physical NHWC
    return x.contiguous(memory_format=torch.channels_last)


ROI_LOCALITY_ORDER = True
# bf16 backward by output tiles (csrc roi_align_bwd_tiles: no fp32 maps, no zero fill, no atomics, no cast pass,
# deterministic summation order).  Round 3: the default - with the pair loop pipelined (next slab in flight, bin ranges
# per tile row / column, one barrier per pair, 8 waves) a hot coarse-level tile no longer dominates the launch; the
# fp32 atomic scatter (OADG_ROI_BWD_TILES=0; fp32 atomics retire at ~1.2 TB/s of 4-byte adds on this chip whatever their
# scope - tools/probe/atomic_lab.hip - so 236 M of them cannot take less than 0.8 ms) stays for fp32 maps and PH, PW > 8.
BWD_TILES = os.environ.get('OADG_ROI_BWD_TILES', '1') == '1'
ROI_ORDER_ONE_LAUNCH = True
_GROUP_KEYS = {}


def _group_keys(levels, n_img, device):
    """first sort key of every (level, image) group of oadg_roi_order_keys (+ the end sentinel), cached"""
    k = (levels, n_img, device)
    g = _GROUP_KEYS.get(k)
    if g is None:
        g = _GROUP_KEYS[k] = torch.arange(levels * n_img + 1, device=device, dtype=torch.int64) << 20
    return g


class _RoIAlignFPN(torch.autograd.Function):

    @staticmethod
    def forward(ctx, rois, out_size, scales, finest_scale, sampling_ratio, aligned, tokens, *feats):
        require_cuda(rois, *feats)
        ctx.tokens = tokens
        L = _lib.lib()
        dt = feats[0].dtype
        if dt not in (torch.float32, torch.bfloat16):
            raise TypeError(f'RoIAlign supports fp32/bf16 feature maps, got {dt}')
        feats = [_as_nhwc(f) for f in feats]
        N, C = feats[0].shape[:2]
        rois = rois.contiguous().float()
        K = rois.shape[0]
        PH, PW = out_size
        out = torch.empty((K, C, PH, PW), dtype=dt, device=rois.device,
                          memory_format=torch.channels_last)
        P, Hs, Ws, Ss = _pyramid_args(feats, scales)
        order = rng_ = None
        tiles = BWD_TILES and dt == torch.bfloat16 and PH <= 8 and PW <= 8 and K > 0
        if ROI_LOCALITY_ORDER and (K >= 512 or tiles):
            # process the RoIs level by level, image by image, cell by cell (the result does not depend on the order):
            # neighbouring RoIs share feature rows and - backward - gradient lines while those are still in L2
            if K <= 8192 and ROI_ORDER_ONE_LAUNCH:
                # stable argsort of the keys + the (level, image) group boundaries by counting, one launch (csrc oadg_roi_order)
                order = torch.empty((K,), dtype=torch.int32, device=rois.device)
                rng_ = torch.empty((len(feats) * N + 1,), dtype=torch.int32, device=rois.device)
                check(L.oadg_roi_order(ptr(rois), K, N, len(feats), float(finest_scale), ptr(order), ptr(rng_),
                                       stream_ptr()), 'oadg_roi_order')
                if not tiles:
                    rng_ = None
                keys = None
            else:
                keys = torch.empty((K,), dtype=torch.int64, device=rois.device)
                check(L.oadg_roi_order_keys(ptr(rois), K, N, len(feats), float(finest_scale), ptr(keys), stream_ptr()),
                      'oadg_roi_order_keys')
            if keys is None:
                pass
            elif tiles:     # the tile-gather backward walks the RoIs of one (level, image) group: group boundaries
                skeys, order = keys.sort()
                order = order.int()
                rng_ = torch.searchsorted(skeys, _group_keys(len(feats), N, rois.device), out_int32=True)
            else:
                order = keys.argsort().int()
        check(_timed('roi_align_fwd', L.oadg_roi_align_fwd, P, Hs, Ws, Ss, len(feats), N, C,
                     0 if dt == torch.float32 else 1, float(finest_scale), ptr(rois), K, PH, PW,
                     int(sampling_ratio), int(bool(aligned)), ptr(out), ptr(order), stream_ptr()), 'oadg_roi_align_fwd')
        ctx.save_for_backward(rois, order if order is not None else rois.new_zeros(0),
                              rng_ if rng_ is not None else rois.new_zeros(0))
        ctx.meta = ([tuple(f.shape) for f in feats], dt, tuple(scales), float(finest_scale),
                    int(sampling_ratio), int(bool(aligned)), (PH, PW))
        return out

    @staticmethod
    def backward(ctx, gout):
        rois, order, rng_ = ctx.saved_tensors
        order = order if order.numel() else None
        shapes, dt, scales, finest_scale, sampling_ratio, aligned, (PH, PW) = ctx.meta
        L = _lib.lib()
        gout = gout.contiguous(memory_format=torch.channels_last)
        if gout.dtype != dt:
            gout = gout.to(dt)
        if rng_.numel() and order is not None:
            # bf16: every element of the gradient maps is written once by its output tile (csrc roi_align_bwd_tiles)
            grads = [torch.empty(s, dtype=dt, device=rois.device, memory_format=torch.channels_last) for s in shapes]
            N, C = shapes[0][:2]
            P, Hs, Ws, Ss = _pyramid_args(grads, scales)
            boxes = torch.empty((max(rois.shape[0], 1), 4), dtype=torch.int32, device=rois.device)
            check(_timed('roi_align_bwd', L.oadg_roi_align_bwd_tiles, P, Hs, Ws, Ss, len(grads), N, C, finest_scale,
                         ptr(rois), rois.shape[0], PH, PW, sampling_ratio, aligned, ptr(gout), ptr(order), ptr(rng_),
                         ptr(boxes), stream_ptr()), 'oadg_roi_align_bwd_tiles')
            return (None, None, None, None, None, None, None, *_deposit(ctx.tokens, grads))
        grads = [torch.empty(s, dtype=torch.float32, device=rois.device,
                             memory_format=torch.channels_last).zero_() for s in shapes]
        N, C = shapes[0][:2]
        P, Hs, Ws, Ss = _pyramid_args(grads, scales)
        check(_timed('roi_align_bwd', L.oadg_roi_align_bwd, P, Hs, Ws, Ss, len(grads), N, C,
                     0 if dt == torch.float32 else 1, finest_scale, ptr(rois), rois.shape[0], PH, PW,
                     sampling_ratio, aligned, ptr(gout), ptr(order), stream_ptr()), 'oadg_roi_align_bwd')
        grads = [g if dt == torch.float32 else g.to(dt) for g in grads]
        return (None, None, None, None, None, None, None, *_deposit(ctx.tokens, grads))


def roi_align_fpn(feats, rois, out_size, scales, finest_scale=56, sampling_ratio=0, aligned=True):
    """RoIAlign of rois [K,5] over a list of maps [N,C,H_l,W_l]; the level of each RoI follows
    single_level_roi_extractor.py:36-55.  Returns [K,C,PH,PW] (channels_last memory)."""
    if isinstance(out_size, int):
        out_size = (out_size, out_size)
    tokens = tuple(getattr(f, '_oadg_token', None) for f in feats)
    return _RoIAlignFPN.apply(rois, tuple(out_size), tuple(scales), finest_scale, sampling_ratio,
                              aligned, tokens if any(t is not None for t in tokens) else None, *feats)


def _deposit(tokens, grads):
    """A map whose gradient another operation finishes (hip_conv.GradToken armed by the RPN convolution's forward and
    not yet closed): leave this part on the token - that launch adds it in its epilogue - and return nothing."""
    if tokens is None:
        return grads
    from . import hip_conv
    out = []
    for tok, g in zip(tokens, grads):
        if tok is not None and hip_conv.DEPOSIT and tok.armed and not tok.closed and g.dtype == torch.bfloat16:
            tok.extra = g if tok.extra is None else tok.extra + g
            out.append(None)
        else:
            if tok is not None and tok.closed:
                tok.grad_ptr = tok.colsum = None        # late depositor: the producer must not trust the finisher's result
            out.append(g)
    return out


# ------------------------------------------------------------------------------- FPN top-down
class _FpnTopDown(torch.autograd.Function):
    """lat + nearest_upsample(top) (necks/fpn.py:166-175) on bf16 NHWC maps, one pass each way."""

    @staticmethod
    def forward(ctx, lat, top):
        require_cuda(lat, top)
        lat, top = _as_nhwc(lat), _as_nhwc(top)
        N, C, H, W = lat.shape
        out = torch.empty_like(lat)
        check(_lib.lib().oadg_fpn_topdown_fwd(ptr(lat), ptr(top), ptr(out), N, H, W, top.shape[2], top.shape[3], C,
                                              stream_ptr()), 'oadg_fpn_topdown_fwd')
        ctx.shapes = (tuple(lat.shape), tuple(top.shape))
        return out

    @staticmethod
    def backward(ctx, g):
        (N, C, H, W), ts = ctx.shapes
        g = _as_nhwc(g.to(torch.bfloat16))
        dtop = None
        if ctx.needs_input_grad[1]:
            dtop = torch.empty(ts, dtype=torch.bfloat16, device=g.device, memory_format=torch.channels_last)
            check(_lib.lib().oadg_fpn_topdown_bwd(ptr(g), ptr(dtop), N, H, W, ts[2], ts[3], C, stream_ptr()),
                  'oadg_fpn_topdown_bwd')
        return (g if ctx.needs_input_grad[0] else None), dtop


def fpn_topdown(lat, top):
    """laterals[i-1] + F.interpolate(laterals[i], size=laterals[i-1].shape[2:], mode='nearest')."""
    return _FpnTopDown.apply(lat, top)


# --------------------------------------------------------------------------------------- NMS
def stem_conv(x, wp):
    """conv2d(x [N,3,H,W], w [64,3,7,7], stride 2, padding 3) without bias for a bf16 channels_last image, on the
    MFMA stem kernel (csrc/stem_conv.hip); ``wp`` from :func:`stem_weights`.  No gradient (frozen stem)."""
    require_cuda(x)
    assert x.dtype == torch.bfloat16 and x.shape[1] == 3 and not x.requires_grad
    x = x.contiguous(memory_format=torch.channels_last)
    N, _, H, W = x.shape
    y = torch.empty((N, 64, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=torch.bfloat16, device=x.device,
                    memory_format=torch.channels_last)
    check(_lib.lib().oadg_stem_conv7x7s2_nhwc_bf16(ptr(x), ptr(wp), ptr(y), N, H, W, stream_ptr()),
          'oadg_stem_conv7x7s2_nhwc_bf16')
    return y


def stem_weights(w):
    """fp32 [64,3,7,7] -> bf16 [64][7][8][4]: filter row, 8 input pixels (the first one left of the filter: zero), 4
    channels (the fourth zero) - the reduction order of the stem kernel."""
    assert tuple(w.shape) == (64, 3, 7, 7)
    wp = torch.zeros((64, 7, 8, 4), dtype=torch.bfloat16, device=w.device)
    wp[:, :, 1:, :3] = w.detach().permute(0, 2, 3, 1).to(torch.bfloat16)
    return wp.contiguous()


def bias_relu_maxpool(x, bias):
    """max_pool2d(relu(x + bias[None, :, None, None]), 3, 2, 1) for a bf16 channels_last map without gradient (the frozen
    ResNet stem, resnet.py:631-637): one pass over the 537 MB stem output instead of three."""
    require_cuda(x)
    assert x.dtype == torch.bfloat16 and not x.requires_grad and x.shape[1] % 8 == 0
    x = x.contiguous(memory_format=torch.channels_last)
    N, C, H, W = x.shape
    out = torch.empty((N, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=torch.bfloat16, device=x.device,
                      memory_format=torch.channels_last)
    b = bias.detach().float().contiguous() if bias is not None else None
    check(_lib.lib().oadg_bias_relu_maxpool_nhwc_bf16(ptr(x), ptr(b), ptr(out), N, H, W, C, stream_ptr()),
          'oadg_bias_relu_maxpool_nhwc_bf16')
    return out


def nms_sorted_batched(boxes, counts, iou_thr, max_keep=-1):
    """Greedy NMS on boxes [I, Mmax, 4] already sorted by descending score (and class-offset).

    counts [I] int32 = valid boxes per image.  Returns (keep [I, Mmax] int32, keep_cnt [I] int32);
    keep[i, :keep_cnt[i]] are indices into the sorted order, ascending."""
    require_cuda(boxes, counts)
    L = _lib.lib()
    boxes = boxes.contiguous().float()
    counts = counts.contiguous().int()
    n_img, Mmax = boxes.shape[:2]
    nbytes = L.oadg_nms_workspace_bytes(n_img, Mmax)
    ws = _ws(nbytes, boxes.device)
    keep = torch.empty((n_img, Mmax), dtype=torch.int32, device=boxes.device)
    keep_cnt = torch.empty((n_img,), dtype=torch.int32, device=boxes.device)
    check(L.oadg_nms_batched(ptr(boxes), ptr(counts), n_img, Mmax, float(iou_thr), int(max_keep),
                             ptr(ws), nbytes, ptr(keep), ptr(keep_cnt), stream_ptr()), 'oadg_nms_batched')
    return keep, keep_cnt
