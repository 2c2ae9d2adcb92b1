"""-m gpu: the fused proposal path (csrc/proposals.hip through RPNHead._fused_proposals) against the tensor expressions of
RPNHead.get_bboxes (rpn_head.py:103-235 batched; delta2bbox; batched_nms offsets) - bit for bit, padded and unpadded lists."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _head(dev, nms_pre=2000, max_per_img=1000, min_bbox_size=0):
    from oadg_amd import Config
    from oadg_amd.dense_heads import RPNHead
    test_cfg = Config(dict(nms_pre=nms_pre, max_per_img=max_per_img, nms=dict(type='nms', iou_threshold=0.7),
                           min_bbox_size=min_bbox_size))
    h = RPNHead(in_channels=256, feat_channels=256,
                anchor_generator=dict(type='AnchorGenerator', scales=[8], ratios=[0.5, 1.0, 2.0], strides=[4, 8, 16, 32, 64]),
                bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[.0, .0, .0, .0], target_stds=[1.0, 1.0, 1.0, 1.0]),
                loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                loss_bbox=dict(type='L1Loss', loss_weight=1.0), test_cfg=test_cfg)
    return h.to(dev)


def _outputs(dev, n, H, W, seed, bf16_fused, saturate=False):
    """per-level (cls_score, bbox_pred): either separate fp32 NCHW tensors or views of one 128-channel channels_last bf16
    tensor (the fused RPN head's layout)"""
    g = torch.Generator(device=dev).manual_seed(seed)
    cs, bp = [], []
    for s in (4, 8, 16, 32, 64):
        h, w = -(-H // s), -(-W // s)
        if bf16_fused:
            y = (torch.randn(n, 128, h, w, device=dev, generator=g) * (8.0 if saturate else 1.0)).bfloat16() \
                .contiguous(memory_format=torch.channels_last)
            y[:, 3:15] *= 0.3
            cs.append(y[:, :3])
            bp.append(y[:, 3:15])
        else:
            c = torch.randn(n, 3, h, w, device=dev, generator=g) * (20.0 if saturate else 2.0)
            if saturate:
                c = c.round()                    # exact ties, sigmoid saturated to 1.0 for many of them
            cs.append(c)
            bp.append(torch.randn(n, 12, h, w, device=dev, generator=g) * 0.4)
    return cs, bp


@pytest.mark.parametrize('bf16_fused,saturate,min_size,shapes', [
    (True, False, 0, None), (False, False, 0, None), (False, True, 0, None), (True, True, 16, None),
    (False, False, 24, [(200, 300), (256, 384), (180, 384)]), (True, False, -1, None)])
def test_fused_proposals_equal_the_tensor_path(dev, bf16_fused, saturate, min_size, shapes):
    from oadg_amd.dense_heads import RPNHead
    n, H, W = 3, 256, 384
    head = _head(dev, min_bbox_size=min_size)
    cs, bp = _outputs(dev, n, H, W, 3 + int(bf16_fused) + 2 * int(saturate), bf16_fused, saturate)
    shapes = shapes or [(H, W)] * n
    metas = [dict(img_shape=(h, w, 3), pad_shape=(H, W, 3)) for h, w in shapes]
    res = {}
    for fused in (True, 'sort', False):          # kernels incl. the radix-select top-k / kernels after torch.sort / tensors
        RPNHead.FUSED_PROPOSALS = bool(fused)
        RPNHead.FUSED_TOPK = fused is True
        try:
            res[fused, True] = head.get_bboxes(cs, bp, img_metas=metas, padded=True)
            res[fused, False] = head.get_bboxes(cs, bp, img_metas=metas, padded=False)
            res[fused, 'n2'] = head.get_bboxes(cs, bp, img_metas=metas, num_imgs=2, padded=True)
        finally:
            RPNHead.FUSED_PROPOSALS = RPNHead.FUSED_TOPK = True
    for key in ((True), (False), ('n2')):
        for variant in (True, 'sort'):
            a, b = res[variant, key], res[False, key]
            assert len(a) == len(b) == (2 if key == 'n2' else n)
            for x, y in zip(a, b):
                assert x.shape == y.shape and x.dtype == y.dtype == torch.float32
                assert torch.equal(x, y), (variant, key, int((x != y).any(1).sum()))
    kept = [int((d[:, 4] >= 0).sum()) for d in res[True, True]]
    assert all(0 < k <= 1000 for k in kept)
    if min_size > 0:
        for d in res[True, False]:
            assert ((d[:, 2] - d[:, 0]) > min_size).all() and ((d[:, 3] - d[:, 1]) > min_size).all()


def test_radix_select_topk_equals_stable_sort(dev):
    """oadg_rpn_topk row by row against sigmoid + stable descending torch.sort: scores AND indices, incl. rows shorter than
    nms_pre, heavy ties (saturated / quantised logits), bf16 strided inputs"""
    import ctypes
    from oadg_amd import _lib
    L = _lib.lib()
    g = torch.Generator(device=dev).manual_seed(4)
    for dtype, quant in ((torch.float32, False), (torch.float32, True), (torch.bfloat16, False), (torch.bfloat16, True)):
        dims = [(40, 64, 3), (20, 32, 3), (7, 9, 3), (3, 5, 3)]
        n_img, nms_pre = 3, 500
        cls = []
        for h, w, a in dims:
            t = torch.randn(n_img + 1, 16, h, w, device=dev, generator=g) * (6.0 if quant else 2.0)
            if quant:
                t = (t * 2).round() / 2
            t = t.to(dtype).contiguous(memory_format=torch.channels_last)
            cls.append(t[:, 2:2 + a])                       # a strided channel slice of a wider NHWC tensor
        nl = len(dims)
        ks = [min(nms_pre, h * w * a) for h, w, a in dims]
        sc = [torch.empty((n_img, k), dtype=torch.float32, device=dev) for k in ks]
        ix = [torch.empty((n_img, k), dtype=torch.int64, device=dev) for k in ks]
        level_n = (ctypes.c_int * nl)(*[h * w * a for h, w, a in dims])
        nbytes = L.oadg_rpn_topk_workspace_bytes(level_n, nl, n_img, nms_pre)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        strides = (ctypes.c_long * (4 * nl))(*[int(v) for c in cls for v in c.stride()])
        dims_c = (ctypes.c_int * (3 * nl))(*[v for d in dims for v in d])
        _lib.check(L.oadg_rpn_topk((ctypes.c_void_p * nl)(*[c.data_ptr() for c in cls]), strides, dims_c,
                                   0 if dtype == torch.float32 else 1, nl, n_img, nms_pre,
                                   (ctypes.c_void_p * nl)(*[t.data_ptr() for t in sc]),
                                   (ctypes.c_void_p * nl)(*[t.data_ptr() for t in ix]), _lib.ptr(ws), nbytes,
                                   _lib.stream_ptr()), 'oadg_rpn_topk')
        for l in range(nl):
            ref = cls[l][:n_img].float().permute(0, 2, 3, 1).reshape(n_img, -1).sigmoid()
            rs, ri = ref.sort(dim=1, descending=True, stable=True)
            assert torch.equal(sc[l], rs[:, :ks[l]]), (dtype, quant, l)
            assert torch.equal(ix[l], ri[:, :ks[l]]), (dtype, quant, l, int((ix[l] != ri[:, :ks[l]]).sum()))


def test_fused_proposals_single_level_more_candidates_than_split_thr(dev):
    """the R101-DC5 shape: one level, 15 anchors per cell, nms_pre = 12000 candidates of 17,280 anchors"""
    from oadg_amd import Config
    from oadg_amd.dense_heads import RPNHead
    test_cfg = Config(dict(nms_pre=12000, max_per_img=2000, nms=dict(type='nms', iou_threshold=0.7), min_bbox_size=0))
    head = RPNHead(in_channels=256, feat_channels=256,
                   anchor_generator=dict(type='AnchorGenerator', scales=[2, 4, 8, 16, 32], ratios=[0.5, 1.0, 2.0], strides=[16]),
                   bbox_coder=dict(type='DeltaXYWHBBoxCoder', target_means=[.0, .0, .0, .0], target_stds=[1.0, 1.0, 1.0, 1.0]),
                   loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                   loss_bbox=dict(type='L1Loss', loss_weight=1.0), test_cfg=test_cfg).to(dev)
    g = torch.Generator(device=dev).manual_seed(9)
    cs = [torch.randn(2, 15, 24, 48, device=dev, generator=g)]
    bp = [torch.randn(2, 60, 24, 48, device=dev, generator=g) * 0.3]
    metas = [dict(img_shape=(384, 768, 3))] * 2
    out = {}
    for fused in (True, False):
        RPNHead.FUSED_PROPOSALS = fused
        try:
            out[fused] = head.get_bboxes(cs, bp, img_metas=metas, padded=True)
        finally:
            RPNHead.FUSED_PROPOSALS = True
    for a, b in zip(out[True], out[False]):
        assert a.shape == (2000, 5) and torch.equal(a, b)
