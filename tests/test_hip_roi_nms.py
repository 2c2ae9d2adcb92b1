"""-m gpu parity: HIP RoIAlign (FPN, fwd+bwd) and batched NMS through the C ABI against oracle/."""
import numpy as np
import pytest
import torch

from oracle import nms as ONMS
from oracle import roi_align as ORA

pytestmark = pytest.mark.gpu


def _rand_rois(rs, n, n_img, W, H, big=False):
    x1 = rs.uniform(-10, W - 8, n); y1 = rs.uniform(-10, H - 8, n)
    w = rs.uniform(0, W if big else W / 4, n); h = rs.uniform(0, H if big else H / 4, n)
    b = rs.randint(0, n_img, n)
    r = np.stack([b, x1, y1, np.minimum(x1 + w, W + 20), np.minimum(y1 + h, H + 20)], 1)
    return r.astype(np.float32)


@pytest.mark.parametrize('C', [8, 256])
def test_roi_align_fpn_fwd_bwd_fp32(dev, C):
    from oadg_amd import hip_ops
    rs = np.random.RandomState(C)
    strides = [4, 8, 16, 32]
    H, W = 128, 256
    feats = [torch.tensor(rs.standard_normal((2, C, H // s, W // s)).astype(np.float32)) for s in strides]
    rois = _rand_rois(rs, 40, 2, W, H, big=True)
    # degenerate / edge RoIs: zero area, inverted, far outside, full image, H/W-swapped random proposal
    extra = np.array([[0, 5, 5, 5, 5], [1, 30, 30, 10, 10], [0, -500, -500, -400, -400],
                      [1, 0, 0, W, H], [0, 100, 47, 256, 293]], np.float32)
    rois = np.concatenate([rois, extra])
    fc = [f.clone().requires_grad_(True) for f in feats]
    oc = ORA.roi_align_fpn(fc, torch.tensor(rois), 7, strides)
    gout = torch.tensor(rs.standard_normal(tuple(oc.shape)).astype(np.float32))
    (oc * gout).sum().backward()
    fg = [f.to(dev).requires_grad_(True) for f in feats]
    og = hip_ops.roi_align_fpn(fg, torch.tensor(rois, device=dev), 7, [1.0 / s for s in strides])
    assert og.shape == oc.shape
    (og * gout.to(dev)).sum().backward()
    err = (og.detach().cpu() - oc.detach()).abs().max().item()
    assert err <= 1e-4 * oc.detach().abs().max().item(), err
    for a, b in zip(fg, fc):
        bg = b.grad if b.grad is not None else torch.zeros_like(b)   # a level no RoI mapped to
        e = (a.grad.cpu() - bg).abs().max().item()
        assert e <= 1e-4 * bg.abs().max().item() + 1e-6, e


@pytest.mark.parametrize('C', [64, 256, 512])
def test_roi_align_forward_by_rows_and_its_per_sample_fallback(dev, C):
    """the footprint-row forward kernel (round 5) against the oracle on one stride-1 level where some RoIs have bins wider
    than its 32-pixel weight tables (those workgroups take the per-sample form), RoIs hanging over every border, tiny RoIs
    (several bins per pixel) and a 512-channel map (two channel chunks on blockIdx.y) - fp32, 1e-4"""
    from oadg_amd import hip_ops
    rs = np.random.RandomState(C)
    H, W = 300, 420
    f = torch.tensor(rs.standard_normal((2, C, H, W)).astype(np.float32))
    rois = np.array([[0, 10, 10, 410, 290],          # 57 x 40 pixel bins: fallback
                     [1, 5, 5, 250, 120],            # 35-pixel bin columns: fallback (columns only)
                     [0, 20, 30, 200, 190],          # 25-pixel bins: table path
                     [1, 100.3, 50.7, 103.1, 52.2],  # tiny: several bins per pixel
                     [0, -40, -30, 60, 50], [1, 380, 260, 470, 340], [0, -3, 100, 2, 180],
                     [1, 50, 60, 50, 60], [0, 419, 299, 419.5, 299.5]], np.float32)
    rois = np.concatenate([rois, _rand_rois(rs, 24, 2, W, H, big=True)])
    oc = ORA.roi_align(f, torch.tensor(rois), 7, 1.0)
    og = hip_ops.roi_align_fpn([f.to(dev)], torch.tensor(rois, device=dev), 7, [1.0])
    err = (og.cpu() - oc).abs().max().item()
    assert err <= 1e-4 * oc.abs().max().item(), err


def test_roi_align_single_level_bf16(dev):
    from oadg_amd import hip_ops
    rs = np.random.RandomState(5)
    f = torch.tensor(rs.standard_normal((1, 64, 32, 48)).astype(np.float32)).bfloat16()
    rois = _rand_rois(rs, 16, 1, 48 * 16, 32 * 16, big=True)
    oc = ORA.roi_align(f.float(), torch.tensor(rois), 7, 1 / 16.)
    og = hip_ops.roi_align_fpn([f.to(dev)], torch.tensor(rois, device=dev), 7, [1 / 16.])
    assert og.dtype == torch.bfloat16
    assert (og.float().cpu() - oc).abs().max().item() <= 2e-2 * oc.abs().max().item()


def test_roi_align_config_size_properties(dev):
    # BASELINE config-2 shape: 8 view-images, P2..P5 of 1024x2048, 4096 RoIs. Size-independent checks:
    # constant maps give constant outputs; output is linear in the feature maps.
    from oadg_amd import hip_ops
    strides = [4, 8, 16, 32]
    g = torch.Generator(device=dev).manual_seed(1)
    shapes = [(8, 256, 1024 // s, 2048 // s) for s in strides]
    rs = np.random.RandomState(0)
    rois = torch.tensor(_rand_rois(rs, 4096, 8, 2048, 1024), device=dev)
    rois[:, 1:] = rois[:, 1:].clamp(min=0)
    rois[:, 3] = rois[:, 3].clamp(max=2047); rois[:, 4] = rois[:, 4].clamp(max=1023)
    a = [torch.randn(s, device=dev, generator=g).contiguous(memory_format=torch.channels_last) for s in shapes]
    ones = [torch.full(s, 2.5, device=dev).contiguous(memory_format=torch.channels_last) for s in shapes]
    sc = [1.0 / s for s in strides]
    oa = hip_ops.roi_align_fpn(a, rois, 7, sc)
    o1 = hip_ops.roi_align_fpn(ones, rois, 7, sc)
    inside = (rois[:, 3] - rois[:, 1] > 1) & (rois[:, 4] - rois[:, 2] > 1)
    assert (o1[inside] - 2.5).abs().max().item() <= 1e-5
    ob = hip_ops.roi_align_fpn([x * 2 + y for x, y in zip(a, ones)], rois, 7, sc)
    assert (ob - (2 * oa + o1)).abs().max().item() <= 1e-4 * ob.abs().max().item()


@pytest.mark.parametrize('M,n_img', [(1, 1), (70, 2), (1000, 3), (9536, 2)])
def test_nms_batched_matches_oracle(dev, M, n_img):
    from oadg_amd import hip_ops
    rs = np.random.RandomState(M)
    boxes = np.zeros((n_img, M, 4), np.float32)
    counts = []
    keeps = []
    for i in range(n_img):
        m = M if i == 0 else max(1, M - 37 * i)
        cx = rs.uniform(0, 2048, m); cy = rs.uniform(0, 1024, m)
        w = rs.uniform(4, 300, m); h = rs.uniform(4, 300, m)
        lvl = rs.randint(0, 5, m)
        b = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1).astype(np.float32)
        if m > 4:
            b[3] = b[1]            # exact duplicates (IoU = 1)
            b[4] = [5, 5, 5, 5]    # zero-area box: IoU is 0/0 = NaN, never suppresses
        b = b + (lvl.astype(np.float32) * np.float32(b.max() + 1))[:, None]
        boxes[i, :m] = b
        counts.append(m)
        keeps.append(ONMS.nms_sorted(b, 0.7, 1000))
    k, kc = hip_ops.nms_sorted_batched(torch.tensor(boxes, device=dev),
                                       torch.tensor(counts, device=dev, dtype=torch.int32), 0.7, 1000)
    k, kc = k.cpu().numpy(), kc.cpu().numpy()
    for i in range(n_img):
        assert kc[i] == len(keeps[i])
        assert np.array_equal(k[i, :kc[i]], keeps[i])


@pytest.mark.parametrize('n_ids', [1, 3])
def test_nms_12000_through_the_split_thr_branch(dev, n_ids):
    """R101-DC5 (BASELINE configs[3]): nms_pre = 12000 >= mmcv's split_thr = 10000, so batched_nms takes its per-id
    loop + global re-sort (configs/_base_/models/faster_rcnn_r50_caffe_dc5.py:75-79).  The product always
    runs ONE offset-coded scan over the stably sorted boxes; both must give the same keep list, in the same order,
    including among exactly tied scores (ascending input index), cut at max_per_img = 2000."""
    from oadg_amd import hip_ops
    M = 12000
    rs = np.random.RandomState(7 + n_ids)
    cx = rs.uniform(0, 1280, M); cy = rs.uniform(0, 736, M)
    w = rs.uniform(8, 400, M); h = rs.uniform(8, 400, M)
    boxes = torch.tensor(np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1).astype(np.float32))
    scores = torch.tensor(rs.uniform(0, 1, M).astype(np.float32))
    scores[rs.randint(0, M, 600)] = 0.99609375            # a tie group, as saturated sigmoids produce
    ids = torch.tensor(rs.randint(0, n_ids, M))
    dets, keep = ONMS.batched_nms(boxes, scores, ids, dict(type='nms', iou_threshold=0.7))
    assert boxes.shape[0] >= 10000                         # i.e. the oracle went through the split branch
    keep = keep[:2000]
    # product formulation (dense_heads.RPNHead.get_bboxes): stable sort, class offset, one scan
    b, s, i = boxes.to(dev), scores.to(dev), ids.to(dev)
    order = s.sort(descending=True, stable=True)[1]
    off = i.to(b) * (b.max() + 1)
    k, kc = hip_ops.nms_sorted_batched((b + off[:, None])[order][None],
                                       torch.tensor([M], device=dev, dtype=torch.int32), 0.7, 2000)
    mine = order[k[0, :int(kc[0])].long()].cpu()
    assert len(mine) == len(keep) == 2000
    assert torch.equal(mine, keep)


@pytest.mark.parametrize('C,K', [(256, 300), (64, 37), (512, 900), (64, 2000)])
def test_roi_align_bf16_backward_by_tiles(dev, C, K, monkeypatch):
    """The bf16 backward organised by output tiles (csrc roi_align_bwd_tiles: no fp32 maps, no atomics) against the
    oracle's fp32 backward on the same bf16-rounded operands (bf16 output rounding: 2^-8 relative) and against the
    atomic path; map sizes that are not multiples of the 8 x 8 tile, degenerate and out-of-image RoIs, empty levels;
    two runs are bit-identical (fixed summation order).  K = 2000: ~670 RoIs per image on level 0 - more than the 512 a
    workgroup tests at once (the tile-by-tile form of the kernel)."""
    from oadg_amd import hip_ops
    rs = np.random.RandomState(C + K)
    strides = [4, 8, 16, 32]
    H, W = 136, 264                                  # 34 x 66, 17 x 33, ... : ragged tiles on every level
    feats = [torch.tensor(rs.standard_normal((3, C, -(-H // s), -(-W // s))).astype(np.float32)).bfloat16() for s in strides]
    rois = _rand_rois(rs, K, 3, W, H, big=(K < 100))
    extra = np.array([[0, 5, 5, 5, 5], [1, 30, 30, 10, 10], [2, -500, -500, -400, -400], [1, 0, 0, W, H],
                      [0, 100, 47, 256, 293]], np.float32)
    rois = np.concatenate([rois, extra])
    fc = [f.float().clone().requires_grad_(True) for f in feats]
    oc = ORA.roi_align_fpn(fc, torch.tensor(rois), 7, strides)
    gout = torch.tensor(rs.standard_normal(tuple(oc.shape)).astype(np.float32)).bfloat16()
    (oc * gout.float()).sum().backward()

    def run(tiles):
        monkeypatch.setattr(hip_ops, 'BWD_TILES', tiles)
        fg = [f.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True) for f in feats]
        og = hip_ops.roi_align_fpn(fg, torch.tensor(rois, device=dev), 7, [1.0 / s for s in strides])
        og.backward(gout.to(dev).contiguous(memory_format=torch.channels_last))
        return [f.grad for f in fg]
    g_tiles, g_tiles2, g_atomic = run(True), run(True), run(False)
    for a, a2, b, ref in zip(g_tiles, g_tiles2, g_atomic, fc):
        rg = ref.grad if ref.grad is not None else torch.zeros_like(ref)
        scale = rg.abs().max().item() + 1e-6
        assert a.dtype == torch.bfloat16 and torch.equal(a, a2)
        assert (a.float().cpu() - rg).abs().max().item() <= 1e-2 * scale
        assert (a.float() - b.float()).abs().max().item() <= 1e-2 * scale


def test_fc_weight_permute_is_the_exact_column_permutation(dev):
    """oadg_fc_weight_permute: bf16(W)[o][p*C + c] = W[o][c*P + p] and the gradient back - both bit-exact"""
    from oadg_amd import hip_ops
    g = torch.Generator(device=dev).manual_seed(0)
    for O, C, P in ((24, 64, 49), (8, 256, 49), (5, 128, 9)):
        w = torch.randn(O, C * P, device=dev, generator=g, requires_grad=True)
        out = hip_ops.fc_weight_permuted(w, C, P)
        ref = w.detach().view(O, C, P).permute(0, 2, 1).reshape(O, P * C).to(torch.bfloat16)
        assert out.dtype == torch.bfloat16 and torch.equal(out, ref)
        go = torch.randn(O, P * C, device=dev, generator=g).to(torch.bfloat16)
        out.backward(go)
        assert torch.equal(w.grad, go.view(O, P, C).permute(0, 2, 1).reshape(O, C * P).float())


def test_bbox_head_on_nhwc_features_matches_the_flatten_path(dev):
    """Shared2FC head under autocast: first FC on the (ph, pw, c) view with permuted weight columns against
    x.flatten(1) with the weight as stored: the same products in another summation order (bf16 GEMM, fp32 accumulate)"""
    from oadg_amd import hip_ops
    from oadg_amd.roi_heads import Shared2FCBBoxHead
    torch.manual_seed(0)
    head = Shared2FCBBoxHead(in_channels=64, fc_out_channels=128, roi_feat_size=7, num_classes=8).to(dev)
    x0 = torch.randn(96, 64, 7, 7, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    res = []
    for on in (True, False):
        hip_ops.FC_PERMUTE = on
        try:
            x = x0.clone().requires_grad_(True)
            head.zero_grad(set_to_none=True)
            with torch.autocast('cuda', dtype=torch.bfloat16):
                cls, reg = head(x)
            (cls.float().square().sum() + reg.float().square().sum()).backward()
            res.append((cls.float(), reg.float(), x.grad.float(), head.shared_fcs[0].weight.grad.clone()))
        finally:
            hip_ops.FC_PERMUTE = True
    for a, b in zip(*res):
        assert a.shape == b.shape
        assert (a - b).abs().max().item() <= 2e-2 * b.abs().max().item() + 1e-6
        assert (a - b).abs().mean().item() <= 4e-3 * b.abs().mean().item() + 1e-7


@pytest.mark.parametrize('K', [1, 300, 4198, 8192])
def test_roi_order_one_launch_equals_stable_sort_of_the_keys(dev, K):
    """oadg_roi_order: order = stable argsort of oadg_roi_order_keys' keys, range = first position of every (level, image)
    group (searchsorted of the group's smallest key) - exact; many RoIs share a cell, so ties are exercised."""
    import ctypes
    from oadg_amd import _lib
    L = _lib.lib()
    g = torch.Generator(device=dev).manual_seed(K)
    n_img, levels = 8, 4
    wh = torch.rand(K, 2, device=dev, generator=g) ** 2 * 600 + 4
    xy = torch.rand(K, 2, device=dev, generator=g) * torch.tensor([2048., 1024.], device=dev)
    rois = torch.cat([torch.randint(0, n_img, (K, 1), device=dev, generator=g).float(), xy, xy + wh], 1).contiguous()
    keys = torch.empty(K, dtype=torch.int64, device=dev)
    _lib.check(L.oadg_roi_order_keys(_lib.ptr(rois), K, n_img, levels, 56.0, _lib.ptr(keys), _lib.stream_ptr()), 'keys')
    order = torch.empty(K, dtype=torch.int32, device=dev)
    rng = torch.empty(n_img * levels + 1, dtype=torch.int32, device=dev)
    _lib.check(L.oadg_roi_order(_lib.ptr(rois), K, n_img, levels, 56.0, _lib.ptr(order), _lib.ptr(rng), _lib.stream_ptr()),
               'order')
    skeys, ref = torch.sort(keys, stable=True)
    assert torch.equal(order.long(), ref)
    bounds = torch.arange(n_img * levels + 1, device=dev, dtype=torch.int64) << 20
    assert torch.equal(rng.long(), torch.searchsorted(skeys, bounds))
    assert int(rng[-1]) == K and (K < 100 or int((skeys[1:] == skeys[:-1]).sum()) > 0)


def test_head_parameters_cast_in_one_pass_give_identical_results(dev):
    """hip_ops.cast_all_bf16 (one multi-tensor cast of the contrastive head's Linear parameters forward, one for their
    gradients backward) against autocast's per-layer casts: the same bf16 values enter the same GEMMs - outputs and
    every parameter gradient bit-identical."""
    from oadg_amd import hip_ops
    from oadg_amd.roi_heads import Shared2FCContrastiveHead
    torch.manual_seed(0)
    head = Shared2FCContrastiveHead(in_channels=64, fc_out_channels=128, roi_feat_size=7, num_classes=8).to(dev)
    x0 = torch.randn(96, 64, 7, 7, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    res = []
    for on in (True, False):
        hip_ops.FC_CAST_ONCE = on
        try:
            x = x0.clone().requires_grad_(True)
            head.zero_grad(set_to_none=True)
            with torch.autocast('cuda', dtype=torch.bfloat16):
                cls, reg, cont = head(x)
            (cls.float().square().sum() + reg.float().square().sum() + cont.float().abs().sum()).backward()
            res.append(([cls, reg, cont, x.grad], {n: p.grad.clone() for n, p in head.named_parameters() if p.grad is not None}))
        finally:
            hip_ops.FC_CAST_ONCE = True
    for a, b in zip(res[0][0], res[1][0]):
        assert a.dtype == b.dtype and torch.equal(a, b)
    assert set(res[0][1]) == set(res[1][1]) and len(res[0][1]) >= 12
    for n in res[0][1]:
        assert res[0][1][n].dtype == torch.float32 and torch.equal(res[0][1][n], res[1][1][n]), n


@pytest.mark.gpu
def test_roi_align_is_bit_stable_beside_matrix_kernels(dev):
    """Round 6 (profiles/r06_packed_fp32_hazard.txt): built with packed fp32 instructions, RoIAlign's forward-by-rows and
    backward-by-tiles kernels returned wrong values in lanes 48-63 of a wave WHENEVER waves issuing matrix instructions shared
    their SIMD - 14,926 of 14,926 launches beside the 128-tile convolution on another stream.  The library is built without
    them (csrc/Makefile); this keeps it that way: 150 forward + backward launches beside that convolution, every result
    bit-identical to the one computed alone."""
    import threading
    import time
    from oadg_amd import hip_conv, hip_ops
    N, C, H, W, K = 4, 256, 384, 768, 2048
    g = torch.Generator().manual_seed(1)
    feats = [(torch.randn(N, C, H // s, W // s, generator=g) * 0.5).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
             .requires_grad_() for s in (4, 8, 16, 32)]
    ctr = torch.rand(24, 2, generator=g) * torch.tensor([W, H])
    pick = torch.randint(0, 24, (K,), generator=g)
    cx, cy = (ctr[pick] + torch.randn(K, 2, generator=g) * 12).unbind(1)
    w = torch.rand(K, generator=g) ** 2 * 160 + 12
    h = torch.rand(K, generator=g) ** 2 * 120 + 12
    rois = torch.stack([torch.randint(0, N, (K,), generator=g).float(), (cx - w / 2).clamp(0, W - 1), (cy - h / 2).clamp(0, H - 1),
                        (cx + w / 2).clamp(1, W), (cy + h / 2).clamp(1, H)], 1).to(dev)
    gout = (torch.randn(K, C, 7, 7, generator=g) * 1e-3).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)

    def once():
        out = hip_ops.roi_align_fpn(feats, rois, 7, (1 / 4, 1 / 8, 1 / 16, 1 / 32))
        grads = torch.autograd.grad(out, feats, gout)
        return [out.detach()] + list(grads)

    ref = [t.clone() for t in once()]
    torch.cuda.synchronize()
    stop = []

    def tenant():
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            cl = torch.channels_last
            x3 = torch.randn(8, 256, 32, 64, device=dev).bfloat16().contiguous(memory_format=cl)
            w3 = (torch.randn(256, 256, 3, 3, device=dev) * 0.02).bfloat16().contiguous(memory_format=cl)
            b3 = torch.randn(256, device=dev)
            while not stop:
                for _ in range(50):
                    hip_conv.conv_forward(x3, w3, b3, None, 1, 1, 1, True, variant=3)       # the 128-tile kernel, 4 workgroups per CU
                st.synchronize()

    th = threading.Thread(target=tenant, daemon=True)
    th.start()
    time.sleep(1.0)
    try:
        bad = 0
        for _ in range(150):
            bad += any(not torch.equal(a, b) for a, b in zip(once(), ref))
    finally:
        stop.append(1)
        th.join(30)
    assert bad == 0, f'{bad} of 150 launches beside the 128-tile convolution differ from the result computed alone'
