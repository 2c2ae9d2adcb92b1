"""-m gpu: the fused RPN loss (csrc/cls_loss.hip oadg_rpn_loss_fwd / _bwd through AnchorHead._fused_loss) against the
per-level path (permute / cast / reshape + CrossEntropyLossPlus + L1LossPlus per level, anchor_head.py:402-544): same
loss values (1e-5: one fp64 accumulation over all levels instead of five fp32 ones) and the same gradients in the head's
input features and parameters (bf16 rounding of the head-output gradient is the same on both sides)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(dev, fused, lam, seed=0, n_img=2, H=192, W=320):
    from oadg_amd import Config, hip_conv
    from oadg_amd.apis import set_random_seed
    from oadg_amd.dense_heads import AnchorHead
    from oadg_amd.registry import HEADS, build_from_cfg
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'oadg', 'faster_rcnn_r50_fpn_1x_cityscapes_oadg.py'))
    hc = cfg.model.rpn_head.to_dict() if hasattr(cfg.model.rpn_head, 'to_dict') else dict(cfg.model.rpn_head)
    hc['loss_cls'] = dict(hc['loss_cls'], lambda_weight=lam)
    hc.update(train_cfg=cfg.model.train_cfg.rpn, test_cfg=cfg.model.test_cfg.rpn)
    set_random_seed(seed)
    head = build_from_cfg(hc, HEADS).to(dev).to(memory_format=torch.channels_last).train()
    with torch.no_grad():
        for p in head.parameters():
            p.add_(torch.randn_like(p) * 0.05)
    g = torch.Generator(device=dev).manual_seed(seed + 1)
    B = 2 * n_img
    feats = [(torch.randn(B, 256, -(-H // s), -(-W // s), device=dev, generator=g) * 0.5).bfloat16()
             .contiguous(memory_format=torch.channels_last).requires_grad_(True) for s in (4, 8, 16, 32, 64)]
    rs = np.random.RandomState(seed)
    gts = []
    for _ in range(n_img):
        xy = rs.uniform(0, [W - 60, H - 60], (5, 2))
        wh = rs.uniform(16, 120, (5, 2))
        gts.append(torch.tensor(np.concatenate([xy, np.minimum(xy + wh, [W - 1, H - 1])], 1).astype(np.float32), device=dev))
    gts = gts + [t.clone() for t in gts]                     # view 2 carries the same boxes
    metas = [dict(img_shape=(H, W, 3), pad_shape=(H, W, 3)) for _ in range(B)]
    AnchorHead.FUSED_LOSS = fused
    hip_conv.enable(True)
    try:
        set_random_seed(7)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            head.begin_targets((H, W), gts, metas, dev)
            losses, _ = head.forward_train(feats, metas, gts, proposal_cfg=cfg.model.train_cfg.rpn_proposal,
                                           num_proposal_imgs=n_img, padded_proposals=True)
        lc = sum(t.mean() for t in losses['loss_rpn_cls'])
        lb = sum(t.mean() for t in losses['loss_rpn_bbox'])
        (lc + 2.0 * lb).backward()
        torch.cuda.synchronize()
    finally:
        hip_conv.enable(False)
        AnchorHead.FUSED_LOSS = True
    return (float(lc.detach()), float(lb.detach()), [f.grad.float().clone() for f in feats],
            {n: p.grad.float().clone() for n, p in head.named_parameters()}, len(losses['loss_rpn_cls']))


@pytest.mark.parametrize('lam', [0.1, 0.0])
def test_fused_rpn_loss_equals_the_per_level_path(dev, lam):
    a = _run(dev, True, lam)
    b = _run(dev, False, lam)
    assert a[4] == 1 and b[4] == 5                        # one fused value / five per-level values
    assert abs(a[0] - b[0]) <= 1e-5 * abs(b[0]) and abs(a[1] - b[1]) <= 1e-5 * abs(b[1]) + 1e-9, (a[:2], b[:2])
    assert b[0] > 0 and b[1] > 0
    for lvl, (x, y) in enumerate(zip(a[2], b[2])):
        scale = y.abs().max().item()              # (0 for a level without sampled anchors when lambda = 0)
        assert (x - y).abs().max().item() <= 2e-2 * scale + 1e-12, (lvl, (x - y).abs().max().item(), scale)
        assert (x - y).abs().mean().item() <= 2e-3 * y.abs().mean().item() + 1e-12, lvl
    for n in b[3]:
        scale = b[3][n].abs().max().item() + 1e-12
        assert (a[3][n] - b[3][n]).abs().max().item() <= 1e-2 * scale, n


def test_fused_rpn_loss_is_reproducible(dev):
    a, b = _run(dev, True, 0.1, seed=3), _run(dev, True, 0.1, seed=3)
    assert a[:2] == b[:2]
    assert all(torch.equal(x, y) for x, y in zip(a[2], b[2]))


@pytest.mark.parametrize('A', [3, 2])
def test_rpn_loss_kernel_forms_agree_bit_for_bit(dev, A):
    """The vector-load kernels (3 anchors, channel-contiguous bf16 maps: two 16-byte loads per view and pixel, rows stored
    cooperatively) against the generic kernels (any strides: here the same values as NCHW-contiguous maps; any anchor
    count): same losses and the same gradient maps bit for bit, padding channels exactly zero."""
    from oadg_amd import hip_ops
    g = torch.Generator(device=dev).manual_seed(11)
    B, Cy, shapes = 4, 32, [(24, 40), (12, 20), (6, 10), (3, 5)]
    At = sum(h * w for h, w in shapes) * A
    labels = torch.randint(0, 3, (B, At), device=dev, generator=g) - 1                 # -1 ignore, 0 fg, 1 bg
    label_w = (labels >= 0).float()
    pos = (labels == 0)
    bbox_w = pos[..., None].float().expand(B, At, 4).contiguous()
    bbox_t = torch.randn(B, At, 4, device=dev, generator=g)
    targets = (labels, label_w, bbox_t, bbox_w)
    vals = [torch.randn(B, Cy, h, w, device=dev, generator=g).bfloat16() for h, w in shapes]
    out = []
    for nhwc in (True, False):
        ys = [(v.contiguous(memory_format=torch.channels_last) if nhwc else v.contiguous()).clone().requires_grad_(True)
              for v in vals]
        lc, lb, parts = hip_ops.rpn_loss(ys, A, targets, 256.0, 1.0, 0.1, 1.0)
        (lc * 1.5 + lb * 0.5).backward()
        out.append((lc.detach().clone(), lb.detach().clone(), [y.grad.clone() for y in ys]))
    (lc1, lb1, g1), (lc2, lb2, g2) = out
    assert torch.equal(lc1, lc2) and torch.equal(lb1, lb2) and float(lc1) > 0 and float(lb1) > 0
    for x, y in zip(g1, g2):
        assert torch.equal(x, y)
        assert float(x[:, :5 * A].abs().sum()) > 0 and float(x[:, 5 * A:].abs().sum()) == 0.0
