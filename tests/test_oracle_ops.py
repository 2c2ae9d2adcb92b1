"""CPU: internal consistency of the oracle's un-vendored-dependency restatements (RoIAlign vectorised vs
line-by-line, NMS vs brute force, exp_det vs libm)."""
import numpy as np
import torch

from oracle import cvleaves as cv
from oracle import nms as ONMS
from oracle import roi_align as R


def test_roi_align_vectorised_equals_scalar():
    rs = np.random.RandomState(0)
    f = torch.randn(2, 16, 40, 60, requires_grad=True)
    K = 120
    x1 = rs.uniform(-20, 220, K); y1 = rs.uniform(-20, 140, K)
    w = rs.uniform(0, 200, K); h = rs.uniform(0, 150, K)
    rois = torch.tensor(np.stack([rs.randint(0, 2, K), x1, y1, x1 + w, y1 + h], 1).astype(np.float32))
    rois[5] = torch.tensor([0, 5, 5, 5, 5.])
    rois[6] = torch.tensor([1, 30, 30, 10, 10.])
    rois[7] = torch.tensor([0, -500, -500, -400, -400.])
    a = R.roi_align(f, rois, 7, 0.25)
    b = R.roi_align_scalar(f, rois, 7, 0.25)
    assert torch.equal(a, b)
    go = torch.randn_like(a)
    ga, = torch.autograd.grad((a * go).sum(), f)
    gb, = torch.autograd.grad((b * go).sum(), f)
    assert (ga - gb).abs().max() <= 1e-5 * gb.abs().max()


def test_roi_align_constant_map_and_linearity():
    c = torch.full((1, 4, 10, 12), 3.0)
    out = R.roi_align(c, torch.tensor([[0, 4., 4., 20., 18.]]), 7, 0.5)
    assert torch.allclose(out, torch.full_like(out, 3.0))
    a, b = torch.randn(1, 4, 10, 12), torch.randn(1, 4, 10, 12)
    r = torch.tensor([[0, 1., 2., 19., 15.], [0, 0., 0., 5., 5.]])
    assert torch.allclose(R.roi_align(2 * a + b, r, 7, 0.5), 2 * R.roi_align(a, r, 7, 0.5) + R.roi_align(b, r, 7, 0.5),
                          atol=1e-5)


def test_nms_matches_brute_force():
    rs = np.random.RandomState(1)
    c = rs.uniform(0, 100, (200, 2)); s = rs.uniform(5, 40, (200, 2))
    b = np.concatenate([c - s / 2, c + s / 2], 1).astype(np.float32)
    keep = ONMS.nms_sorted(b, 0.5)
    alive, ref = np.ones(200, bool), []
    for i in range(200):
        if not alive[i]:
            continue
        ref.append(i)
        for j in range(i + 1, 200):
            xx1, yy1 = max(b[i, 0], b[j, 0]), max(b[i, 1], b[j, 1])
            xx2, yy2 = min(b[i, 2], b[j, 2]), min(b[i, 3], b[j, 3])
            inter = max(xx2 - xx1, 0) * max(yy2 - yy1, 0)
            a1 = (b[i, 2] - b[i, 0]) * (b[i, 3] - b[i, 1]); a2 = (b[j, 2] - b[j, 0]) * (b[j, 3] - b[j, 1])
            if inter / (a1 + a2 - inter) > 0.5:
                alive[j] = False
    assert keep.tolist() == ref


def test_exp_det_accuracy():
    x = -np.random.RandomState(0).rand(20000) * 700
    assert np.max(np.abs(cv.exp_det(x) - np.exp(x)) / np.exp(x)) < 4e-16
