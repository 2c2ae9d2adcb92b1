"""CPU: the host logic in oa-dg_amd/core and detectors.generate_random_bboxes_xy against vectors produced by the
reference's own functions (tests/golden/make_golden_core.py): random-proposal boxes and numpy stream state,
anchors, IoU, MaxIoU assignment, RandomSampler indices + torch generator state, box coder (with the fork's
zero-size guard)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
from inputs import synthetic_boxes  # noqa: E402


@pytest.fixture(scope='module')
def g(golden_dir):
    return np.load(os.path.join(golden_dir, 'core_reference.npz'))


@pytest.mark.parametrize('seed', range(4))
def test_generate_random_bboxes_xy(g, seed):
    from oadg_amd.detectors import generate_random_bboxes_xy
    np.random.seed(seed)
    b = generate_random_bboxes_xy((256, 512), num_bboxes=10, bboxes_xy=g[f'rand_gts{seed}'], scales=(0.01, 0.3),
                                  ratios=(0.3, 1 / 0.3), iou_max=0.7, iou_min=0.0)
    assert np.array_equal(np.asarray(b), g[f'rand_boxes{seed}'])
    assert np.random.uniform() == float(g[f'rand_rng{seed}'][0])


def test_anchor_generator(g):
    from oadg_amd.core.anchor import AnchorGenerator
    ag = AnchorGenerator(strides=[4, 8, 16, 32, 64], ratios=[0.5, 1.0, 2.0], scales=[8])
    for i, a in enumerate(ag.grid_priors([(6, 9), (3, 5), (2, 3), (1, 2), (1, 1)], device='cpu')):
        assert np.array_equal(a.numpy(), g[f'anchors{i}'])


def _inputs(seed, n_prop=1000, n_gt=5, W=320, H=200):
    rs = np.random.RandomState(500 + seed)
    gts = synthetic_boxes(rs, n_gt, H, W, 12, 120)
    props = synthetic_boxes(rs, n_prop, H, W, 6, 150)
    k = min(n_gt, n_prop // 4)
    props[:k] = gts[:k] + rs.uniform(-3, 3, (k, 4)).astype(np.float32)
    props[k:2 * k] = gts[:k]
    return props.astype(np.float32), gts.astype(np.float32), rs.randint(0, 8, n_gt).astype(np.int64)


@pytest.mark.parametrize('seed', range(3))
def test_iou_assign_sample(g, seed):
    from oadg_amd.core.bbox import MaxIoUAssigner, RandomSampler, bbox_overlaps
    props, gts, labels = _inputs(seed)
    P, G, L = torch.from_numpy(props), torch.from_numpy(gts), torch.from_numpy(labels)
    assert np.array_equal(bbox_overlaps(G, P).numpy(), g[f'iou{seed}'])
    cfgs = [dict(pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3, match_low_quality=True, ignore_iof_thr=-1),
            dict(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, match_low_quality=False, ignore_iof_thr=-1)]
    samplers = [dict(num=64, pos_fraction=0.5, neg_pos_ub=-1, add_gt_as_proposals=False),
                dict(num=128, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=True)]
    for c, (acfg, scfg) in enumerate(zip(cfgs, samplers)):
        ar = MaxIoUAssigner(**acfg).assign(P, G, None, L)
        assert np.array_equal(ar.gt_inds.numpy(), g[f'assign{seed}_{c}_gt_inds'])
        assert np.array_equal(ar.max_overlaps.numpy(), g[f'assign{seed}_{c}_max_overlaps'])
        assert np.array_equal(ar.labels.numpy(), g[f'assign{seed}_{c}_labels'])
        torch.manual_seed(seed)
        sr = RandomSampler(**scfg).sample(ar, P, G, L)
        assert np.array_equal(sr.pos_inds.numpy(), g[f'sample{seed}_{c}_pos'])
        assert np.array_equal(sr.neg_inds.numpy(), g[f'sample{seed}_{c}_neg'])
        assert np.array_equal(torch.rand(1).numpy(), g[f'sample{seed}_{c}_rng'])


@pytest.mark.parametrize('seed', range(3))
def test_box_coder(g, seed):
    from oadg_amd.core.bbox import bbox2delta, delta2bbox
    p, gt = torch.from_numpy(g[f'delta{seed}_p']), torch.from_numpy(g[f'delta{seed}_g'])
    d = bbox2delta(p, gt, (0., 0., 0., 0.), (0.1, 0.1, 0.2, 0.2))
    assert np.array_equal(d.numpy(), g[f'delta{seed}_d'])
    b = delta2bbox(p, torch.from_numpy(g[f'decode{seed}_d']), (0., 0., 0., 0.), (1., 1., 1., 1.),
                   max_shape=(200, 320), wh_ratio_clip=16 / 1000)
    assert np.array_equal(b.numpy(), g[f'decode{seed}_b'])
