"""CPU: the OA-Mix oracle (oracle/oamix.py) against the fixture produced by the GENUINE reference OAMix
(tests/golden/make_golden_oamix.py): augmented view, box lists and the state of the global numpy stream must be
identical, for both op lists ('augmix' and the DWD config's 'augmix.all')."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
from inputs import lowpass_image, synthetic_boxes  # noqa: E402


def _case_inputs(seed, H, W, n_gt):
    rs = np.random.RandomState(1000 + seed)
    return lowpass_image(rs, H, W), synthetic_boxes(rs, n_gt, H, W, 10, min(H, W) // 2)


@pytest.mark.parametrize('idx', range(8))
def test_oamix_oracle_reproduces_reference_fixture(golden_dir, idx):
    from oracle.oamix import OAMixOracle
    g = np.load(os.path.join(golden_dir, 'oamix_reference.npz'))
    seed, ver, H, W, n_gt = [int(v) for v in g['cases'][idx]]
    img, gts = _case_inputs(seed, H, W, n_gt)
    t = OAMixOracle(version='augmix.all' if ver else 'augmix')
    np.random.seed(seed)
    r = t(dict(img=img.copy(), gt_bboxes=gts.copy()))
    tag = f's{seed}'
    assert np.array_equal(np.asarray(r['multilevel_boxes'], dtype=np.int64), g[tag + '_multilevel_boxes'])
    assert np.array_equal(np.asarray(r['oamix_boxes'], dtype=np.int64), g[tag + '_oamix_boxes'])
    assert np.array_equal(np.asarray(r['gt_bboxes2'], dtype=np.float32), g[tag + '_gt_bboxes2'])
    assert np.array_equal(np.asarray(r['img']), g[tag + '_img'])
    assert np.array_equal(np.asarray(r['img2']), g[tag + '_img2'])
    assert np.random.uniform() == float(g[tag + '_rng_after'][0])
