"""-m gpu parity: OA-Mix on the device (C ABI of csrc/oamix.hip driven by oadg_amd.pipelines.OAMix) against
oracle/oamix.py + oracle/cvleaves.py.  uint8 images and int boxes must match bit for bit; saliency scores
(fp64 FFT on both sides, different FFT algorithms) to 0.05 absolute."""
import ctypes

import numpy as np
import pytest
import torch

from inputs import lowpass_image, synthetic_boxes
from oracle import cvleaves as cv
from oracle import oamix as OO

pytestmark = pytest.mark.gpu


def _case(seed, H, W, n_gt, small=True):
    rs = np.random.RandomState(seed)
    img = lowpass_image(rs, H, W, 4 if H < 512 else 8)
    gts = synthetic_boxes(rs, n_gt, H, W, 6 if small else 24, min(W // 3, 400))
    if n_gt > 2:
        gts[0, 2] = gts[0, 0] + 2.5        # thinner than spatial_ratio: score -1
        gts[1, :2] = np.floor(gts[1, :2])   # integer corner
    return img, gts


def test_box_profiles_bit_exact(dev):
    from oadg_amd.pipelines.oa_mix import _ImageState
    img, gts = _case(1, 200, 328, 9)
    gts = np.concatenate([gts, np.array([[0, 0, 328, 200], [3, 5, 6, 9], [100, 100, 101.5, 180]], np.float32)])
    st = _ImageState(torch.from_numpy(img).to(dev), gts, 4, 0.3)
    My, Mx = st.My.cpu().numpy(), st.Mx.cpu().numpy()
    uf = st.union_f.cpu().numpy()
    ref_union = None
    for i, g in enumerate(gts):
        my, mx = cv.box_mask_profiles(g, 200, 328, 4, 0.3)
        assert np.array_equal(My[i], my), i
        assert np.array_equal(Mx[i], mx), i
        m = (my[:, None] * mx[None, :]).astype(np.float32)
        ref_union = m if ref_union is None else np.maximum(ref_union, m)
        sup = st.support[i]
        nz = np.argwhere(m != 0)
        if len(nz):
            assert sup is not None
            assert nz[:, 1].min() >= sup[0] and nz[:, 1].max() < sup[0] + sup[2]
            assert nz[:, 0].min() >= sup[1] and nz[:, 0].max() < sup[1] + sup[3]
    assert np.array_equal(uf, ref_union)
    assert np.array_equal(st.union_u8.cpu().numpy(), np.asarray(ref_union * 255, dtype=np.uint8))


def test_saliency_scores_close(dev):
    from oadg_amd.pipelines.oa_mix import _ImageState
    img, gts = _case(2, 256, 512, 8)
    st = _ImageState(torch.from_numpy(img).to(dev), gts, 4, 0.3)
    got = st.scores()
    for g, s in zip(gts, got):
        x1, y1, x2, y2 = np.array(g, dtype=np.int32)
        if x2 - x1 < 4 or y2 - y1 < 4:
            assert s == -1
        else:
            ref = cv.saliency_score(img[y1:y2, x1:x2])
            assert abs(s - ref) <= 0.05, (s, ref)


@pytest.mark.parametrize('seed,version,H,W,n_gt', [(0, 'augmix', 128, 256, 5), (1, 'augmix', 160, 256, 5),
                                                   (2, 'augmix', 192, 320, 7), (3, 'augmix', 128, 256, 0),
                                                   (4, 'augmix', 256, 512, 12), (5, 'augmix', 130, 254, 4),
                                                   (6, 'augmix', 128, 256, 5), (7, 'augmix', 128, 256, 5),
                                                   (0, 'augmix.all', 128, 256, 5), (1, 'augmix.all', 160, 256, 5),
                                                   (2, 'augmix.all', 192, 320, 7), (3, 'augmix.all', 128, 256, 4),
                                                   (4, 'augmix.all', 256, 512, 9), (5, 'augmix.all', 130, 254, 4),
                                                   (8, 'augmix.all', 128, 256, 5), (9, 'augmix.all', 128, 256, 5)])
def test_oamix_view_bit_exact(dev, seed, version, H, W, n_gt):
    from oadg_amd.pipelines import OAMix
    img, gts = _case(seed, H, W, n_gt)
    r_ref = dict(img=img.copy(), gt_bboxes=gts.copy())
    np.random.seed(1000 + seed)
    oracle = OO.OAMixOracle(version=version)
    r_ref = oracle(r_ref)
    rng_ref = np.random.random()
    fg_scores = [t[1] for t in oracle.trace if t[0] == 'fg_scores']
    if fg_scores and any(abs(s - 10) < 0.2 for s in fg_scores[0] if s >= 0):
        pytest.skip('a saliency score sits on the decision threshold')
    r = dict(img=img.copy(), gt_bboxes=gts.copy())
    np.random.seed(1000 + seed)
    mix = OAMix(version=version)
    mix.trace = []
    r = mix(r)
    rng = np.random.random()
    assert mix.trace == [t[1] for t in oracle.trace if t[0] == 'op']
    assert rng == rng_ref, 'the global numpy stream was consumed differently'
    assert np.array_equal(r['multilevel_boxes'], r_ref['multilevel_boxes'])
    assert np.array_equal(r['oamix_boxes'], r_ref['oamix_boxes'])
    assert r['img_fields'] == r_ref['img_fields'] and r['custom_field'] == r_ref['custom_field']
    diff = np.abs(r['img2'].astype(int) - r_ref['img2'].astype(int))
    assert diff.max() == 0, (int(diff.max()), int((diff > 0).sum()))
    assert np.array_equal(r['img'], img)


def test_oamix_full_size_and_device_pipeline(dev):
    """BASELINE shape: 1024x2048, 20 boxes.  The device pipeline's normalised view 2 must equal Normalize(Pad)
    of the oracle's uint8 view, and view 1 the normalised original."""
    from oadg_amd import Config
    from oadg_amd.pipelines import DevicePipeline
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, 'configs/oadg/faster_rcnn_r50_fpn_1x_cityscapes_oadg.py'))
    img, gts = _case(11, 1024, 2048, 20, small=False)
    np.random.seed(77)
    oracle = OO.OAMixOracle(version='augmix')
    ref = oracle(dict(img=img.copy(), gt_bboxes=gts.copy()))
    fg = [t[1] for t in oracle.trace if t[0] == 'fg_scores'][0]
    if any(abs(s - 10) < 0.2 for s in fg if s >= 0):
        pytest.skip('a saliency score sits on the decision threshold')
    pipe = DevicePipeline(cfg.data.train.pipeline, dtype=torch.float32)
    np.random.seed(77)
    out = pipe(torch.from_numpy(img[None]).to(dev), [gts], [np.zeros(len(gts), np.int64)])
    mean = np.array([123.675, 116.28, 103.53], np.float32)
    stdinv = (1.0 / np.array([58.395, 57.12, 57.375], np.float64)).astype(np.float32)

    def norm(u8):
        return ((u8[..., ::-1].astype(np.float32) - mean) * stdinv).transpose(2, 0, 1)
    assert np.array_equal(out['img'][0].cpu().numpy(), norm(img))
    assert np.array_equal(out['img2'][0].cpu().numpy(), norm(ref['img2']))
    assert np.array_equal(out['oamix_boxes'][0].numpy(), ref['oamix_boxes'])
    assert np.array_equal(out['multilevel_boxes'][0].numpy(), ref['multilevel_boxes'])
    assert out['img'].is_contiguous(memory_format=torch.channels_last)
