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


def test_saliency_degenerate_crop_with_zero_nyquist_bin(dev):
    """A crop whose 64x64 resized gray image has an alternating sum of exactly 0: the Nyquist bin of the FFT is an
    exact integer 0 on both sides, log(0) = -inf turns the whole map into NaN and uint8(NaN) = 0 -> score 0
    (<= 10: the box becomes a mixing target).  Small boxes of the stress config hit this regularly."""
    from oadg_amd.pipelines.oa_mix import _ImageState
    rs = np.random.RandomState(5)
    img = lowpass_image(rs, 256, 512, 4)
    box = np.array([[173.8381, 45.01254, 205.2561, 53.769485]], np.float32)
    with np.errstate(all='ignore'):
        ref = cv.saliency_score(img[45:53, 173:205])
    assert ref == 0.0
    st = _ImageState(torch.from_numpy(img).to(dev), box, 4, 0.3)
    assert st.scores()[0] == 0.0


def test_oamix_stress_many_small_boxes_bit_exact(dev):
    """BASELINE config 5's box statistics (w, h ~ U(8, 48)) at an oracle-sized problem: 128 boxes on 256x512."""
    from oadg_amd.pipelines import OAMix
    rs = np.random.RandomState(5)
    img, gts = lowpass_image(rs, 256, 512, 4), synthetic_boxes(rs, 128, 256, 512, 8, 48)
    np.random.seed(3)
    oracle = OO.OAMixOracle(version='augmix')
    with np.errstate(all='ignore'):
        r_ref = oracle(dict(img=img.copy(), gt_bboxes=gts.copy()))
    rng_ref = np.random.random()
    fg = [t[1] for t in oracle.trace if t[0] == 'fg_scores'][0]
    assert not any(abs(s - 10) < 0.2 for s in fg if s >= 0)
    np.random.seed(3)
    mix = OAMix(version='augmix')
    mix.trace = []
    r = mix(dict(img=img.copy(), gt_bboxes=gts.copy()))
    assert mix.trace == [t[1] for t in oracle.trace if t[0] == 'op']
    assert np.random.random() == rng_ref
    assert np.array_equal(r['multilevel_boxes'], r_ref['multilevel_boxes'])
    assert np.array_equal(r['oamix_boxes'], r_ref['oamix_boxes'])
    assert np.array_equal(r['img2'], r_ref['img2'])


def test_oamix_stress_config5_full_size_properties(dev):
    """BASELINE config 5 proper: 4096 boxes per 1024x2048 image (the CPU path would need 4096 full-resolution fp32
    masks = 103 GB).  Size-independent checks: the first RNG-only stage (multi-level random boxes) equals the oracle's,
    the run is reproducible byte for byte under the same seed, view 1 is untouched, saliency scores of a sample of the
    boxes match the oracle."""
    from oadg_amd.pipelines import OAMix
    from oadg_amd.pipelines.oa_mix import _ImageState
    H, W, n = 1024, 2048, 4096
    rs = np.random.RandomState(9)
    img, gts = lowpass_image(rs, H, W, 8), synthetic_boxes(rs, n, H, W, 8, 48)
    outs = []
    for _ in range(2):
        np.random.seed(21)
        mix = OAMix(version='augmix')
        mix.trace = []
        r = mix(dict(img=img.copy(), gt_bboxes=gts.copy()))
        # the draw must exercise the per-box path (4096 sequential warps + blends) and the 4096-mask union
        assert any(t.startswith('bboxes_only') for t in mix.trace) and any(t.startswith('bg_only') for t in mix.trace)
        outs.append((r['img2'].copy(), np.asarray(r['multilevel_boxes']).copy(), np.asarray(r['oamix_boxes']).copy(),
                     np.random.random()))
        assert np.array_equal(r['img'], img)
    assert np.array_equal(outs[0][0], outs[1][0]) and outs[0][3] == outs[1][3]
    assert np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])
    assert (outs[0][0] != img).any()
    np.random.seed(21)
    oracle = OO.OAMixOracle(version='augmix')
    np.random.dirichlet([1] * 3)
    boxes, _ = oracle.get_random_regions(img, oracle.random_box_scale, oracle.random_box_ratio, (1, 3))[:2]
    assert np.array_equal(np.concatenate(boxes).reshape(-1, 4), outs[0][1].reshape(-1, 4))
    st = _ImageState(torch.from_numpy(img).to(dev), gts, 4, 0.3)
    got = st.scores()
    for i in range(0, n, 97):
        x1, y1, x2, y2 = np.array(gts[i], dtype=np.int32)
        if x2 - x1 < 4 or y2 - y1 < 4:
            assert got[i] == -1
        else:
            with np.errstate(all='ignore'):
                assert abs(got[i] - cv.saliency_score(img[y1:y2, x1:x2])) <= 0.05


@pytest.mark.parametrize('H,W', [(131, 253), (133, 250), (129, 255)])
def test_bbox_chain_rect_areas_not_multiple_of_4(dev, H, W):
    """Supports clipped at the right / bottom border of an image whose sides are not multiples of 4 have areas with
    area % 4 in {1, 2, 3}: the tail thread of such a rect must not store past its own pixels (the next rect of the level
    starts at the next 4-byte boundary of the packed scratch image).  Level-batched path == per-box path == oracle."""
    from oadg_amd.pipelines import OAMix
    from oadg_amd.pipelines import oa_mix as PM
    rs = np.random.RandomState(H)
    img = lowpass_image(rs, H, W, 4)
    gts = np.array([[W - 23.4, 5, W - 0.5, 31], [W - 21, H - 19.2, W - 1, H - 0.7], [3, H - 26, 30, H - 1.2],
                    [60, 40, 93, 77], [W - 41.5, 60, W - 2.2, 90.5], [120, H - 31, 150.5, H - 0.2],
                    [10, 10, 33, 29], [W // 2, 5, W // 2 + 27, 33]], np.float32)
    outs = []
    prev = PM.BATCH_BOXES
    try:
        for batch in (True, False):
            PM.BATCH_BOXES = batch
            for seed in range(6):
                np.random.seed(400 + seed)
                mix = OAMix(version='augmix.all')
                mix.trace = []
                r = mix(dict(img=img.copy(), gt_bboxes=gts.copy()))
                outs.append((batch, seed, r['img2'].copy(), list(mix.trace)))
    finally:
        PM.BATCH_BOXES = prev
    n_bbox = 0
    for (b0, s0, a, tr), (b1, s1, c, _) in zip(outs[:6], outs[6:]):
        assert s0 == s1 and np.array_equal(a, c), (s0, int((a != c).sum()))
        n_bbox += sum(t.startswith('bboxes_only') for t in tr)
    assert n_bbox > 0
    for _, seed, a, _tr in outs[:6]:
        np.random.seed(400 + seed)
        oracle = OO.OAMixOracle(version='augmix.all')
        ref = oracle(dict(img=img.copy(), gt_bboxes=gts.copy()))
        fg = [t[1] for t in oracle.trace if t[0] == 'fg_scores']
        if fg and any(abs(s - 10) < 0.2 for s in fg[0] if s >= 0):
            continue
        assert np.array_equal(a, ref['img2']), seed


def test_oamix_helper_threads_equal_independent_sequential_workers(dev):
    """DevicePipeline(oamix_workers=2): helper k augments the images k, k + 2, ... on its own HIP stream with its own numpy
    stream (seeded from the caller's stream) - byte for byte what a single-threaded pipeline produces for those images
    with that RandomState installed, like independent DataLoader workers; the batch is complete when the call returns
    its stream's work (both views of every image, boxes lists in image order)."""
    import os
    from oadg_amd import Config
    from oadg_amd.pipelines import DevicePipeline
    from oadg_amd.pipelines.oa_mix import use_random_state
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, 'configs/oadg/faster_rcnn_r50_fpn_1x_cityscapes_oadg.py'))
    cases = [_case(20 + i, 256, 384, 6 + i, small=True) for i in range(5)]
    imgs = torch.from_numpy(np.stack([c[0] for c in cases])).to(dev)
    gts = [c[1] for c in cases]
    labels = [np.zeros(len(g), np.int64) for g in gts]
    np.random.seed(5)
    par = DevicePipeline(cfg.data.train.pipeline, dtype=torch.float32, oamix_workers=2)
    out = par(imgs, gts, labels)
    torch.cuda.synchronize()
    np.random.seed(5)
    seeds = [int(np.random.randint(0, 2 ** 31 - 1)) for _ in range(2)]
    try:
        for k in range(2):
            idx = list(range(k, 5, 2))
            seq = DevicePipeline(cfg.data.train.pipeline, dtype=torch.float32)
            use_random_state(np.random.RandomState(seeds[k]))
            ref = seq(imgs[idx], [gts[i] for i in idx], [labels[i] for i in idx])
            torch.cuda.synchronize()
            for j, i in enumerate(idx):
                assert torch.equal(out['img2'][i], ref['img2'][j]), i
                assert torch.equal(out['img'][i], ref['img'][j]), i
                assert np.array_equal(out['oamix_boxes'][i].numpy(), ref['oamix_boxes'][j].numpy())
                assert np.array_equal(out['multilevel_boxes'][i].numpy(), ref['multilevel_boxes'][j].numpy())
    finally:
        use_random_state(None)
    # a second batch through the same helpers (their streams continue): still one result per image, finite
    out2 = par(imgs, gts, labels)
    torch.cuda.synchronize()
    assert out2['img2'].shape == out['img2'].shape and bool(torch.isfinite(out2['img2']).all())
    assert not torch.equal(out2['img2'], out['img2'])


@pytest.mark.parametrize('plan_in_c,planner_threads', [(True, False), (True, True), (False, False)])
def test_oamix_lockstep_batch_is_byte_identical_to_the_sequential_pass(dev, monkeypatch, plan_in_c, planner_threads):
    """OAMix.oamix_many (round 4): the images of a batch record their device commands and advance their bboxes_only_*
    chains level by level TOGETHER (csrc oadg_oamix_bbox_chain_multi: one launch pair per level for all images).  Against
    the image-by-image pass with the same numpy stream: both views byte-identical, the same box lists, the stream left
    in the same state - for images with many, few and no boxes, dense small boxes (deep chains) and both host planners;
    and the lockstep pass really shares launches (fewer level rounds than the sum of the chains' depths).  Since round 6
    the lockstep pass is dependency-driven (OAMix.execute): the three mixture chains of a view record on buffer sets of
    their own and every per-box chain that is ready - of any image, any mixture chain, any region op - joins the same
    launch pair per level; with ``planner_threads`` every plan call runs on a planner thread (as for images with >= 512
    boxes)."""
    import os
    from oadg_amd import Config
    from oadg_amd.pipelines import DevicePipeline
    from oadg_amd.pipelines import oa_mix
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, 'configs/oadg/faster_rcnn_r50_fpn_1x_cityscapes_oadg.py'))
    monkeypatch.setattr(oa_mix, 'PLAN_IN_C', plan_in_c)
    monkeypatch.setattr(oa_mix, 'ASYNC_PLAN_MIN_BOXES', 1 if planner_threads else 1 << 30)
    cases = [_case(40 + i, 256, 384, n, small=True) for i, n in enumerate((9, 3, 0, 14, 6))]
    rs = np.random.RandomState(9)
    cases.append((lowpass_image(rs, 256, 384, 4), synthetic_boxes(rs, 96, 256, 384, 8, 40)))     # config-5-like density
    imgs = torch.from_numpy(np.stack([c[0] for c in cases])).to(dev)
    gts = [c[1] for c in cases]
    labels = [np.zeros(len(g), np.int64) for g in gts]
    out, stats, rng_after = {}, {}, {}
    for lock in (True, False):
        monkeypatch.setattr(oa_mix, 'LOCKSTEP', lock)
        pipe = DevicePipeline(cfg.data.train.pipeline, dtype=torch.float32)
        pipe.oamix.stats = {}
        np.random.seed(11)
        out[lock] = pipe(imgs, gts, labels)
        torch.cuda.synchronize()
        rng_after[lock] = np.random.random()
        stats[lock] = dict(pipe.oamix.stats)
        # a second batch through the same pipeline object (buffers of the slots are reused)
        out[lock, 2] = pipe(imgs.flip(0), gts[::-1], labels[::-1])
        torch.cuda.synchronize()
    assert rng_after[True] == rng_after[False]
    for key in (True, (True, 2)):
        a, b = out[key], out[False if key is True else (False, 2)]
        assert torch.equal(a['img'], b['img']) and torch.equal(a['img2'], b['img2'])
        for i in range(len(cases)):
            assert np.array_equal(a['oamix_boxes'][i].numpy(), b['oamix_boxes'][i].numpy()), i
            assert np.array_equal(a['multilevel_boxes'][i].numpy(), b['multilevel_boxes'][i].numpy()), i
    assert not torch.equal(out[True]['img2'], out[True]['img'])
    s = stats[True]
    assert s['bbox_ops'] == stats[False]['bbox_ops'] > 0 and s['bbox_levels'] == stats[False]['bbox_levels']
    assert 0 < s['lockstep_rounds'] <= s['bbox_ops']
    assert 0 < s['lockstep_chains'] <= s['bbox_ops']
    assert s['lockstep_rounds'] < s['lockstep_chains']          # chains of one image really shared launches (round 6)
    assert s['lockstep_levels'] < s['bbox_levels'], s          # levels issued (deepest chain per round) < sum of depths
    assert 'lockstep_rounds' not in stats[False]


def test_fg_union_from_support_rects_is_byte_identical(dev, monkeypatch):
    """oadg_oamix_fg_union_rects (images with many boxes: the union of the blurred fg masks built from the masks' support
    rects by integer atomicMax) against oadg_oamix_fg_union (every box at every pixel): the float union and its uint8
    image byte for byte - dense small boxes, boxes at the image border, empty masks."""
    from oadg_amd.pipelines import oa_mix
    H, W = 256, 384
    rs = np.random.RandomState(5)
    gt = synthetic_boxes(rs, 70, H, W, 8, 36).astype(np.float32)
    gt[:4] = [[0, 0, 30, 20], [W - 33, H - 21, W - 1, H - 1], [10, 10, 12, 11], [100, 50, 100, 90]]   # border, tiny, empty
    img = torch.from_numpy(lowpass_image(rs, H, W, 4)).to(dev)
    out = {}
    for mode, thr in (('rects', 1), ('dense', 1 << 30)):
        monkeypatch.setattr(oa_mix, 'UNION_RECTS_MIN_BOXES', thr)
        st = oa_mix._ImageState(img, gt, 4, 0.3)
        torch.cuda.synchronize()
        out[mode] = (st.union_f.clone(), st.union_u8.clone())
    assert torch.equal(out['rects'][0], out['dense'][0])
    assert torch.equal(out['rects'][1], out['dense'][1])
    assert out['dense'][0].max().item() > 0.5 and (out['dense'][0] == 0).any().item()


def test_oamix_many_box_paths_are_byte_identical(dev, monkeypatch):
    """The paths an image with MANY boxes takes (fg-mask union from the support rects, object-aware mixing over tile-binned
    target lists with the targets' weights drawn in one call, plans on planner threads) against the few-box paths on the
    same images with the same numpy stream: both views byte-identical, the stream left in the same state."""
    import os
    from oadg_amd import Config
    from oadg_amd.pipelines import DevicePipeline
    from oadg_amd.pipelines import oa_mix
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, 'configs/oadg/faster_rcnn_r50_fpn_1x_cityscapes_oadg.py'))
    rs = np.random.RandomState(3)
    cases = [(lowpass_image(rs, 256, 384, 4), synthetic_boxes(rs, n, 256, 384, 8, 40)) for n in (150, 40, 260)]
    imgs = torch.from_numpy(np.stack([c[0] for c in cases])).to(dev)
    gts = [c[1] for c in cases]
    labels = [np.zeros(len(g), np.int64) for g in gts]
    out, after, n_targets = {}, {}, {}
    real_final = oa_mix._lib.lib().oadg_oamix_final_tiles
    for many in (True, False):
        thr = 1 if many else 1 << 30
        for name in ('MIX_TILES_MIN_TARGETS', 'UNION_RECTS_MIN_BOXES', 'ASYNC_PLAN_MIN_BOXES'):
            monkeypatch.setattr(oa_mix, name, thr)
        pipe = DevicePipeline(cfg.data.train.pipeline, dtype=torch.float32)
        np.random.seed(21)
        out[many] = pipe(imgs, gts, labels)
        torch.cuda.synchronize()
        after[many] = np.random.random()
    assert after[True] == after[False]
    assert torch.equal(out[True]['img'], out[False]['img']) and torch.equal(out[True]['img2'], out[False]['img2'])
    assert not torch.equal(out[True]['img2'], out[True]['img'])
    for i in range(len(cases)):
        assert np.array_equal(out[True]['oamix_boxes'][i].numpy(), out[False]['oamix_boxes'][i].numpy())


def test_batched_image_states_equal_the_per_image_states(dev, monkeypatch):
    """_ImageState.batch (round 6: the boxes of a whole same-shape batch through ONE profile launch and ONE saliency launch
    triple) against one _ImageState per image: mask profiles, unions, saliency scores bit for bit (an image without boxes
    and a 70-box image take part), and the views of the whole pipeline byte for byte."""
    import os
    from oadg_amd import Config
    from oadg_amd.pipelines import DevicePipeline, device_pipeline
    from oadg_amd.pipelines.oa_mix import _ImageState
    cases = [_case(90 + i, 256, 384, n, small=True) for i, n in enumerate((9, 0, 70, 4))]
    imgs = torch.from_numpy(np.stack([c[0] for c in cases])).to(dev)
    gts = [c[1] for c in cases]
    bs = _ImageState.batch(imgs, gts, 4, 0.3)
    for i, st in enumerate(bs):
        one = _ImageState(imgs[i], gts[i], 4, 0.3)
        n = one.n
        assert st.n == n and st.scores() == one.scores()
        assert torch.equal(st.My[:n], one.My[:n]) and torch.equal(st.Mx[:n], one.Mx[:n])
        assert torch.equal(st.union_f, one.union_f) and torch.equal(st.union_u8, one.union_u8)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = Config.fromfile(os.path.join(root, 'configs/oadg/faster_rcnn_r50_fpn_1x_cityscapes_oadg.py'))
    labels = [np.zeros(len(g), np.int64) for g in gts]
    out = {}
    for flag in (True, False):
        monkeypatch.setattr(device_pipeline, 'BATCH_STATES', flag)
        pipe = DevicePipeline(cfg.data.train.pipeline, dtype=torch.float32)
        np.random.seed(33)
        out[flag] = pipe(imgs, gts, labels)
        torch.cuda.synchronize()
    assert torch.equal(out[True]['img2'], out[False]['img2']) and torch.equal(out[True]['img'], out[False]['img'])
    for i in range(len(cases)):
        assert np.array_equal(out[True]['oamix_boxes'][i].numpy(), out[False]['oamix_boxes'][i].numpy())
