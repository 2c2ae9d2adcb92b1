"""RandomSampler of the RoI head on the device (csrc/roi_sampler.hip, oadg_amd.device_rng): the same indices and the same
CPU-generator state as the reference's host-side draws (mmdet/core/bbox/samplers/random_sampler.py:32-82 on
torch.randperm), and a whole training step that is bit-identical with and without it."""
import ctypes
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, 'configs', 'oadg', 'faster_rcnn_r50_fpn_1x_cityscapes_oadg.py')


@pytest.fixture
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs the MI355X')
    return torch.device('cuda:0')


def _reference_sample(gt_inds, num, pos_fraction, neg_pos_ub=-1):
    """random_sampler.py / base_sampler.py on the host, drawing from torch's global CPU generator"""
    num_pos = int(num * pos_fraction)
    pos = torch.nonzero(gt_inds > 0, as_tuple=False).squeeze(1)
    if pos.numel() > num_pos:
        pos = pos[torch.randperm(pos.numel())[:num_pos]]
    pos = pos.unique()
    num_neg = num - pos.numel()
    if neg_pos_ub >= 0:
        num_neg = min(num_neg, int(neg_pos_ub * max(1, pos.numel())))
    neg = torch.nonzero(gt_inds == 0, as_tuple=False).squeeze(1)
    if neg.numel() > num_neg:
        neg = neg[torch.randperm(neg.numel())[:num_neg]]
    neg = neg.unique()
    return pos, neg


def _device_sample(gts, num, pos_fraction, neg_pos_ub, dev):
    from oadg_amd import _lib, device_rng
    L = _lib.lib()
    B = len(gts)
    images = (_lib.RoiSampleImage * B)()
    for i, g in enumerate(gts):
        images[i].gt_inds, images[i].n = g.data_ptr(), g.numel()
    sel = torch.full((B, num), -7, dtype=torch.long, device=dev)
    meta = torch.full((3 * B,), -1, dtype=torch.int32, device=dev)
    gen = device_rng.generator(dev)
    state, state_out = gen.upload()
    _lib.check(L.oadg_roi_sample_device(ctypes.cast(images, ctypes.c_void_p), B, num, int(num * pos_fraction),
                                        float(neg_pos_ub), _lib.ptr(state), _lib.ptr(state_out), _lib.ptr(sel), _lib.ptr(meta[:2 * B]),
                                        _lib.ptr(meta[2 * B:]), _lib.stream_ptr()), 'oadg_roi_sample_device')
    gen.download_async()
    gen.sync_host()
    m = meta.cpu().numpy()
    return sel.cpu(), m[:2 * B].reshape(B, 2), m[2 * B:]


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['config2', 'many_positives', 'short_image', 'neg_pos_ub', 'reload_boundary', 'eight_images'])
def test_device_roi_sampler_matches_the_host_draws(dev, case):
    """indices, counts, flags and the generator state afterwards, for: the benchmark's shape (4 x 1000 proposals + gts, few
    positives), more positives than num * pos_fraction (a positive permutation is drawn too), an image with fewer
    candidates than num (flag), a negative/positive bound, an engine state a few draws before a reload, eight images"""
    rs = np.random.RandomState({'config2': 0, 'many_positives': 1, 'short_image': 2, 'neg_pos_ub': 3, 'reload_boundary': 4,
                                'eight_images': 5}[case])
    num, frac, ub = 512, 0.25, -1
    B = 8 if case == 'eight_images' else 4
    gts = []
    for b in range(B):
        n = int(rs.randint(900, 1100))
        g = np.zeros(n, np.int64)
        npos = int(rs.randint(5, 100))
        if case == 'many_positives' and b % 2 == 0:
            npos = int(rs.randint(200, 400))
        idx = rs.permutation(n)
        g[idx[:npos]] = rs.randint(1, 20, npos)
        g[idx[npos:npos + int(rs.randint(0, 60))]] = -1            # ignored rows
        if case == 'short_image' and b == 1:
            g[idx[npos + 300:]] = -1                               # ~300 negatives only: fewer than 512 rows in all
        gts.append(g)
    if case == 'neg_pos_ub':
        ub = 3
    torch.manual_seed(1234)
    if case == 'reload_boundary':
        torch.randperm(624 * 3 - 200)      # leaves the engine a couple of hundred draws before its next reload
    start = torch.get_rng_state()
    ref = [_reference_sample(torch.from_numpy(g), num, frac, ub) for g in gts]
    ref_state, ref_next = torch.get_rng_state(), torch.rand(3)
    torch.set_rng_state(start)
    sel, counts, flags = _device_sample([torch.from_numpy(g).to(dev) for g in gts], num, frac, ub, dev)
    for b, (pos, neg) in enumerate(ref):
        kp, kn = int(counts[b, 0]), int(counts[b, 1])
        assert (kp, kn) == (pos.numel(), neg.numel()), (b, kp, kn, pos.numel(), neg.numel())
        assert torch.equal(sel[b, :kp], pos) and torch.equal(sel[b, kp:kp + kn], neg), b
        assert int(flags[b]) == int(kp + kn != num)
    assert (case == 'short_image') == bool(flags.any()) or case == 'neg_pos_ub'
    assert torch.equal(torch.get_rng_state(), ref_state)          # the engine consumed exactly the reference's draws
    assert torch.equal(torch.rand(3), ref_next)


@pytest.mark.gpu
def test_device_roi_sampler_refuses_images_beyond_its_lds_arrays(dev):
    from oadg_amd import _lib
    n = _lib.lib().oadg_roi_sample_max_rows() + 1
    g = torch.zeros(n, dtype=torch.long, device=dev)
    torch.manual_seed(0)
    before = torch.get_rng_state()
    sel, counts, flags = _device_sample([g], 512, 0.25, -1, dev)
    assert int(flags[0]) == 3 and counts.sum() == 0
    assert torch.equal(torch.get_rng_state(), before)             # nothing drawn: the host path takes over from here


def _engine(dev):
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from test_model_parity import build_and_load
    from oadg_amd import Config
    from oadg_amd.apis import TrainEngine, build_optimizer, set_random_seed
    cfg = Config.fromfile(CFG)
    set_random_seed(0)
    det = build_and_load(dev).to(memory_format=torch.channels_last).train()
    return cfg, det, TrainEngine(det, build_optimizer(det, cfg.optimizer), amp_dtype=torch.bfloat16)


def _steps(dev, eng, det, batches, device_sampler, n_steps=2):
    from oadg_amd.apis import set_random_seed
    eng.speculative_sampling = device_sampler
    set_random_seed(11)
    out = []
    for i in range(n_steps):
        data = {k: (list(v) if isinstance(v, list) else v) for k, v in batches[i].items()}
        r = eng.step(data)
        torch.cuda.synchronize()
        out.append(({k: float(v) for k, v in r['log_vars'].items()}, det.roi_head.bbox_targets[0].clone(),
                    det.roi_head._last_rois[0].clone()))
    return out, torch.get_rng_state().clone(), np.random.get_state()[1].copy(), \
        [p.detach().flatten()[:32].clone() for p in det.parameters() if p.requires_grad]


@pytest.mark.gpu
def test_training_steps_are_bit_identical_with_the_device_sampler(dev):
    """two optimizer steps of the benchmarked configuration at 512 x 1024 (bs 2 x 2 views, bf16, csrc convolutions) with the
    RoI sampler on the device and on the host: the same sampled rows, labels, losses, updated parameters, and the same
    torch / numpy generator states afterwards - and the device run never repeated a step"""
    import copy
    from oadg_amd import hip_conv
    from oadg_amd.apis import set_random_seed
    from oadg_amd.pipelines import DevicePipeline, SyntheticCityscapes
    cfg, det, eng = _engine(dev)
    try:
        ds = SyntheticCityscapes(img_shape=(512, 1024), num_boxes=12, num_classes=8, box_size=(16, 200), seed=0, device=dev)
        pipe = DevicePipeline(cfg.data.train.pipeline, dtype=torch.bfloat16)
        set_random_seed(5)
        batches = [pipe(*ds.batch([2 * i, 2 * i + 1])) for i in range(2)]
        state0 = copy.deepcopy(det.state_dict())
        opt0 = copy.deepcopy(eng.optimizer.state_dict())
        runs = {}
        for mode in (True, False):
            det.load_state_dict(state0)
            eng.optimizer.load_state_dict(copy.deepcopy(opt0))
            hip_conv.refresh_prepared()
            runs[mode] = _steps(dev, eng, det, batches, mode)
        assert eng.respeculated == 0
        (a, ta, na, pa), (b, tb, nb, pb) = runs[True], runs[False]
        for (lva, laba, roia), (lvb, labb, roib) in zip(a, b):
            assert lva == lvb, (lva, lvb)
            assert torch.equal(laba, labb) and torch.equal(roia, roib)
        assert torch.equal(ta, tb) and np.array_equal(na, nb)
        for x, y in zip(pa, pb):
            assert torch.equal(x, y)
    finally:
        hip_conv.enable(False)


@pytest.mark.gpu
def test_a_short_image_repeats_the_step_on_the_host_path(dev, monkeypatch):
    """rpn_proposal.max_per_img = 300: every image has fewer candidates than the sampler's 512 rows, so the device sampler
    raises its flag and TrainEngine repeats the step through the host path - the result equals a run with the device
    sampler switched off, generator states included"""
    import copy
    from oadg_amd import hip_conv
    from oadg_amd.apis import set_random_seed
    from oadg_amd.pipelines import DevicePipeline, SyntheticCityscapes
    cfg, det, eng = _engine(dev)
    try:
        det.train_cfg.rpn_proposal['max_per_img'] = 300
        ds = SyntheticCityscapes(img_shape=(384, 768), num_boxes=8, num_classes=8, box_size=(16, 160), seed=3, device=dev)
        pipe = DevicePipeline(cfg.data.train.pipeline, dtype=torch.bfloat16)
        set_random_seed(5)
        batches = [pipe(*ds.batch([0, 1]))]
        state0 = copy.deepcopy(det.state_dict())
        opt0 = copy.deepcopy(eng.optimizer.state_dict())
        runs = {}
        for mode in (True, False):
            det.load_state_dict(state0)
            eng.optimizer.load_state_dict(copy.deepcopy(opt0))
            hip_conv.refresh_prepared()
            runs[mode] = _steps(dev, eng, det, batches, mode, n_steps=1)
        assert eng.respeculated == 1
        (a, ta, na, pa), (b, tb, nb, pb) = runs[True], runs[False]
        assert a[0][0] == b[0][0] and torch.equal(a[0][1], b[0][1]) and torch.equal(a[0][2], b[0][2])
        assert a[0][1].shape[0] < 2 * 2 * 512
        assert torch.equal(ta, tb) and np.array_equal(na, nb)
        for x, y in zip(pa, pb):
            assert torch.equal(x, y)
    finally:
        hip_conv.enable(False)


@pytest.mark.gpu
def test_shared_rpn_convolution_weight_gradient_as_one_group(dev, monkeypatch):
    """hip_conv.SHARED_GROUP (round 5): the weight gradients of the RPN convolution on P3 ... P6 - one weight, one use per
    pyramid level - run as ONE grouped launch whose consumer sums all their split partials, instead of four single-job
    launches added by autograd.  Same gradient to fp32 rounding (another summation order), same training step."""
    import copy
    from oadg_amd import hip_conv
    from oadg_amd.apis import set_random_seed
    from oadg_amd.pipelines import DevicePipeline, SyntheticCityscapes
    cfg, det, eng = _engine(dev)
    try:
        ds = SyntheticCityscapes(img_shape=(512, 1024), num_boxes=12, num_classes=8, box_size=(16, 200), seed=0, device=dev)
        pipe = DevicePipeline(cfg.data.train.pipeline, dtype=torch.bfloat16)
        set_random_seed(5)
        batch = pipe(*ds.batch([0, 1]))
        state0 = copy.deepcopy(det.state_dict())
        opt0 = copy.deepcopy(eng.optimizer.state_dict())
        grads, groups = {}, {}
        orig = hip_conv.wgrad_multi
        for mode in (True, False):
            det.load_state_dict(state0)
            eng.optimizer.load_state_dict(copy.deepcopy(opt0))
            hip_conv.refresh_prepared()
            monkeypatch.setattr(hip_conv, 'SHARED_GROUP', mode)
            sizes = []
            monkeypatch.setattr(hip_conv, 'wgrad_multi', lambda jobs, *a, **k: (sizes.append(len(jobs)), orig(jobs, *a, **k))[1])
            set_random_seed(11)
            captured = {}
            h = det.rpn_head.rpn_conv.weight.register_hook(lambda g: None)      # (a tensor hook would switch deferral off:
            h.remove()                                                          #  make sure none is left)
            eng.step({k: (list(v) if isinstance(v, list) else v) for k, v in batch.items()})
            torch.cuda.synchronize()
            grads[mode] = {n: p.grad.detach().float().clone() for n, p in det.rpn_head.named_parameters()}
            grads[mode]['layer3'] = det.backbone.layer3[0].conv1.weight.grad.detach().float().clone()
            groups[mode] = sizes
        assert len(groups[True]) < len(groups[False]), (groups[True], groups[False])       # fewer grouped launches
        assert sum(groups[True]) == sum(groups[False])                                      # the same jobs
        for n in grads[True]:
            a, b = grads[True][n], grads[False][n]
            assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item() + 1e-9, n
    finally:
        hip_conv.enable(False)
