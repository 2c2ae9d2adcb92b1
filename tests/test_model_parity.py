"""Whole-detector parity against tests/golden/model_step_256x512.npz, which was produced by running the
REFERENCE FasterRCNN+ContrastiveRoIHead train step (tests/golden/make_golden_model.py).

* CPU (not gpu): our host logic (anchors, assign, sample, targets, random proposals, list plumbing, module
  layout / state_dict names) with the four HIP entry points swapped for the oracle -> must reproduce the
  reference's losses, sampled labels and gradient norms essentially exactly.
* GPU: the product path (HIP losses / RoIAlign / NMS, torch fp32 convs) on the same inputs.
"""
import os

import numpy as np
import pytest
import torch

from inputs import DC5_SCALE, model_batch, named_weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, 'configs', 'oadg', 'faster_rcnn_r50_fpn_1x_cityscapes_oadg.py')


DC5_CFG = os.path.join(ROOT, 'configs', 'oadg', 'faster_rcnn_r101_dc5_1x_dwd_oadg.py')


def build_and_load(device='cpu', cfg=CFG):
    from oadg_amd import Config, build_detector
    det = build_detector(Config.fromfile(cfg).model)
    w = named_weights({k: v.shape for k, v in det.state_dict().items()}, DC5_SCALE if cfg == DC5_CFG else None)
    det.load_state_dict({k: torch.as_tensor(v) for k, v in w.items()})
    return det.to(device).train()


def make_data(g, device='cpu'):
    kw = dict(n_gt=int(g['n_gt']), n_cls=int(g['n_cls'])) if 'n_gt' in g.files else {}
    b = model_batch(int(g['seed']), int(g['n_img']), int(g['h']), int(g['w']), **kw)
    shape = b['img'].shape[2:] + (3,)
    t = lambda x: torch.tensor(x, device=device)  # noqa: E731
    return dict(img=t(b['img']), img2=t(b['img2']), gt_bboxes=[t(x) for x in b['gt_bboxes']],
                gt_bboxes2=[t(x) for x in b['gt_bboxes']], gt_labels=[t(x) for x in b['gt_labels']],
                multilevel_boxes=[torch.tensor(x) for x in b['multilevel_boxes']],
                oamix_boxes=[torch.tensor(x) for x in b['oamix_boxes']],
                img_metas=[dict(img_shape=shape, pad_shape=shape, ori_shape=shape, scale_factor=1.0, flip=False)
                           for _ in range(b['img'].shape[0])])


def grad_groups(det):
    groups = {}
    for n, p in det.named_parameters():
        if p.grad is not None:
            top = '.'.join(n.split('.')[:2])
            groups[top] = groups.get(top, 0.0) + float(p.grad.double().pow(2).sum())
    return {k: np.sqrt(v) for k, v in groups.items()}


def test_state_dict_layout_and_param_counts(golden_dir):
    g = np.load(os.path.join(golden_dir, 'model_step_256x512.npz'))
    det = build_and_load()
    assert sum(p.numel() for p in det.parameters()) == int(g['n_params'])
    assert sum(p.numel() for p in det.parameters() if p.requires_grad) == int(g['n_trainable']) == 41486904
    keys = set(det.state_dict())
    for k in ['backbone.layer2.0.conv1.weight', 'backbone.layer1.0.downsample.1.running_var',
              'neck.lateral_convs.0.conv.weight', 'neck.fpn_convs.3.conv.bias', 'rpn_head.rpn_conv.weight',
              'rpn_head.rpn_cls.bias', 'roi_head.bbox_head.shared_fcs.0.weight',
              'roi_head.bbox_head.fc_cont.0.weight', 'roi_head.bbox_head.fc_cont.2.bias',
              'roi_head.bbox_head.fc_cls.weight', 'roi_head.bbox_head.fc_reg.bias']:
        assert k in keys, k


@pytest.mark.parametrize('fold_bn', [False, True])
def test_host_logic_reproduces_reference_step(golden_dir, fold_bn, monkeypatch):
    """fold_bn=False: operation-for-operation the reference's network -> agreement to 1e-6.
    fold_bn=True (the default): eval-mode BN folded into the convolutions, same function, fp32 rounding
    differs -> 2e-3 on the RoI terms (1e-7 on the RPN terms), and the sampled labels must still be identical."""
    from oracle.backend import oracle_ops
    from oadg_amd import layers
    monkeypatch.setattr(layers, 'FOLD_EVAL_BN', fold_bn)
    # folding changes fp32 rounding: the RPN terms stay at 1e-7, but the order of two near-tied proposals flips and
    # with it one sampled negative (loss_cls 7e-4, loss_cont 7e-5; measured).  Sampled labels must still be identical.
    tol_l, tol_g = (2e-3, 2e-3) if fold_bn else (1e-6, 1e-5)
    g = np.load(os.path.join(golden_dir, 'model_step_256x512.npz'))
    det = build_and_load()
    data = make_data(g)
    torch.manual_seed(int(g['seed']))
    np.random.seed(int(g['seed']))
    with oracle_ops():
        out = det.train_step(data, None)
        out['loss'].backward()
    for k, v in out['log_vars'].items():
        ref = float(g['lv_' + k])
        assert abs(v - ref) <= tol_l * abs(ref), (k, v, ref)
    assert np.array_equal(det.roi_head.bbox_targets[0].numpy(), g['roi_labels'])
    for k, v in grad_groups(det).items():
        ref = float(g['gn_' + k])
        assert abs(v - ref) <= tol_g * ref, (k, v, ref)
    params = dict(det.named_parameters())
    for k in g.files:
        if k.startswith('g_'):
            mine = params[k[2:]].grad.flatten()[:4096].numpy()
            assert np.abs(mine - g[k]).max() <= (1e-2 if fold_bn else 1e-5) * np.abs(g[k]).max() + 1e-9, k


@pytest.mark.gpu
def test_product_step_matches_reference_on_gpu(dev, golden_dir, monkeypatch):
    """fp32 on the MI355X.  Convolutions come from a different library (MIOpen vs the CPU's oneDNN), and the
    step contains discrete decisions (top-k, NMS, IoU thresholds, sampling): measured (round 3) 1e-7 on the RPN
    terms, 3.3e-4 on loss_cls, 9e-5 on loss_cont, 2e-3 on acc; asserted at 1e-3 (acc 1e-2) and 1e-3 for the
    gradient norms - about 3x the measured deviations; the 1e-4 bar is held by the per-kernel tests, which feed
    identical inputs to both sides."""
    g = np.load(os.path.join(golden_dir, 'model_step_256x512.npz'))
    torch.backends.cudnn.allow_tf32 = False
    # MIOpen's default fp32 solvers reduce split-K partials with atomics: ulp-level run-to-run noise that flips a
    # discrete decision (NMS / top-k) about once in 15 runs and with it the whole RoI sample.  Deterministic
    # solvers make the step reproducible (tools/probe/nondet_probe.py: 40/40 identical).
    monkeypatch.setattr(torch.backends.cudnn, 'deterministic', True)
    det = build_and_load(dev)
    data = make_data(g, dev)
    torch.manual_seed(int(g['seed']))
    np.random.seed(int(g['seed']))
    out = det.train_step(data, None)
    out['loss'].backward()
    torch.cuda.synchronize()
    print({k: (v, float(g['lv_' + k])) for k, v in out['log_vars'].items()})
    for k, v in out['log_vars'].items():
        ref = float(g['lv_' + k])
        tol = 1e-3 if k != 'acc' else 1e-2
        assert abs(v - ref) <= tol * abs(ref), (k, v, ref)
    same = (det.roi_head.bbox_targets[0].cpu().numpy() == g['roi_labels']).mean()
    assert same >= 0.999, same
    gdev = {k: abs(v - float(g['gn_' + k])) / float(g['gn_' + k]) for k, v in grad_groups(det).items()}
    print('grad-norm dev', {k: f'{v:.2e}' for k, v in gdev.items()})
    for k, v in gdev.items():
        assert v <= 1e-3, (k, v)


@pytest.mark.gpu
def test_r101_dc5_oadg_config_trains_one_step(dev):
    """BASELINE configs[3]: R101-DC5 (dilated C5, no FPN, 2048-channel RoI path, allowed_border=0, OA-Mix
    'augmix.all') builds from configs/oadg/faster_rcnn_r101_dc5_1x_dwd_oadg.py and runs a full bf16 step."""
    from oadg_amd import Config, build_detector, hip_conv
    from oadg_amd.apis import TrainEngine, build_optimizer, set_random_seed
    from oadg_amd.pipelines import DevicePipeline, SyntheticCityscapes
    cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'oadg', 'faster_rcnn_r101_dc5_1x_dwd_oadg.py'))
    set_random_seed(0)
    det = build_detector(cfg.model)
    det.init_weights(allow_missing_pretrained=True)
    det = det.to(dev).to(memory_format=torch.channels_last).train()
    assert sum(p.numel() for p in det.parameters()) == 184580783
    hip_conv.enable()
    try:
        eng = TrainEngine(det, build_optimizer(det, cfg.optimizer), amp_dtype=torch.bfloat16)
        # BASELINE configs[3] at its full per-GPU size: bs = 2 at the DWD image size 736 x 1280 (the reference's
        # configs/OA-DG/dwd/*.py resize to (1280, 720); 736 = padded to a multiple of 32).  Size-independent properties:
        # finite positive losses over consecutive steps, 2000 proposals per image through nms_pre = 12000 (55,200
        # anchors), a contrastive batch of 2 * 512 * 2 sampled RoIs + the random proposals, all parameters updated.
        import time
        ds = SyntheticCityscapes(img_shape=(736, 1280), num_boxes=12, num_classes=7, box_size=(24, 300), device=dev)
        pipe = DevicePipeline(cfg.data.train.pipeline, dtype=torch.bfloat16)
        w0 = det.backbone.layer4[2].conv2.weight.detach().clone()
        times = []
        for it in range(4):
            imgs, boxes, labels = ds.batch([2 * it, 2 * it + 1])
            data = pipe(imgs, boxes, labels)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = eng.step(data)
            loss = float(out['loss'])
            times.append(time.perf_counter() - t0)
            assert np.isfinite(loss) and loss > 0
        print(f'R101-DC5 OA-DG bs 2 at 736x1280: {min(times[1:]) * 1e3:.1f} ms per step (model only, bf16)')
        assert det.roi_head.bbox_targets[0].shape[0] == 2 * 512 * 2
        assert not torch.equal(w0, det.backbone.layer4[2].conv2.weight.detach())
    finally:
        hip_conv.enable(False)
    assert {'loss_rpn_cls', 'loss_rpn_bbox', 'loss_cls', 'acc', 'loss_bbox', 'loss_cont', 'loss'} <= set(out['log_vars'])


def _gpu_step_vs_fixture(dev, golden_dir, fixture, cfg, bf16, tol_loss, tol_acc, tol_gn, min_same, monkeypatch):
    """one train step of the PRODUCT path on the device against a fixture written by the reference's own python.
    bf16=True is the benchmarked configuration: channels_last model, autocast bf16, every convolution on the csrc MFMA
    kernels (hip_conv.enable()), BN folded, HIP losses / RoIAlign / NMS / assigner."""
    from oadg_amd import hip_conv
    from oadg_amd.detectors import integrate_data
    g = np.load(os.path.join(golden_dir, fixture))
    torch.backends.cudnn.allow_tf32 = False
    monkeypatch.setattr(torch.backends.cudnn, 'deterministic', True)
    det = build_and_load(dev, cfg)
    data = make_data(g, dev)
    torch.manual_seed(int(g['seed']))
    np.random.seed(int(g['seed']))
    if bf16:
        det = det.to(memory_format=torch.channels_last)
        hip_conv.enable()
        try:
            with torch.autocast('cuda', dtype=torch.bfloat16):
                losses = det(**integrate_data(data, det.train_cfg))
            loss, log_vars = det._parse_losses(losses)
            loss.backward()
        finally:
            hip_conv.enable(False)
    else:
        out = det.train_step(data, None)
        out['loss'].backward()
        log_vars = out['log_vars']
    torch.cuda.synchronize()
    dev_l = {k: abs(v - float(g['lv_' + k])) / abs(float(g['lv_' + k])) for k, v in log_vars.items()}
    same = (det.roi_head.bbox_targets[0].cpu().numpy() == g['roi_labels']).mean()
    dev_g = {k: abs(v - float(g['gn_' + k])) / float(g['gn_' + k]) for k, v in grad_groups(det).items()}
    print(fixture, 'bf16' if bf16 else 'fp32', 'loss dev', {k: f'{v:.2e}' for k, v in dev_l.items()},
          'labels same', same, 'grad-norm dev', {k: f'{v:.2e}' for k, v in dev_g.items()})
    for k, v in dev_l.items():
        assert v <= (tol_acc if k == 'acc' else tol_loss), (k, v)
    assert same >= min_same, same
    assert set(dev_g) == {k[3:] for k in g.files if k.startswith('gn_')}
    for k, v in dev_g.items():
        assert v <= tol_gn, (k, v)


@pytest.mark.gpu
def test_bf16_mfma_step_matches_reference_on_gpu(dev, golden_dir, monkeypatch):
    """THE BENCHMARKED PATH (bench.py: bf16 autocast + csrc MFMA convolutions + folded BN) against the reference's
    fp32 step (base.py:413-455 via tests/golden/make_golden_model.py).  bf16 has 8 mantissa bits: activations carry
    ~4e-3 relative rounding per layer, which also flips a few discrete decisions (top-k / NMS / IoU thresholds, ~1 %
    of the sampled RoIs).  Measured on MI355X (round 2): loss terms 1e-4..3e-3 (3e-2 for loss_cls / loss_bbox
    of the DC5 fixture, where 2 of 2048 sampled RoIs flip), acc <= 3.6e-2, per-module gradient norms 1e-5..1.1e-2,
    sampled labels >= 99.9 % identical.  Round 3 (measured again, RoIAlign backward by tiles): loss terms <= 5.3e-3,
    acc 3.2e-2, gradient norms <= 8.9e-3, labels identical.  Asserted at ~3x that: 1.5e-2 / 1e-1 (acc) / 2.5e-2 /
    99.9 %.  north_star's 1e-4 bar is held by the per-kernel tests on identical operands and by the fp32 product
    step (1e-7..3.3e-4 against the same fixtures)."""
    # round 6 (profiles/r06_parity.txt): loss terms <= 4.3e-3, acc 5.7e-2, gradient norms <= 1.23e-2, labels identical: the loss
    # gate now sits at 2x the measured deviation (was 3.5x)
    _gpu_step_vs_fixture(dev, golden_dir, 'model_step_256x512.npz', CFG, True, 9e-3, 1e-1, 2.5e-2, 0.999, monkeypatch)


@pytest.mark.gpu
@pytest.mark.parametrize('bf16', [False, True])
def test_full_size_config1_step_matches_reference_on_gpu(dev, golden_dir, bf16, monkeypatch):
    """SURVEY 8c G7 at BASELINE config 1's real shape: N=2, 1024x2048 (the reference needs 72 s and 13.5 GB for this
    step on the build container's 8 cores)."""
    # bf16 `acc`: an argmax over 9 near-tied random-init logits per RoI (reference 35.7 % of 2048 RoIs).  The losses agree
    # to 2e-3 and the sampled labels are identical, yet 25-55 argmaxes flip with the bf16 rounding pattern of the
    # convolutions (3.6e-2 relative with tap-major K order, 7.5e-2 with the chunk-major order of round 2): 1.5e-1 asserted.
    # Measured (round 3): fp32 loss terms <= 2.5e-4, acc 4.1e-3, gradient norms <= 2e-4, labels identical; bf16 loss
    # terms <= 2e-3, gradient norms <= 7.1e-3, labels identical.  Asserted at ~3x the measured deviations.
    # Round 6 (profiles/r06_parity.txt, the fp32 path on the csrc fp32-MFMA convolutions): fp32 loss terms <= 9.6e-5 - north_star's
    # 1e-4 holds END TO END on the full-size step -, acc identical, gradient norms <= 2.8e-4, labels identical; bf16 loss terms
    # <= 2.04e-3, acc 4.8e-2, gradient norms <= 5.7e-3, labels identical.  Gates at ~2x measured (a 10x regression
    # passed the old 1e-3 / 1.5e-2 / 1e-3).  fp32 `acc`: identical today; 3e-3 admits two flipped argmaxes of 2048.
    tol = (4.1e-3, 1e-1, 1.2e-2, 0.999) if bf16 else (2e-4, 3e-3, 6e-4, 0.999)
    _gpu_step_vs_fixture(dev, golden_dir, 'model_step_1024x2048.npz', CFG, bf16, *tol, monkeypatch)


def test_r101_dc5_host_logic_reproduces_reference_step(golden_dir):
    """BASELINE configs[3] on CPU with oracle ops: single-level RoI path (single_level_roi_extractor.py:107-110),
    2048-channel RoIAlign, 15 anchors per location, allowed_border=0, nms_pre=12000 >= split_thr=10000 (the per-level
    loop of batched_nms), max_per_img=2000, num_classes=7."""
    from oracle.backend import oracle_ops
    from oadg_amd import layers
    g = np.load(os.path.join(golden_dir, 'model_step_dc5_384x768.npz'))
    det = build_and_load(cfg=DC5_CFG)
    assert sum(p.numel() for p in det.parameters()) == int(g['n_params']) == 184580783
    assert sum(p.numel() for p in det.parameters() if p.requires_grad) == int(g['n_trainable'])
    data = make_data(g)
    torch.manual_seed(int(g['seed']))
    np.random.seed(int(g['seed']))
    prev, layers.FOLD_EVAL_BN = layers.FOLD_EVAL_BN, False
    try:
        with oracle_ops():
            out = det.train_step(data, None)
            out['loss'].backward()
    finally:
        layers.FOLD_EVAL_BN = prev
    for k, v in out['log_vars'].items():
        ref = float(g['lv_' + k])
        assert abs(v - ref) <= 1e-6 * abs(ref), (k, v, ref)
    assert np.array_equal(det.roi_head.bbox_targets[0].numpy(), g['roi_labels'])
    for k, v in grad_groups(det).items():
        ref = float(g['gn_' + k])
        assert abs(v - ref) <= 1e-5 * ref, (k, v, ref)
    params = dict(det.named_parameters())
    for k in g.files:
        if k.startswith('g_'):
            mine = params[k[2:]].grad.flatten()[:4096].numpy()
            assert np.abs(mine - g[k]).max() <= 1e-5 * np.abs(g[k]).max() + 1e-9, k


@pytest.mark.gpu
@pytest.mark.parametrize('bf16', [False, True])
def test_r101_dc5_step_matches_reference_on_gpu(dev, golden_dir, bf16, monkeypatch):
    """R101-DC5 OA-DG product path on the device against the reference-generated fixture: fp32 (library convolutions +
    HIP RoIAlign / NMS / losses) and the bf16 MFMA path incl. the dilated 3x3 and the 2048->2048 RPN convolution.
    Measured (round 3): fp32 loss terms <= 1.1e-6, gradient norms <= 5.8e-5, labels identical; bf16 loss terms <= 1.6e-2
    (loss_cls: 2 of 2048 sampled RoIs flip), acc 5.7e-4, gradient norms <= 2.1e-2, labels 99.9 %."""
    # (round 4: with the frozen stage-1 blocks fused into one launch the bf16 rounding pattern of their outputs changed and
    #  `acc` - an argmax over near-tied random-init logits, see the R50 test above - moved from 5.7e-4 to 1.0e-2)
    # Round 6 (profiles/r06_parity.txt): fp32 loss terms <= 1.3e-6, acc identical, gradient norms <= 7.9e-5; bf16 loss terms
    # <= 3.6e-2 (loss_cls: two sampled RoIs flip), acc 1.02e-2, gradient norms <= 1.12e-2, labels 99.8 %.  Gates at <= 2x measured.
    tol = (5e-2, 2e-2, 2.3e-2, 0.998) if bf16 else (1e-5, 3e-3, 2e-4, 0.999)
    _gpu_step_vs_fixture(dev, golden_dir, 'model_step_dc5_384x768.npz', DC5_CFG, bf16, *tol, monkeypatch)


@pytest.mark.gpu
def test_bf16_training_overfits_a_fixed_batch(dev):
    """System-level check of the benchmarked configuration's gradients: 60 SGD steps (bf16 autocast, csrc MFMA
    convolutions, folded BN, HIP losses) on ONE fixed pair of images - OA-Mix still draws a new second view every step -
    must drive the total loss down substantially; wrong-signed or mis-scaled gradients anywhere in the hand-written
    backward kernels would not."""
    from oadg_amd import Config, build_detector, hip_conv
    from oadg_amd.apis import TrainEngine, build_optimizer, set_random_seed
    from oadg_amd.pipelines import DevicePipeline, SyntheticCityscapes
    cfg = Config.fromfile(CFG)
    set_random_seed(0)
    det = build_detector(cfg.model)
    det.init_weights(allow_missing_pretrained=True)
    det = det.to(dev).to(memory_format=torch.channels_last).train()
    opt = build_optimizer(det, dict(type='SGD', lr=0.01, momentum=0.9, weight_decay=0.0001))
    try:
        eng = TrainEngine(det, opt, amp_dtype=torch.bfloat16)
        ds = SyntheticCityscapes(img_shape=(256, 512), num_boxes=8, box_size=(24, 160), device=dev)
        pipe = DevicePipeline(cfg.data.train.pipeline, dtype=torch.bfloat16)
        imgs, boxes, labels = ds.batch([0, 1])
        hist = []
        for it in range(60):
            out = eng.step(pipe(imgs, boxes, labels))
            hist.append({k: float(v) for k, v in out['log_vars'].items()})
        first = np.mean([h['loss'] for h in hist[:5]])
        last = np.mean([h['loss'] for h in hist[-5:]])
        print('loss', round(first, 4), '->', round(last, 4), {k: round(hist[-1][k], 4) for k in hist[-1]})
        assert all(np.isfinite(h['loss']) for h in hist)
        assert last < 0.6 * first, (first, last)
        assert np.mean([h['loss_rpn_cls'] for h in hist[-5:]]) < 0.5 * np.mean([h['loss_rpn_cls'] for h in hist[:5]])
    finally:
        hip_conv.enable(False)


@pytest.mark.gpu
def test_config2_whole_step_bs4_1024x2048_bf16(dev):
    """BASELINE configs[1] - the benchmarked workload - as ONE whole step through the product path (bench.py runs the same
    step but only asserts a finite loss): bs 4 x 2 views at 1024 x 2048, 20 boxes per image, bf16 autocast, every
    convolution on the csrc MFMA kernels, OA-Mix on the device, HIP losses / RoIAlign / NMS / assigner
    (mmdet/models/detectors/base.py:413-455 is the reference step).  Size-independent properties: the 7 log vars are
    finite, 8 x 512 = 4096 sampled RoIs, the contrastive batch is 4096 + the random proposals, EVERY trainable parameter
    receives a finite, non-zero gradient, and two runs from the same seeds are bit-identical (deterministic kernels:
    fixed-order weight-gradient / column-sum reductions, RoIAlign backward by output tiles)."""
    from oadg_amd import Config, build_detector, hip_conv
    from oadg_amd.apis import TrainEngine, build_optimizer, set_random_seed
    from oadg_amd.pipelines import DevicePipeline, SyntheticCityscapes
    cfg = Config.fromfile(CFG)
    set_random_seed(0)
    # name-seeded weights as in the fixtures (inputs.named_weights): mmdet's random initialisation zeroes the last BatchNorm
    # scale of every bottleneck (zero_init_residual), which makes 91 backbone gradients EXACTLY zero by construction
    det = build_and_load(dev).to(memory_format=torch.channels_last).train()
    try:
        eng = TrainEngine(det, build_optimizer(det, cfg.optimizer), amp_dtype=torch.bfloat16)
        ds = SyntheticCityscapes(img_shape=(1024, 2048), num_boxes=20, num_classes=8, seed=0, device=dev)
        pipe = DevicePipeline(cfg.data.train.pipeline, dtype=torch.bfloat16)
        set_random_seed(5)
        data = pipe(*ds.batch(range(4)))
        assert tuple(data['img'].shape) == (4, 3, 1024, 2048) and tuple(data['img2'].shape) == (4, 3, 1024, 2048)
        runs = []
        for rep in range(2):
            set_random_seed(11)                      # RandomSampler (torch CPU generator) + random proposals (numpy)
            det.zero_grad(set_to_none=True)
            # (integrate_data merges the views INTO the dict it is given, like base.py:22-48: every run gets its own copy)
            fresh = {k: (list(v) if isinstance(v, list) else v) for k, v in data.items()}
            (loss, log_vars), n = eng.forward_losses(fresh)
            loss.backward()
            torch.cuda.synchronize()
            lv = {k: float(v) for k, v in log_vars.items()}
            assert n == 8                     # num_samples = len(img_metas) after integrate_data: 4 images x 2 views
            assert set(lv) == {'loss_rpn_cls', 'loss_rpn_bbox', 'loss_cls', 'acc', 'loss_bbox', 'loss_cont', 'loss'}
            assert all(np.isfinite(v) for v in lv.values()), lv
            assert lv['loss_cont'] > 0 and lv['loss_cls'] > 0 and lv['loss_rpn_cls'] > 0
            labels = det.roi_head.bbox_targets[0]
            assert labels.shape[0] == 8 * 512
            rois = det.roi_head._last_rois
            assert rois[0].shape[0] == 4096 and len(rois) == 2 and 8 * 11 <= rois[1].shape[0] <= 8 * 18
            grads = {}
            for name, p in det.named_parameters():
                if p.requires_grad:
                    assert p.grad is not None, name
                    g = p.grad.float()
                    assert torch.isfinite(g).all(), name
                    assert float(g.abs().sum()) > 0, name
                    grads[name] = p.grad.detach().clone()
            runs.append((lv, grads))
        assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
        bad = [k for k in runs[0][1] if not torch.equal(runs[0][1][k], runs[1][1][k])]
        assert not bad, bad[:5]
        print('config 2 whole step:', {k: round(v, 4) for k, v in runs[0][0].items()}, 'trainable tensors', len(runs[0][1]))
        # the same step the way TrainEngine.step runs it (round 4): column sums AND the small maps' weight gradients
        # deferred and launched in groups (hip_conv.DEFER_WGRAD).  Run to run bit-identical; against the immediate form
        # equal to fp32 rounding (other split counts) - layer4's 512 x 3 x 3 filters to bf16 rounding (the immediate
        # form hands them over in bf16, the grouped form keeps fp32 partials).
        deferred = []
        for rep in range(2):
            set_random_seed(11)
            det.zero_grad(set_to_none=True)
            fresh = {k: (list(v) if isinstance(v, list) else v) for k, v in data.items()}
            hip_conv.begin_step(defer=True)
            try:
                (loss, log_vars), n = eng.forward_losses(fresh)
                loss.backward()
                pending = len(hip_conv._WQ)
            finally:
                hip_conv.end_backward()
            torch.cuda.synchronize()
            assert not hip_conv._WQ and not hip_conv._PENDING
            deferred.append(({k: float(v) for k, v in log_vars.items()},
                             {name: p.grad.detach().clone() for name, p in det.named_parameters() if p.requires_grad}))
        assert deferred[0][0] == deferred[1][0] == runs[0][0]
        bad = [k for k in deferred[0][1] if not torch.equal(deferred[0][1][k], deferred[1][1][k])]
        assert not bad, bad[:5]
        for k, a in runs[0][1].items():
            b = deferred[0][1][k]
            err = (a.float() - b.float()).abs().max().item() / (a.float().abs().max().item() + 1e-20)
            wide = 'layer4' in k and ('conv2' in k or 'bn2' in k)
            assert err <= (5e-3 if wide else 1e-4), (k, err)
    finally:
        hip_conv.enable(False)


@pytest.mark.gpu
def test_flat_reducer_over_rccl_world1_with_deferred_launches(dev):
    """The data-parallel step on the device with world size 1 (an RCCL process group of one rank - there is no multi-GPU
    box): FlatGradReducer's hooks complete buckets while weight gradients / column sums of the same bucket are still
    DEFERRED (hip_conv.flush_deferred runs before a bucket is packed).  One TrainEngine.step with the reducer leaves the
    same parameters as one step without it (averaging over one rank is the identity; the groups are cut at other
    points, so fp32 rounding differs): relative parameter-update error <= 1e-4, everything finite, and the reducer really
    re-pointed every .grad into its flat buffer."""
    import torch.distributed as dist
    from oadg_amd import Config, hip_conv
    from oadg_amd.apis import TrainEngine, build_optimizer, set_random_seed
    from oadg_amd.pipelines import DevicePipeline, SyntheticCityscapes
    cfg = Config.fromfile(CFG)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(23450 + os.getpid() % 2000), RANK='0', WORLD_SIZE='1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        ds = SyntheticCityscapes(img_shape=(512, 1024), num_boxes=12, num_classes=8, seed=0, device=dev)
        pipe = DevicePipeline(cfg.data.train.pipeline, dtype=torch.bfloat16)
        set_random_seed(5)
        data = pipe(*ds.batch(range(2)))
        after = {}
        for mode in ('plain', 'reducer'):
            set_random_seed(0)
            det = build_and_load(dev).to(memory_format=torch.channels_last).train()
            det.log_vars_on_host = False
            before = {n: p.detach().clone() for n, p in det.named_parameters() if p.requires_grad}
            eng = TrainEngine(det, build_optimizer(det, cfg.optimizer), distributed=(mode == 'reducer'),
                              amp_dtype=torch.bfloat16)
            assert (eng.reducer is not None) == (mode == 'reducer')
            set_random_seed(11)
            fresh = {k: (list(v) if isinstance(v, list) else v) for k, v in data.items()}
            out = eng.step(fresh)
            torch.cuda.synchronize()
            assert np.isfinite(float(out['loss']))
            if eng.reducer is not None:
                flat = eng.reducer.flat
                lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * 4
                assert all(lo <= p.grad.data_ptr() < hi for p in eng.reducer.params)
                assert len(eng.reducer.buckets) == 4
            after[mode] = {n: (p.detach() - before[n]) for n, p in det.named_parameters() if p.requires_grad}
            del eng, det
        worst = 0.0
        for n, a in after['plain'].items():
            b = after['reducer'][n]
            assert torch.isfinite(b).all(), n
            scale = a.abs().max().item()
            assert scale > 0, n
            err = (a - b).abs().max().item() / scale
            # (BatchNorm scale gradients are (dot product - mean x bias gradient) / sigma: a difference that amplifies the
            #  fp32 rounding of the regrouped sums; layer4's 3x3 filters: bf16 hand-over in the immediate form)
            wide = 'layer4' in n and ('conv2' in n or 'bn2' in n)
            bn_scale = '.bn' in n and n.endswith('weight')
            assert err <= (5e-3 if wide else 2e-3 if bn_scale else 1e-4), (n, err)
            worst = max(worst, err)
        print('reducer vs plain: worst relative update error', worst)
    finally:
        hip_conv.enable(False)
        dist.destroy_process_group()


@pytest.mark.gpu
def test_r101_dc5_step_through_the_rccl_reducer_with_128_mb_buckets(dev, monkeypatch):
    """BASELINE configs[3]'s 737 MB of gradients through FlatGradReducer over RCCL (world size 1 - no
    multi-GPU box) with OADG_BUCKET_MB=128: the five buckets of DESIGN section 7 (the 411 MB FC weight alone makes the first
    one 416.7 MB), every collective issued in bucket order, finite losses over two steps at bs 2 / 736 x 1280, all
    parameters updated, .grad of every parameter a slice of the flat buffer."""
    import torch.distributed as dist
    from oadg_amd import Config, build_detector, hip_conv
    from oadg_amd.apis import TrainEngine, build_optimizer, set_random_seed
    from oadg_amd.pipelines import DevicePipeline, SyntheticCityscapes
    monkeypatch.setenv('OADG_BUCKET_MB', '128')
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(25450 + os.getpid() % 2000), RANK='0', WORLD_SIZE='1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'oadg', 'faster_rcnn_r101_dc5_1x_dwd_oadg.py'))
        set_random_seed(0)
        det = build_detector(cfg.model)
        det.init_weights(allow_missing_pretrained=True)
        det = det.to(dev).to(memory_format=torch.channels_last).train()
        det.log_vars_on_host = False
        eng = TrainEngine(det, build_optimizer(det, cfg.optimizer), distributed=True, amp_dtype=torch.bfloat16)
        red = eng.reducer
        assert red is not None
        sizes = [round((b['end'] - b['start']) * 4 / 1e6, 1) for b in red.buckets]
        assert sizes == [416.7, 151.6, 134.5, 28.8, 5.4], sizes
        order = []
        orig = red._launch
        red._launch = lambda b: (order.append([x is b for x in red.buckets].index(True)), orig(b))[1]
        ds = SyntheticCityscapes(img_shape=(736, 1280), num_boxes=12, num_classes=7, box_size=(24, 300), device=dev)
        pipe = DevicePipeline(cfg.data.train.pipeline, dtype=torch.bfloat16)
        w0 = {n: p.detach().clone() for n, p in det.named_parameters() if p.requires_grad}
        for it in range(2):
            out = eng.step(pipe(*ds.batch([2 * it, 2 * it + 1])))
            torch.cuda.synchronize()
            assert np.isfinite(float(out['loss'])) and float(out['loss']) > 0
        assert order == [0, 1, 2, 3, 4] * 2, order
        lo, hi = red.flat.data_ptr(), red.flat.data_ptr() + red.flat.numel() * 4
        assert red.flat.numel() * 4 > 730e6
        assert all(lo <= p.grad.data_ptr() < hi for p in red.params)
        same = [n for n, p in det.named_parameters() if p.requires_grad and torch.equal(p.detach(), w0[n])]
        assert not same, same[:5]
    finally:
        hip_conv.enable(False)
        dist.destroy_process_group()
