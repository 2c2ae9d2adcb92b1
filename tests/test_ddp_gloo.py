"""CPU, world_size 2 over gloo: the data-parallel path (one process per device, gradient all-reduce by DDP,
packed log-var all-reduce) - SURVEY.md 8e.  The HIP entry points are swapped for the oracle (no GPU here)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q, torch_ddp=False, sparse_rank1=False):
    os.environ['OADG_USE_TORCH_DDP'] = '1' if torch_ddp else '0'
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(3)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import oadg_amd  # noqa: F401
    from oadg_amd import Config, build_detector
    from oadg_amd.apis import TrainEngine, build_optimizer, set_random_seed
    from oracle.backend import oracle_ops
    from inputs import model_batch
    cfg = Config.fromfile(os.path.join(ROOT, 'configs/oadg/faster_rcnn_r50_fpn_1x_cityscapes_oadg.py'))
    set_random_seed(0)
    det = build_detector(cfg.model)
    det.init_weights(allow_missing_pretrained=True)
    det.train()
    eng = TrainEngine(det, build_optimizer(det, cfg.optimizer), distributed=True)
    b = model_batch(10 + rank, 1, 192, 320, n_gt=6)          # different data on each rank
    if sparse_rank1 and rank == 1:
        # ONE small object: a handful of foreground RoIs (the gt itself per view) <= ContrastiveLossPlus.min_samples = 10,
        # the state in which the reference skips the contrastive branch (contrastive_head.py:123-129)
        b['gt_bboxes'] = [np.array([[40., 30., 52., 44.]], np.float32)]
        b['gt_labels'] = [np.array([2], np.int64)]
    order = []
    if eng.reducer is not None:
        orig = eng.reducer._launch
        eng.reducer._launch = lambda bk: (order.append([x is bk for x in eng.reducer.buckets].index(True)), orig(bk))[1]
    shape = b['img'].shape[2:] + (3,)
    t = torch.tensor
    data = dict(img=t(b['img']), img2=t(b['img2']), gt_bboxes=[t(x) for x in b['gt_bboxes']],
                gt_bboxes2=[t(x) for x in b['gt_bboxes']], gt_labels=[t(x) for x in b['gt_labels']],
                multilevel_boxes=[t(x) for x in b['multilevel_boxes']], oamix_boxes=[t(x) for x in b['oamix_boxes']],
                img_metas=[dict(img_shape=shape, pad_shape=shape, ori_shape=shape, scale_factor=1.0, flip=False)])
    set_random_seed(1 + rank)
    w0 = det.roi_head.bbox_head.fc_cls.weight.detach().clone()
    with oracle_ops():
        out = eng.step(data)
    p = torch.cat([q_.detach().flatten()[:64] for q_ in det.parameters() if q_.requires_grad])
    g = det.roi_head.bbox_head.fc_cls.weight.grad.detach().flatten()[:256].clone()
    lv = out['log_vars']
    extra = None
    if sparse_rank1:
        fc = det.roi_head.bbox_head.fc_cont[0]
        lab = det.roi_head.bbox_head.roi_targets[0]
        extra = (order, int((lab != lab.max()).sum()),
                 fc.weight.grad.detach().flatten()[:512].clone().numpy(), len(eng.reducer.buckets))
    q.put((rank, p.numpy(), g.numpy(), dict(lv), float(out['loss']),
           bool((det.roi_head.bbox_head.fc_cls.weight.detach() != w0).any()), extra))
    dist.barrier()
    dist.destroy_process_group()


def _collect(procs, q, timeout):
    """one result per process; fails fast when a worker died instead of waiting out the queue timeout"""
    import queue
    import time
    out, t0 = [], time.time()
    while len(out) < len(procs):
        try:
            out.append(q.get(timeout=2))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            assert not dead, f'worker exited with {dead}'
            assert time.time() - t0 < timeout, 'timed out waiting for the workers'
    return sorted(out, key=lambda r: r[0])


def _worker_unused(rank, world, port, q):
    """toy net with one bucket per layer; rank 1 skips the middle branch, so that parameter has NO gradient there"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import torch.nn as nn
    import oadg_amd  # noqa: F401
    from oadg_amd.apis import FlatGradReducer
    torch.manual_seed(0)
    net = nn.ModuleList([nn.Linear(32, 32, bias=False), nn.Linear(32, 48, bias=False), nn.Linear(32, 32, bias=False),
                         nn.Linear(32, 8, bias=False)])
    red = FlatGradReducer(net, bucket_mb=1e-4, tail_mb=0)      # ~26 floats per bucket: every layer its own bucket
    assert len(red.buckets) == 4
    order = []
    orig = red._launch
    red._launch = lambda b: (order.append([x is b for x in red.buckets].index(True)), orig(b))[1]
    torch.manual_seed(100 + rank)
    out = []
    for step in range(2):
        for p_ in net.parameters():
            p_.grad = None
        x = torch.randn(4, 32)
        h = net[0](x)
        y = net[3](net[2](h)).sum()
        if rank == 0 or step == 1:          # rank 1, step 0: net[1] takes no part in the graph
            y = y + net[1](h).sum()
        y.backward()
        local = [None if p_.grad is None else p_.grad.clone() for p_ in net.parameters()]
        red.finish()
        out.append(([None if g is None else g.numpy() for g in local],
                    [p_.grad.clone().numpy() for p_ in net.parameters()]))
    q.put((rank, out, order))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_flat_reducer_parameter_without_gradient_on_one_rank():
    """a parameter that received no gradient on ONE rank: collectives are still issued in bucket order on both ranks
    (no hang, no mismatched buffers), the missing gradient counts as zeros, and nothing crashes on the aliasing
    ``cat(out=)`` path."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_unused, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = _collect(procs, q, 240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, out0, order0), (_, out1, order1) = res
    assert order0 == order1 == [0, 1, 2, 3] * 2
    assert out1[0][0][1] is None and out0[0][0][1] is not None
    for step in range(2):
        (loc0, red0), (loc1, red1) = out0[step], out1[step]
        for i in range(4):
            a = loc0[i] if loc0[i] is not None else np.zeros_like(red0[i])
            b = loc1[i] if loc1[i] is not None else np.zeros_like(red0[i])
            assert np.allclose(red0[i], 0.5 * (a + b), rtol=1e-6, atol=1e-7), (step, i)
            assert np.array_equal(red0[i], red1[i])


def _worker_sink(rank, world, port, q, overlap=True):
    """gradients produced by a function with a kernel of its own (here hip_ops.cast_all_bf16, which runs on CPU tensors
    too) are written into the reducer's bucket slices: ``param.grad`` lies inside the flat buffer BEFORE the collective"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import torch.nn as nn
    import torch.nn.functional as F
    import oadg_amd  # noqa: F401
    from oadg_amd import hip_ops
    from oadg_amd.apis import FlatGradReducer
    torch.manual_seed(0)
    net = nn.ModuleList([nn.Linear(32, 64), nn.Linear(64, 16), nn.Linear(16, 4)])
    red = FlatGradReducer(net, bucket_mb=1e-3, tail_mb=0, overlap=overlap)          # 262 floats per bucket: three buckets
    phase, phases = ['-'], []
    assert hip_ops.GRAD_SINK is red.views
    lo, hi = red.flat.data_ptr(), red.flat.data_ptr() + red.flat.numel() * 4
    seen = []

    def pre(r, b):       # right before the bucket's all-reduce is issued
        phases.append(phase[0])
        for p_ in b['params']:
            seen.append((p_.grad is not None and lo <= p_.grad.data_ptr() < hi and
                         p_.grad.data_ptr() == r.views[p_].data_ptr(), p_.grad.clone()))
    red.pre_collective = pre
    torch.manual_seed(100 + rank)
    out = []
    for step in range(2):
        for p_ in net.parameters():
            p_.grad = None
        seen.clear()
        x = torch.randn(8, 32)
        ps = [t for m in net[:2] for t in (m.weight, m.bias)]          # layers 0 and 1 through the one-pass cast
        c = dict(zip(ps, hip_ops.cast_all_bf16(ps)))
        h = F.linear(x.bfloat16(), c[net[0].weight], c[net[0].bias]).relu()
        h = F.linear(h, c[net[1].weight], c[net[1].bias]).relu()
        y = net[2](h.float()).sum()                                      # layer 2: a torch operator produces its gradients
        phases.clear()
        phase[0] = 'backward'
        y.backward()
        phase[0] = 'finish'
        red.finish()
        out.append(([f_ for f_, _ in seen], [g_.numpy() for _, g_ in seen],
                    [p_.grad.clone().numpy() for b in red.buckets for p_ in b['params']],
                    red.in_place_bytes, red.packed_bytes, list(phases)))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize('overlap', [True, False])
def test_flat_reducer_gradients_are_written_into_the_buckets(overlap):
    """no packing copy - the producers' gradients ARE bucket slices before the collective (4 of the 6
    parameters here; the last layer's come from torch operators and are packed by the multi-tensor copy), and the result
    is the mean of the two ranks' local gradients either way.  ``overlap=False`` (round 6, the safety valve of
    FlatGradReducer): the same buckets, the same result, every collective issued by finish() instead of from the hooks."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 25500 + (os.getpid() % 2000) + (0 if overlap else 3)
    procs = [ctx.Process(target=_worker_sink, args=(r, 2, port, q, overlap)) for r in range(2)]
    for p in procs:
        p.start()
    res = _collect(procs, q, 240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, out0), (_, out1) = res
    for step in range(2):
        in0, loc0, red0, ipb0, pkb0, ph0 = out0[step]
        in1, loc1, red1, _, _, ph1 = out1[step]
        # with overlap every bucket is reduced from the hooks while the backward pass runs, without it by finish()
        assert len(ph0) >= 2 and ph0 == ph1 == ['backward' if overlap else 'finish'] * len(ph0), (ph0, ph1)
        # bucket order = reverse registration: layer 2 (bias, weight: packed), then layers 1 and 0 (in place)
        assert in0 == in1 == [False, False, True, True, True, True], in0
        for a, b, r0, r1 in zip(loc0, loc1, red0, red1):
            assert np.allclose(r0, 0.5 * (a + b), rtol=1e-6, atol=1e-7)
            assert np.array_equal(r0, r1)
        assert ipb0 == (step + 1) * 4 * (32 * 64 + 64 + 64 * 16 + 16) and pkb0 == (step + 1) * 4 * (16 * 4 + 4)


_RESULTS = {}


@pytest.mark.timeout(600)
@pytest.mark.parametrize('torch_ddp', [False, True])
def test_two_rank_data_parallel_step(torch_ddp):
    """torch_ddp=False: apis.FlatGradReducer (the default); True: torch DDP.  Both must agree with each other."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + (7 if torch_ddp else 0)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, torch_ddp)) for r in range(2)]
    for p in procs:
        p.start()
    res = _collect(procs, q, 540)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, p0, g0, lv0, l0, ch0, _x0), (_, p1, g1, lv1, l1, ch1, _x1) = res
    assert ch0 and ch1, 'the optimizer step changed nothing'
    assert np.array_equal(g0, g1), 'gradients were not all-reduced to the same mean'
    assert np.array_equal(p0, p1), 'parameters diverged across ranks'
    assert l0 != l1, 'ranks saw different data, their local losses must differ'
    assert lv0.keys() == lv1.keys()
    for k in lv0:                      # log vars are the cross-rank means (base.py:270-275)
        assert abs(lv0[k] - lv1[k]) <= 1e-6 * max(1.0, abs(lv0[k])), k
    assert abs(lv0['loss'] - 0.5 * (l0 + l1)) <= 1e-5 * abs(lv0['loss'])
    _RESULTS[torch_ddp] = (g0, p0)
    if len(_RESULTS) == 2:             # the lean reducer and torch DDP average to the same gradients / parameters
        assert np.allclose(_RESULTS[False][0], _RESULTS[True][0], rtol=1e-6, atol=1e-9)
        assert np.allclose(_RESULTS[False][1], _RESULTS[True][1], rtol=1e-6, atol=1e-9)


def test_flat_reducer_layout_helpers_and_tail_bucket(monkeypatch):
    """FlatGradReducer without a process group: gradient slices carry the parameter's memory layout (channels_last conv
    weights), gradients land in parameter memory order whatever their own strides, and the last-ready parameters get a
    small bucket of their own."""
    import torch.nn as nn
    from oadg_amd import apis
    R = apis.FlatGradReducer
    p = torch.randn(8, 4, 3, 3).contiguous(memory_format=torch.channels_last)
    flat = torch.zeros(p.numel())
    v = R._like_param(flat, p)
    assert v.stride() == p.stride() and v.shape == p.shape
    for g in (torch.randn(8, 4, 3, 3).contiguous(memory_format=torch.channels_last), torch.randn(8, 4, 3, 3)):
        torch.cat([R._memory_order(g, p)], out=flat)
        assert torch.equal(v, g)
    monkeypatch.setattr(apis.dist, 'get_world_size', lambda g=None: 1)
    monkeypatch.setattr(apis.dist, 'broadcast', lambda *a, **k: None)
    net = nn.Sequential(nn.Conv2d(8, 8, 3), nn.Conv2d(8, 16, 3), nn.Linear(64, 4096), nn.Linear(4096, 512))
    red = R(net, bucket_mb=4, tail_mb=1)
    sizes = [b['end'] - b['start'] for b in red.buckets]
    assert sum(sizes) == sum((q.numel() + R.ALIGN - 1) // R.ALIGN * R.ALIGN for q in net.parameters())
    assert red.buckets[0]['params'][0] is list(net.parameters())[-1]          # reverse registration order
    assert sizes[-1] * 4 <= (1 << 20) and red.buckets[-1]['params'][-1] is list(net.parameters())[0]
    for q in net.parameters():
        assert red.views[q].shape == q.shape and red.views[q].stride() == q.stride()


@pytest.mark.timeout(600)
def test_two_rank_step_when_one_rank_has_no_foreground_for_the_contrastive_loss():
    """rank 1's image holds one small object, so its sampled RoIs carry fewer foreground rows than
    ContrastiveLossPlus.min_samples - the state in which the reference drops ``loss_cont`` from the dict
    (contrastive_head.py:123-129; its ranks would then fail the key-count assertion of base.py:258-265).  Here the
    key stays (value 0 on that rank, the rule is applied inside the loss kernel), so both ranks build the same graph:
    the gradient buckets are all-reduced in the SAME order on both ranks, the contrastive branch's parameters
    receive the mean of rank 0's gradient and zero, and the replicas stay identical."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 27500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, False, True)) for r in range(2)]
    for p in procs:
        p.start()
    res = _collect(procs, q, 540)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, p0, g0, lv0, l0, ch0, x0), (_, p1, g1, lv1, l1, ch1, x1) = res
    (order0, fg0, cont0, nb0), (order1, fg1, cont1, nb1) = x0, x1
    assert nb0 == nb1 == 4
    assert fg1 <= 10 < fg0, (fg0, fg1)      # rank 1 is below min_samples (contrastive_head.py:125), rank 0 above
    assert order0 == order1 == list(range(nb0)), (order0, order1)
    assert np.array_equal(g0, g1) and np.array_equal(p0, p1) and np.array_equal(cont0, cont1)
    assert lv0.keys() == lv1.keys() and 'loss_cont' in lv0
    assert np.abs(cont0).max() > 0          # rank 0's contrastive gradient arrived on both ranks (halved by the mean)


def test_bucket_plan_on_the_real_parameter_sets(monkeypatch):
    """FlatGradReducer on the parameters of the two benchmarked detectors (no process group needed
    for the plan).  R50-FPN OA-DG, 64 MiB buckets: 68.9 / 67.2 / 24.4 / 5.4 MB (= 65.7 / 64.1 / 23.3 / 5.2 MiB), the
    RoI head + neck first, the 5.4 MB tail = the first trainable backbone stage (ready last).  R101-DC5 OA-DG with
    OADG_BUCKET_MB=128: FIVE buckets - the 100352 x 1024 FC weight is one 411 MB tensor and cannot be split, so the first
    bucket (RoI head) is 416.7 MB and is the first collective of the backward pass."""
    import oadg_amd  # noqa: F401
    from oadg_amd import Config, apis, build_detector
    monkeypatch.setattr(apis.dist, 'get_world_size', lambda g=None: 1)
    monkeypatch.setattr(apis.dist, 'broadcast', lambda *a, **k: None)
    plans = {}
    for name, cfgf, mb in (('r50', 'configs/oadg/faster_rcnn_r50_fpn_1x_cityscapes_oadg.py', 64),
                           ('dc5', 'configs/oadg/faster_rcnn_r101_dc5_1x_dwd_oadg.py', 128)):
        det = build_detector(Config.fromfile(os.path.join(ROOT, cfgf)).model)
        red = apis.FlatGradReducer(det, bucket_mb=mb)
        names = {p: n for n, p in det.named_parameters()}
        plans[name] = ([round((b['end'] - b['start']) * 4 / 1e6, 1) for b in red.buckets],
                       [(names[b['params'][0]], names[b['params'][-1]]) for b in red.buckets])
        trainable = [p for p in det.parameters() if p.requires_grad]
        pad = lambda n: (n + red.ALIGN - 1) // red.ALIGN * red.ALIGN  # noqa: E731 (slices start on 256-byte boundaries)
        assert sum(b['end'] - b['start'] for b in red.buckets) == sum(pad(p.numel()) for p in trainable) == red.flat.numel()
        assert all(red.views[p].data_ptr() % 256 == red.flat.data_ptr() % 256 for p in trainable)
        assert [p for b in red.buckets for p in b['params']] == trainable[::-1]      # reverse registration order
        for p in trainable:
            assert red.views[p].shape == p.shape and red.views[p].stride() == p.stride()
        del det, red
    sizes, ends = plans['r50']
    assert sizes == [68.9, 67.2, 24.4, 5.4], sizes
    assert ends[0][0].startswith('roi_head.bbox_head.fc_cont') and ends[-1][1] == 'backbone.layer2.0.conv1.weight'
    sizes, ends = plans['dc5']
    assert sizes == [416.7, 151.6, 134.5, 28.8, 5.4], sizes
    assert ends[0][1] == 'roi_head.bbox_head.shared_fcs.0.weight' and ends[-1][1] == 'backbone.layer2.0.conv1.weight'


def test_grad_sink_slice_is_handed_out_once_per_backward_pass():
    """``param.grad`` stays None for EVERY use of a leaf until AccumulateGrad runs, so a head whose forward runs
    twice in one step asked twice for the same bucket slice and autograd summed two aliases (2 * g2 instead of g1 + g2).
    The claim set gives the second request of a pass a tensor of its own; FlatGradReducer.finish / begin_step reset it."""
    from oadg_amd import hip_ops, hip_conv
    w = torch.nn.Parameter(torch.zeros(4, 3))
    view = torch.zeros(4, 3)
    old = hip_ops.GRAD_SINK
    try:
        hip_ops.GRAD_SINK = {w: view}
        hip_ops.sink_reset()
        # the RoI head's cast function twice on the same leaf inside one backward pass
        a1, = hip_ops._CastAll.apply(w)
        a2, = hip_ops._CastAll.apply(w)
        g1, g2 = torch.full((4, 3), 1.0), torch.full((4, 3), 2.0)
        (a1.float() * g1 + a2.float() * g2).sum().backward()
        assert torch.equal(w.grad, g1 + g2), w.grad
        first = hip_ops.grad_dest(torch.nn.Parameter(torch.zeros(2)))          # a parameter without a slice: its own tensor
        assert first.shape == (2,)
        w.grad = None
        assert hip_ops.grad_dest(w).data_ptr() != view.data_ptr()              # still claimed in this pass
        hip_conv.begin_step(False)                                              # the next step: handed out again
        assert hip_ops.grad_dest(w).data_ptr() == view.data_ptr()
        assert hip_ops.grad_dest(w).data_ptr() != view.data_ptr()
    finally:
        hip_ops.GRAD_SINK = old
        hip_ops.sink_reset()
