"""run-to-run reproducibility of one bf16 MFMA train step from identical seeds (tools/probe)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import oadg_amd  # noqa: E402,F401
from oadg_amd import Config  # noqa: E402
from oadg_amd.apis import TrainEngine, build_optimizer, set_random_seed  # noqa: E402
from oadg_amd.pipelines import DevicePipeline, SyntheticCityscapes  # noqa: E402
from test_model_parity import build_and_load, CFG  # noqa: E402

dev = torch.device('cuda:0')
cfg = Config.fromfile(CFG)
det = build_and_load(dev).to(memory_format=torch.channels_last).train()
eng = TrainEngine(det, build_optimizer(det, cfg.optimizer), amp_dtype=torch.bfloat16)
H, W = (int(v) for v in os.environ.get('HW', '512x1024').split('x'))
ds = SyntheticCityscapes(img_shape=(H, W), num_boxes=20, device=dev)
pipe = DevicePipeline(cfg.data.train.pipeline, dtype=torch.bfloat16)
set_random_seed(5)
data = pipe(*ds.batch(range(4)))
captured = {}
head = det.roi_head.bbox_head
orig = head.forward


def spy(x):
    out = orig(x)
    captured['x'] = float(x.float().double().sum())
    captured['out'] = [float(o.float().double().sum()) for o in out]
    return out
head.forward = spy
neck = det.neck
norig = neck.forward


def nspy(inputs):
    outs = norig(inputs)
    captured['bb'] = [float(t.float().double().sum()) for t in inputs]
    captured['neck'] = [(float(o[:4].float().double().sum()), float(o[4:].float().double().sum())) for o in outs]
    return outs
neck.forward = nspy
ext = det.roi_head.bbox_roi_extractor
eorig = ext.forward


def espy(feats, rois, *a, **k):
    out = eorig(feats, rois, *a, **k)
    captured['rois'] = (tuple(rois.shape), float(rois.double().sum()))
    captured['rois_t'] = rois.detach().clone()
    captured['roi_out_t'] = out.detach().float().clone()
    n1 = 2048
    captured['roi_out'] = [float(out[i * n1:(i + 1) * n1].float().double().sum()) for i in range(3)]
    return out
ext.forward = espy
rh = det.roi_head
forig = rh._bbox_forward_train


def fspy(x, sampling_results, *a, **k):
    captured['sr'] = [(id(r) % 100000, float(r.bboxes.double().sum()), float(r.neg_bboxes.double().sum()), r.neg_inds.data_ptr() % 1000000,
                       r._src[0].data_ptr() % 1000000, int(r.neg_inds.sum())) for r in sampling_results]
    torch.cuda.synchronize()
    captured['sr2'] = [(float(r.bboxes.double().sum()), int(r.neg_inds.sum())) for r in sampling_results]
    return forig(x, sampling_results, *a, **k)
rh._bbox_forward_train = fspy
prev = None
for rep in range(4):
    set_random_seed(11)
    det.zero_grad(set_to_none=True)
    (loss, lv), n = eng.forward_losses(data)
    loss.backward()
    torch.cuda.synchronize()
    if prev is not None:
        d = (prev[0] != captured['rois_t']).any(1).nonzero().flatten()
        print('rois rows that differ from the previous run:', d.numel(), d[:10].tolist(), prev[0][d[:3]].tolist(), captured['rois_t'][d[:3]].tolist())
        d2 = (prev[1] != captured['roi_out_t']).flatten(1).any(1).nonzero().flatten()
        print('roi feature rows that differ:', d2.numel(), d2[:10].tolist())
    print('sampling results', captured['sr'])
    print('again after sync', captured['sr2'])
    r_ = captured['rois_t']
    print('view-2 rois replicate view-1 rois:', bool(torch.equal(r_[:2048, 1:], r_[2048:4096, 1:])), 'rows equal', int((r_[:2048, 1:] == r_[2048:4096, 1:]).all(1).sum()), 'batch idx', r_[::512, 0].tolist())
    prev = (captured['rois_t'], captured['roi_out_t'])
    g = sum(float(p.grad.double().abs().sum()) for p in det.parameters() if p.grad is not None)
    print(rep, 'backbone', captured['bb'], 'neck', captured['neck'], 'rois', captured['rois'], 'roi_out', captured['roi_out'])
    print(rep, {k: round(float(v), 6) for k, v in lv.items()}, 'roi feats sum', captured['x'], 'head outs', captured['out'],
          'labels sum', int(det.roi_head.bbox_targets[0].sum()), 'grad abs sum', g)
