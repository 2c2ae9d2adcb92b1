import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
from inputs import model_batch
import oadg_amd
from oadg_amd import Config, build_detector, hip_conv
from oadg_amd.apis import TrainEngine, build_optimizer, set_random_seed
dev = torch.device('cuda:0')
cfg = Config.fromfile(os.path.join(ROOT, 'configs/oadg/faster_rcnn_r50_fpn_1x_cityscapes_oadg.py'))
b = model_batch(3, 2, 256, 512, n_gt=8)
shape = b['img'].shape[2:] + (3,)
t = lambda x: torch.tensor(x, device=dev)
def data():
    return dict(img=t(b['img']).contiguous(memory_format=torch.channels_last), img2=t(b['img2']).contiguous(memory_format=torch.channels_last),
                gt_bboxes=[t(x) for x in b['gt_bboxes']], gt_bboxes2=[t(x) for x in b['gt_bboxes']], gt_labels=[t(x) for x in b['gt_labels']],
                multilevel_boxes=[torch.tensor(x) for x in b['multilevel_boxes']], oamix_boxes=[torch.tensor(x) for x in b['oamix_boxes']],
                img_metas=[dict(img_shape=shape, pad_shape=shape, ori_shape=shape, scale_factor=1.0, flip=False) for _ in range(2)])
hip_conv.enable(True)
out = {}
for mode in ('single', 'side', 'side2'):
    os.environ['OADG_WGRAD_STREAM'] = '0' if mode.startswith('single') else '1'
    set_random_seed(0)
    det = build_detector(cfg.model); det.init_weights()
    det = det.to(dev).to(memory_format=torch.channels_last).train()
    eng = TrainEngine(det, build_optimizer(det, cfg.optimizer), amp_dtype=torch.bfloat16)
    set_random_seed(1)
    (loss, lv), _ = eng.forward_losses(data())
    loss.backward(); hip_conv.join_wgrad_streams(); torch.cuda.synchronize()
    out[mode] = {n: p.grad.float().clone() for n, p in det.named_parameters() if p.grad is not None}
    print(mode, float(loss))
import itertools
for a, b_ in itertools.combinations(out.keys(), 2):
    bad = []
    for n, g in out[a].items():
        d = (out[b_][n] - g).abs().max().item(); r = g.abs().max().item()
        if d > 0.05 * r + 1e-9:
            bad.append((n, d, r))
    print(a, b_, 'bad', len(bad))
    for x in bad:
        print('    %-50s d=%.2e ref=%.2e' % x)
