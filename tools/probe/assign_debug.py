import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
import oadg_amd
from test_model_parity import build_and_load, make_data
from oadg_amd.core.bbox import MaxIoUAssigner
dev = torch.device('cuda:0')
g = np.load(os.path.join(ROOT, 'tests/golden/model_step_256x512.npz'))
orig = MaxIoUAssigner.assign_many
res = {}
for mode in ('many', 'loop'):
    MaxIoUAssigner.assign_many = orig if mode == 'many' else (lambda self, *a, **k: None)
    det = build_and_load(dev)
    data = make_data(g, dev)
    torch.manual_seed(int(g['seed'])); np.random.seed(int(g['seed']))
    out = det.train_step(data, None)
    torch.cuda.synchronize()
    res[mode] = (det.roi_head.bbox_targets[0].cpu().numpy(), {k: v for k, v in out['log_vars'].items()})
    print(mode, (res[mode][0] == g['roi_labels']).mean(), res[mode][1])
# direct comparison of the RoI assignment on the same proposals
det = build_and_load(dev)
asg = det.roi_head.bbox_assigner
gen = torch.Generator(device=dev).manual_seed(0)
props = [torch.rand(1000, 5, generator=gen, device=dev) * 200 for _ in range(2)]
for p in props:
    p[:, 2:4] += p[:, :2]
    p[900:, :] = 0; p[900:, 4] = -1
gts = [data['gt_bboxes'][i] for i in range(2)]
gls = [data['gt_labels'][i] for i in range(2)]
MaxIoUAssigner.assign_many = orig
ars, cnt = asg.assign_many(props, [p[:, 4] >= 0 for p in props], gts, gls)
for i in range(2):
    ref = asg.assign_masked(props[i][:, :4], props[i][:, 4] >= 0, gts[i], gls[i])
    print(i, torch.equal(ars[i].gt_inds, ref.gt_inds), torch.equal(ars[i].labels, ref.labels), cnt[i].tolist(),
          int((ref.gt_inds > 0).sum()), int((ref.gt_inds == 0).sum()))
