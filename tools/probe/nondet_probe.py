"""Which stage of the fp32 GPU step is non-deterministic from run to run?"""
import os, sys, hashlib, collections
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ('', 'tests', 'tests/golden'):
    sys.path.insert(0, os.path.join(ROOT, p))
import oadg_amd
from test_model_parity import build_and_load, make_data
from oadg_amd.detectors import integrate_data
dev = torch.device("cuda:0")
torch.backends.cudnn.deterministic = bool(int(os.environ.get("DET", "0")))
g = np.load(os.path.join(ROOT, 'tests/golden/model_step_256x512.npz'))
det = build_and_load(dev)
h = lambda t: hashlib.md5(t.detach().float().cpu().numpy().tobytes()).hexdigest()[:8]
seen = collections.defaultdict(set)
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 25):
    data = integrate_data(make_data(g, dev), det.train_cfg)
    torch.manual_seed(int(g['seed'])); np.random.seed(int(g['seed']))
    with torch.no_grad():
        x = det.extract_feat(data['img'])
        for i, f in enumerate(x):
            seen[f'feat{i}'].add(h(f))
        outs = det.rpn_head(x)
        for i, (c, r) in enumerate(zip(*outs)):
            seen[f'rpn_cls{i}'].add(h(c)); seen[f'rpn_reg{i}'].add(h(r))
        props = det.rpn_head.get_bboxes(*outs, img_metas=data['img_metas'], cfg=det.train_cfg.rpn_proposal)
        for i, p in enumerate(props):
            seen[f'prop{i}'].add(h(p) + f':{p.shape[0]}')
        props = det.rpn_head.get_bboxes(*outs, img_metas=data['img_metas'], cfg=det.train_cfg.rpn_proposal, padded=True)
        for i, p in enumerate(props):
            seen[f'pprop{i}'].add(h(p[p[:, 4] >= 0]) + f':{int((p[:, 4] >= 0).sum())}')
    out = det.train_step(make_data(g, dev), None)
    seen['loss_cls'].add(round(out['log_vars']['loss_cls'], 6))
    seen['loss_rpn_cls'].add(round(out['log_vars']['loss_rpn_cls'], 7))
for k, v in seen.items():
    if len(v) > 1 or k.startswith('loss'):
        print(k, len(v), sorted(v)[:4])
print('done')
