"""wall time of the phases of one OA-Mix pipeline pass at BASELINE configs[4] (4096 boxes, bs 8): image states, planning
(record) and lockstep execution, with 1 and 4 planning threads   (GPU)"""
import os
import sys
import time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oadg_amd  # noqa: F401,E402
from oadg_amd import Config  # noqa: E402
from oadg_amd.pipelines import DevicePipeline, SyntheticCityscapes  # noqa: E402
from oadg_amd.pipelines import oa_mix, device_pipeline  # noqa: E402

dev = torch.device('cuda:0')
cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'oadg', 'faster_rcnn_r50_fpn_1x_cityscapes_oadg.py'))
n_boxes, box_size, batch = (4096, (8, 48), 8) if '--config2' not in sys.argv else (20, (24, 400), 4)
ds = SyntheticCityscapes(img_shape=(1024, 2048), num_boxes=n_boxes, num_classes=8, box_size=box_size, device=dev)
imgs, boxes, labels = ds.batch(range(batch))
T = {}


def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            T[name] = T.get(name, 0.0) + time.perf_counter() - t0
    return w


oa_mix._ImageState.__init__ = timed('image_state', oa_mix._ImageState.__init__)
oa_mix.OAMix.execute = timed('execute', oa_mix.OAMix.execute)
oa_mix.OAMix.record = timed('record (sum over threads)', oa_mix.OAMix.record)
oa_mix.OAMix._bbox_chain_c = timed('  of which _bbox_chain_c', oa_mix.OAMix._bbox_chain_c)
for workers in (1, 4):
    pipe = DevicePipeline(cfg.data.train.pipeline, dtype=torch.bfloat16, oamix_workers=workers)
    np.random.seed(0)
    pipe(imgs, boxes, labels)
    torch.cuda.synchronize()
    T.clear()
    np.random.seed(1)
    t0 = time.perf_counter()
    for _ in range(2):
        pipe(imgs, boxes, labels)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    v = 2 * batch
    print(f'workers {workers}: host {t_host / v * 1e3:.2f} ms/view, with device {t_all / v * 1e3:.2f}; ' +
          ', '.join(f'{k} {x / v * 1e3:.2f}' for k, x in T.items()), flush=True)
