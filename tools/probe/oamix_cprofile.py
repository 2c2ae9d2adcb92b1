"""cProfile of the host side of one OA-Mix pipeline pass at BASELINE configs[4] (4096 boxes, bs 8), one worker   (GPU)"""
import cProfile
import os
import pstats
import sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oadg_amd  # noqa: F401,E402
from oadg_amd import Config  # noqa: E402
from oadg_amd.pipelines import DevicePipeline, SyntheticCityscapes  # noqa: E402

dev = torch.device('cuda:0')
cfg = Config.fromfile(os.path.join(ROOT, 'configs', 'oadg', 'faster_rcnn_r50_fpn_1x_cityscapes_oadg.py'))
n_boxes, box_size, batch = (4096, (8, 48), 8) if '--config2' not in sys.argv else (20, (24, 400), 4)
ds = SyntheticCityscapes(img_shape=(1024, 2048), num_boxes=n_boxes, num_classes=8, box_size=box_size, device=dev)
imgs, boxes, labels = ds.batch(range(batch))
pipe = DevicePipeline(cfg.data.train.pipeline, dtype=torch.bfloat16, oamix_workers=1)
np.random.seed(0)
pipe(imgs, boxes, labels)
torch.cuda.synchronize()
np.random.seed(1)
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    pipe(imgs, boxes, labels)
pr.disable()
torch.cuda.synchronize()
print(f'(per view: divide by {3 * batch})')
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
pstats.Stats(pr).sort_stats('tottime').print_stats(30)
