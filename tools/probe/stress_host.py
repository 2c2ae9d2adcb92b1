import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import oadg_amd
from oadg_amd import Config
from oadg_amd.apis import set_random_seed, pin_rank_to_cores
from oadg_amd.pipelines import DevicePipeline, SyntheticCityscapes
pin_rank_to_cores(0, 1)
dev = torch.device('cuda:0')
cfg = Config.fromfile('configs/oadg/faster_rcnn_r50_fpn_1x_cityscapes_oadg.py')
set_random_seed(1)
ds = SyntheticCityscapes(img_shape=(1024, 2048), num_boxes=4096, num_classes=8, box_size=(8, 48), seed=0, device=dev)
pipe = DevicePipeline(cfg.data.train.pipeline, dtype=torch.bfloat16)
pipe.inputs_resident = True
batches = [ds.batch(range(i * 8, (i + 1) * 8)) for i in range(2)]
torch.cuda.synchronize()
np.random.seed(1000)
for i in range(4): pipe(*batches[i % 2])
torch.cuda.synchronize()
import cProfile, pstats
hs = []
t_all = time.time()
for i in range(10):
    t0 = time.time(); pipe(*batches[i % 2]); hs.append((time.time() - t0) * 1e3)
torch.cuda.synchronize()
print('host ms per pipe() call:', [round(h, 1) for h in hs], 'wall per step', round((time.time() - t_all) * 100, 1))
pr = cProfile.Profile(); pr.enable()
for i in range(4): pipe(*batches[i % 2])
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
