import os, sys, time, json
sys.argv = ['bench.py', '--steps', '1', '--warmup', '0', '--no-cpu-baseline']
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import bench
# replicate bench.main's setup but time segments of 10 steps
a = bench.parse()
import oadg_amd
from oadg_amd import Config, build_detector, hip_conv
from oadg_amd.apis import TrainEngine, build_optimizer, set_random_seed
from oadg_amd.pipelines import DevicePipeline, SyntheticCityscapes
dev = torch.device('cuda', 0)
cfg = Config.fromfile(bench.CFG)
hip_conv.enable()
set_random_seed(0)
det = build_detector(cfg.model); det.init_weights(allow_missing_pretrained=True)
det = det.to(dev).to(memory_format=torch.channels_last).train(); det.log_vars_on_host = False
engine = TrainEngine(det, build_optimizer(det, cfg.optimizer), amp_dtype=torch.bfloat16)
set_random_seed(1)
ds = SyntheticCityscapes(img_shape=(1024, 2048), num_boxes=20, num_classes=8, seed=0, device=dev)
pipe = DevicePipeline(cfg.data.train.pipeline, dtype=torch.bfloat16)
batches = [ds.batch(range(i * 4, (i + 1) * 4)) for i in range(6)]
torch.cuda.synchronize()
nxt = pipe.prefetch(*batches[0], worker_seed=1000)
out = []
for seg in range(14):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(10):
        data = nxt.get()
        nxt = pipe.prefetch(*batches[(seg * 10 + i + 1) % 6], worker_seed=1000)
        engine.step(data)
    torch.cuda.synchronize()
    out.append(round((time.perf_counter() - t0) * 100, 2))
print('ms/step per 10-step segment:', out)
