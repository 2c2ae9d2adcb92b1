"""Host-side timing of the reducer path at world size 1 (where does the host block?)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29544')
import oadg_amd  # noqa
from oadg_amd import Config, build_detector, hip_conv
from oadg_amd.apis import TrainEngine, build_optimizer, init_dist, set_random_seed
from oadg_amd.pipelines import DevicePipeline, SyntheticCityscapes

dev = torch.device('cuda:0')
hip_conv.enable()
cfg = Config.fromfile(os.path.join(ROOT, 'configs/oadg/faster_rcnn_r50_fpn_1x_cityscapes_oadg.py'))
set_random_seed(0)
det = build_detector(cfg.model); det.init_weights(allow_missing_pretrained=True)
det = det.to(dev).to(memory_format=torch.channels_last).train(); det.log_vars_on_host = False
ddp = os.environ.get('DDP', '1') == '1'
if ddp:
    init_dist('pytorch', backend='nccl')
eng = TrainEngine(det, build_optimizer(det, cfg.optimizer), distributed=ddp, amp_dtype=torch.bfloat16)
ds = SyntheticCityscapes(device=dev)
pipe = DevicePipeline(cfg.data.train.pipeline, dtype=torch.bfloat16)
batches = [ds.batch(range(i * 4, i * 4 + 4)) for i in range(3)]
T = {}
def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); T[name] = T.get(name, 0.0) + time.perf_counter() - t; return r
    return w
if eng.reducer is not None:
    eng.reducer.finish = timed('finish', eng.reducer.finish)
    eng.reducer._launch = timed('launch', eng.reducer._launch)
eng.optimizer.step = timed('opt', eng.optimizer.step)
eng.forward_losses = timed('fwd', eng.forward_losses)
nxt = pipe.prefetch(*batches[0], worker_seed=5)
for i in range(30):
    if i == 10:
        torch.cuda.synchronize(); T.clear(); t0 = time.perf_counter()
    data = nxt.get()
    nxt = pipe.prefetch(*batches[(i + 1) % 3], worker_seed=5)
    ts = time.perf_counter()
    eng.step(data)
    T['step'] = T.get('step', 0.0) + time.perf_counter() - ts
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / 20
print(f'ddp={ddp} {tot * 1e3:.2f} ms/step; host per step: ' + ', '.join(f'{k} {v / 20 * 1e3:.2f}' for k, v in T.items()))
