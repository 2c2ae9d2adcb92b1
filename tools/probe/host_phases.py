"""host time of the phases of a training step (main thread, no synchronisation inside the loop)   (GPU)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oadg_amd import Config, build_detector, hip_conv
from oadg_amd.apis import TrainEngine, build_optimizer, set_random_seed
from oadg_amd.pipelines import DevicePipeline, SyntheticCityscapes
import bench

dev = torch.device('cuda:0')
hip_conv.enable()
cfg = Config.fromfile(bench.CFG)
set_random_seed(0)
det = build_detector(cfg.model)
det.init_weights(allow_missing_pretrained=True)
det = det.to(dev).to(memory_format=torch.channels_last).train()
det.log_vars_on_host = False
eng = TrainEngine(det, build_optimizer(det, cfg.optimizer), distributed=False, amp_dtype=torch.bfloat16)
ds = SyntheticCityscapes(device=dev)
pipe = DevicePipeline(cfg.data.train.pipeline, dtype=torch.bfloat16)
batches = [ds.batch(range(i * 4, i * 4 + 4)) for i in range(3)]
nxt = pipe.prefetch(*batches[0])
acc = {}
def add(k, t): acc[k] = acc.get(k, 0.0) + t
N = 30
for i in range(N + 5):
    if i == 5:
        torch.cuda.synchronize(); acc.clear(); T0 = time.perf_counter()
    t = time.perf_counter(); data = nxt.get(); add('get', time.perf_counter() - t)
    t = time.perf_counter(); nxt = pipe.prefetch(*batches[(i + 1) % 3]); add('prefetch', time.perf_counter() - t)
    t = time.perf_counter(); eng.optimizer.zero_grad(set_to_none=True); add('zero_grad', time.perf_counter() - t)
    t = time.perf_counter(); (loss, lv), n = eng.forward_losses(data); add('forward', time.perf_counter() - t)
    t = time.perf_counter(); loss.backward(); add('backward', time.perf_counter() - t)
    t = time.perf_counter(); eng.optimizer.step(); add('optimizer', time.perf_counter() - t)
torch.cuda.synchronize()
tot = (time.perf_counter() - T0) / N * 1e3
print(f'step {tot:.2f} ms; host ms per step: ' + ', '.join(f'{k} {v / N * 1e3:.2f}' for k, v in acc.items()) +
      f'; sum {sum(acc.values()) / N * 1e3:.2f}')
