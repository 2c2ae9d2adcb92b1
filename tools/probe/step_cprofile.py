"""cProfile of the MAIN thread over steady-state training steps of the benchmarked configuration, set up exactly as
bench.py does (worker-thread pipeline, speculative device sampler): where the ~23 ms of host time per step go.   (GPU)
The device is kept BEHIND the host (no synchronisation inside the loop), so waits show up in the functions that wait."""
import cProfile
import os
import pstats
import sys
import time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oadg_amd import Config, build_detector, hip_conv  # noqa: E402
from oadg_amd.apis import TrainEngine, build_optimizer, set_random_seed  # noqa: E402
from oadg_amd.pipelines import DevicePipeline, SyntheticCityscapes  # noqa: E402

dev = torch.device('cuda:0')
hip_conv.enable()
cfg = Config.fromfile(bench.CFG)
set_random_seed(0)
det = build_detector(cfg.model)
det.init_weights(allow_missing_pretrained=True)
det = det.to(dev).to(memory_format=torch.channels_last).train()
det.log_vars_on_host = False
eng = TrainEngine(det, build_optimizer(det, cfg.optimizer), distributed=False, amp_dtype=torch.bfloat16)
ds = SyntheticCityscapes(img_shape=(1024, 2048), num_boxes=20, num_classes=8, seed=0, device=dev)
pipe = DevicePipeline(cfg.data.train.pipeline, dtype=torch.bfloat16)
batches = [ds.batch(range(i * 4, i * 4 + 4)) for i in range(6)]
nxt = pipe.prefetch(*batches[0], worker_seed=1000)
N = int(os.environ.get('N', 20))
prof = os.environ.get('PROFILE', '1') == '1'
pr = cProfile.Profile()
for i in range(N + 8):
    if i == 8:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if prof:
            pr.enable()
    data = nxt.get()
    nxt = pipe.prefetch(*batches[(i + 1) % 6], worker_seed=1000)
    eng.step(data)
if prof:
    pr.disable()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'host loop {(t1 - t0) / N * 1e3:.2f} ms/step, with the final drain {(t2 - t0) / N * 1e3:.2f} ms/step (profile {prof}); divide the tables by {N}')
if prof:
    pstats.Stats(pr).sort_stats('cumulative').print_stats(70)
    pstats.Stats(pr).sort_stats('tottime').print_stats(60)
