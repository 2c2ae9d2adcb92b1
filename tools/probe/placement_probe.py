"""Does the 256-tile convolution's rate depend on WHERE the allocator put the tensors?  Same process, same engine: cycles of
N steps with live event pairs on the 256-tile launches; between cycles the caching allocator's blocks are returned to the
driver (torch.cuda.empty_cache) so the next cycle's tensors land on other physical pages.   (GPU)
Prints per cycle: ms per step, the P2 3x3 launch (mean over the cycle), the 256-tile family per step, and the addresses of the
largest saved activations."""
import os
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
N = int(os.environ.get('N', 16))
CYCLES = int(os.environ.get('CYCLES', 5))
step_no = 0
for cyc in range(CYCLES):
    for i in range(4):                       # re-warm (allocations of the cycle)
        data = nxt.get(); nxt = pipe.prefetch(*batches[(step_no + 1) % 6], worker_seed=1000); eng.step(data); step_no += 1
    torch.cuda.synchronize()
    hip_conv.TIMERS, hip_conv.TIMERS_ONLY_VARIANT = [], (2,)
    t0 = time.perf_counter()
    for i in range(N):
        data = nxt.get(); nxt = pipe.prefetch(*batches[(step_no + 1) % 6], worker_seed=1000); eng.step(data); step_no += 1
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / N * 1e3
    tm, hip_conv.TIMERS = hip_conv.TIMERS, None
    fam = sum(t[0].elapsed_time(t[1]) for t in tm) / N
    p2 = [t[0].elapsed_time(t[1]) for t in tm if tuple(t[5][:7]) == (8, 256, 512, 256, 256, 3, 1) and not t[5][7]]
    st = torch.cuda.memory_stats()
    print(f'cycle {cyc}: {dt:6.2f} ms/step with event pairs; 256-tile forward / data gradient {fam:6.3f} ms/step; P2 3x3 launch '
          f'{sum(p2) / len(p2):.4f} ms (min {min(p2):.4f} max {max(p2):.4f}); reserved {st["reserved_bytes.all.current"] >> 20} MB, '
          f'segments {st["segment.all.current"]}', flush=True)
    nxt.get()
    torch.cuda.synchronize()
    if os.environ.get('EMPTY', '1') == '1':
        torch.cuda.empty_cache()
    nxt = pipe.prefetch(*batches[step_no % 6], worker_seed=1000)
