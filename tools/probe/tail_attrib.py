"""Which Python lines launch the training stream's small kernels: one torch-profiled step (with_stack), device kernels
under 25 us grouped by (kernel, launching op, input shapes, innermost package frame)   (GPU)
usage: python tools/probe/tail_attrib.py [substring of kernel names, default: all small kernels]"""
import collections
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import oadg_amd  # noqa: F401,E402
from oadg_amd import Config, build_detector, hip_conv  # noqa: E402
from oadg_amd.apis import TrainEngine, build_optimizer, set_random_seed  # noqa: E402
from profile_step import synth  # noqa: E402

pat = sys.argv[1] if len(sys.argv) > 1 else ''
dev = torch.device('cuda:0')
hip_conv.enable()
cfg = Config.fromfile(os.path.join(ROOT, 'configs/oadg/faster_rcnn_r50_fpn_1x_cityscapes_oadg.py'))
det = build_detector(cfg.model)
det.init_weights(allow_missing_pretrained=True)
det = det.to(dev).to(memory_format=torch.channels_last).train()
set_random_seed(0)
eng = TrainEngine(det, build_optimizer(det, cfg.optimizer), amp_dtype=torch.bfloat16)
for _ in range(3):
    eng.step(synth(4, 1024, 2048, dev))
torch.cuda.synchronize()
b = synth(4, 1024, 2048, dev)
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as p:
    eng.step(b)
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in p.events():
    ks = getattr(e, 'kernels', None)
    if not ks:
        continue
    for k in ks:
        if k.duration > 25 or pat not in k.name:
            continue
        frame = next((s for s in (e.stack or []) if 'oa-dg_amd' in s or 'oadg_amd' in s), (e.stack or ['?'])[0] if e.stack else '?')
        frame = frame.replace(ROOT, '')
        key = (k.name.replace('(anonymous namespace)::', '').replace('void ', '')[:48], e.name[:28],
               str(e.input_shapes)[:60], frame[:90])
        agg[key][0] += 1
        agg[key][1] += k.duration
tot = sum(v[1] for v in agg.values())
print(f'{sum(v[0] for v in agg.values())} launches under 25 us, {tot / 1e3:.3f} ms')
for key, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:110]:
    print(f'{n:4d} x {t / n:5.1f} us = {t / 1e3:6.3f} ms  {key[0]:48s} {key[1]:28s} {key[2]:60s} {key[3]}')
