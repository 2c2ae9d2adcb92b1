"""which trainable parameters receive an all-zero gradient in one bf16 MFMA step?  (tools/probe)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oadg_amd  # noqa: E402,F401
from oadg_amd import Config, build_detector, hip_conv  # noqa: E402
from oadg_amd.apis import TrainEngine, build_optimizer, set_random_seed  # noqa: E402
from oadg_amd.pipelines import DevicePipeline, SyntheticCityscapes  # noqa: E402

dev = torch.device('cuda:0')
H, W = (int(v) for v in os.environ.get('HW', '1024x2048').split('x'))
cfg = Config.fromfile(os.path.join(ROOT, 'configs/oadg/faster_rcnn_r50_fpn_1x_cityscapes_oadg.py'))
set_random_seed(0)
det = build_detector(cfg.model)
det.init_weights(allow_missing_pretrained=True)
det = det.to(dev).to(memory_format=torch.channels_last).train()
eng = TrainEngine(det, build_optimizer(det, cfg.optimizer), amp_dtype=torch.bfloat16)
ds = SyntheticCityscapes(img_shape=(H, W), num_boxes=20, device=dev)
pipe = DevicePipeline(cfg.data.train.pipeline, dtype=torch.bfloat16)
nb = int(os.environ.get('BS', '4'))
for it in range(2):
    data = pipe(*ds.batch(range(it * nb, it * nb + nb)))
    det.zero_grad(set_to_none=True)
    (loss, lv), n = eng.forward_losses(data)
    loss.backward()
    torch.cuda.synchronize()
    zero = [(n_, tuple(p.shape)) for n_, p in det.named_parameters() if p.requires_grad and (p.grad is None or float(p.grad.float().abs().sum()) == 0)]
    print('iter', it, 'loss', float(loss), 'zero-gradient parameters:', zero[:10], len(zero))
