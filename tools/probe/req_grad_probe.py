"""Which backbone / neck tensors require grad in the benchmarked training configuration?  (GPU)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oadg_amd import Config, build_detector, hip_conv
import bench

cfg = Config.fromfile(bench.CFG)
det = build_detector(cfg.model).cuda()
det.init_weights(allow_missing_pretrained=True)
det.train()
hip_conv.enable(True)
x = torch.randn(2, 3, 256, 512, device='cuda').contiguous(memory_format=torch.channels_last)
with torch.autocast('cuda', dtype=torch.bfloat16):
    feats = det.backbone(x)
    print('backbone outputs require grad:', [f.requires_grad for f in feats], [tuple(f.shape) for f in feats])
    outs = det.neck(feats)
    print('neck outputs require grad:', [f.requires_grad for f in outs])
print('frozen params in layer1:', all(not p.requires_grad for p in det.backbone.layer1.parameters()))
