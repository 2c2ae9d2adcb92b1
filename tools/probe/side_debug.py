import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oadg_amd
from oadg_amd import hip_conv
dev = torch.device('cuda:0')
hip_conv.enable(True)
g = torch.Generator(device=dev).manual_seed(0)
for (C, K, R, stride, pad) in ((256, 512, 1, 2, 0), (256, 512, 1, 1, 0), (128, 128, 3, 2, 1)):
    x = torch.randn(4, C, 64, 96, device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last)
    conv = torch.nn.Conv2d(C, K, R, stride, pad, bias=False).to(dev)
    bn = torch.nn.BatchNorm2d(K).to(dev).eval()
    res = {}
    for side in (True, False, True):
        hip_conv.WGRAD_SIDE_STREAM = side
        conv.zero_grad(); bn.zero_grad()
        with torch.autocast('cuda', dtype=torch.bfloat16):
            y = hip_conv.conv_bn(x, conv, bn)
        gy = torch.randn(y.shape, device=dev, generator=g).bfloat16().contiguous(memory_format=torch.channels_last) if side is True and 'gy' not in res else res['gy']
        res['gy'] = gy
        y.backward(gy)
        hip_conv.join_wgrad_streams(); torch.cuda.synchronize()
        print(C, K, R, stride, 'side' if side else 'main', conv.weight.grad.abs().max().item(), conv.weight.grad.abs().sum().item(),
              bn.weight.grad.abs().sum().item(), bn.bias.grad.abs().sum().item())
