import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oadg_amd
from oadg_amd import hip_conv
from oadg_amd.dense_heads import RPNHead
dev = torch.device('cuda:0')
torch.manual_seed(0)
head = RPNHead(in_channels=256, feat_channels=256,
               anchor_generator=dict(type='AnchorGenerator', scales=[8], ratios=[0.5, 1.0, 2.0], strides=[4]),
               loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
               loss_bbox=dict(type='L1Loss', loss_weight=1.0)).to(dev)
for m in (head.rpn_conv, head.rpn_cls, head.rpn_reg):
    torch.nn.init.normal_(m.weight, 0, 0.05); torch.nn.init.normal_(m.bias, 0, 0.1)
g = torch.Generator(device=dev).manual_seed(1)
x0 = torch.randn(2, 256, 40, 56, device=dev, generator=g).contiguous(memory_format=torch.channels_last)
gc = torch.randn(2, 3, 40, 56, device=dev, generator=g); gr = torch.randn(2, 12, 40, 56, device=dev, generator=g)
res = {}
for mode in ('fused', 'separate', 'fp32'):
    hip_conv.enable(mode == 'fused')
    head.zero_grad(set_to_none=True)
    x = x0.clone().requires_grad_(True)
    if mode == 'fp32':
        cls, reg = head.forward_single(x)
    else:
        with torch.autocast('cuda', dtype=torch.bfloat16):
            cls, reg = head.forward_single(x)
    ((cls.float() * gc).sum() + (reg.float() * gr).sum()).backward()
    res[mode] = (x.grad.float().clone(), {n: p.grad.float().clone() for n, p in head.named_parameters()})
    hip_conv.enable(False)
ref = res['fp32']
for mode in ('fused', 'separate'):
    d = (res[mode][0] - ref[0]).abs()
    print(mode, 'x.grad max err', d.max().item(), 'mean err', d.mean().item(), 'ref max', ref[0].abs().max().item(), 'ref mean', ref[0].abs().mean().item())
    for n in ref[1]:
        d = (res[mode][1][n] - ref[1][n]).abs()
        print('   ', n, d.max().item() / ref[1][n].abs().max().item(), d.mean().item() / ref[1][n].abs().mean().item())
