"""a frozen identity bottleneck block of ResNet stage 1 at the benchmark's size (8 x 256 x 512 x 256 channels): the fused
launch (csrc/bottleneck_frozen.hip) against the three convolution launches; every launch after a 1 GiB fill (operands from
HBM, as inside the training step)   (GPU)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oadg_amd  # noqa: E402,F401
from oadg_amd import hip_conv  # noqa: E402
from oadg_amd.backbones import Bottleneck  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(0)
first = '--first' in sys.argv          # the stage's first block (64 input channels, downsample convolution)
if first:
    from oadg_amd.backbones import make_res_layer
    blk = make_res_layer(64, 64, 1, 1, 1, 'pytorch', dict(type='BN'))[0].to(dev).eval()
else:
    blk = Bottleneck(256, 64).to(dev).eval()
for p in blk.parameters():
    p.requires_grad_(False)
x = torch.randn(8, 64 if first else 256, 256, 512, device=dev).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
flush = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
hip_conv.enable(True)
for fused in (True, False, True, False):
    hip_conv.FUSED_FROZEN_BLOCK = fused
    with torch.autocast('cuda', dtype=torch.bfloat16):
        for _ in range(2):
            y = blk(x)
        ev = []
        for i in range(8):
            flush.fill_(i)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); y = blk(x); b.record()
            ev.append((a, b))
        torch.cuda.synchronize()
    us = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)[len(ev) // 2]
    gb = (x.numel() + y.numel()) * 2 / 1e9
    print(f'fused={fused}: {us:7.1f} us per block  ({gb / us * 1e6 / 1e3:.2f} TB/s of the {gb:.2f} GB a fused block has to move)', flush=True)
