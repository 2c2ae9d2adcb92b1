"""fixed cost of a single-round launch of the 256-tile convolution kernel: time against the number of K-tiles (3x3 on the
layer3 map, 256 workgroups, C = 64 ... 1024 -> 9 ... 144 K-tiles) and for 1, 2, 4 rounds (larger maps)   (GPU)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oadg_amd  # noqa: E402,F401
from oadg_amd import hip_conv  # noqa: E402
dev = torch.device('cuda:0')
cl = dict(memory_format=torch.channels_last)


def t_us(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


for (N, H, W) in ((8, 64, 128), (8, 128, 128), (8, 128, 256)):
    for C in (64, 128, 256, 512, 1024):
        x = torch.randn(N, C, H, W, device=dev).bfloat16().contiguous(**cl)
        w = (torch.randn(256, C, 3, 3, device=dev) / (9 * C) ** 0.5).bfloat16().contiguous(**cl)
        us = t_us(lambda: hip_conv.conv_forward(x, w, None, None, 1, 1, 1, False, variant=2))
        nk = 9 * C // 64
        wgs = N * H * W // 256
        print(f'M {N * H * W:7d} ({wgs} workgroups, {wgs / 256:.0f} rounds) C {C:5d} K-tiles {nk:4d}: {us:7.1f} us  '
              f'{us / (nk * wgs / 256):6.3f} us per K-tile-round  {2.0 * N * H * W * 256 * C * 9 / us / 1e6:7.1f} TFLOP/s', flush=True)
