"""Micro-benchmark: hand-written MFMA implicit-GEMM conv vs torch.conv2d (MIOpen) on the R50-FPN shapes
(SURVEY.md App. B), bf16 NHWC, HIP-event timing."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oadg_amd  # noqa: E402
from oadg_amd import hip_conv  # noqa: E402

SHAPES = [  # name, N, C, H, W, K, R, stride, pad
    ('FPN/RPN 3x3 P2', 8, 256, 256, 512, 256, 3, 1, 1),
    ('FPN/RPN 3x3 P3', 8, 256, 128, 256, 256, 3, 1, 1),
    ('FPN/RPN 3x3 P4', 8, 256, 64, 128, 256, 3, 1, 1),
    ('layer2 3x3', 8, 128, 128, 256, 128, 3, 1, 1),
    ('layer3 3x3', 8, 256, 64, 128, 256, 3, 1, 1),
    ('layer4 3x3', 8, 512, 32, 64, 512, 3, 1, 1),
    ('layer2 1x1 512->128', 8, 512, 128, 256, 128, 1, 1, 0),
    ('layer2 1x1 128->512', 8, 128, 128, 256, 512, 1, 1, 0),
    ('layer3 1x1 1024->256', 8, 1024, 64, 128, 256, 1, 1, 0),
    ('layer3 1x1 256->1024', 8, 256, 64, 128, 1024, 1, 1, 0),
    ('lateral P2 1x1', 8, 256, 256, 512, 256, 1, 1, 0),
    ('layer4 1x1 2048->512', 8, 2048, 32, 64, 512, 1, 1, 0),
    ('layer4 1x1 512->2048', 8, 512, 32, 64, 2048, 1, 1, 0),
]


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    dev = torch.device('cuda:0')
    print(f'{"shape":26s} {"GFLOP":>8s} {"ours ms":>9s} {"TF/s":>7s} {"torch ms":>9s} {"TF/s":>7s}  speedup')
    for name, N, C, H, W, K, R, st, pad in SHAPES:
        x = torch.randn(N, C, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
        w = (torch.randn(K, C, R, R, device=dev) / (C * R * R) ** 0.5).bfloat16().contiguous(memory_format=torch.channels_last)
        b = torch.randn(K, device=dev)
        Ho, Wo = (H + 2 * pad - R) // st + 1, (W + 2 * pad - R) // st + 1
        gf = 2.0 * N * Ho * Wo * K * C * R * R / 1e9
        t1 = timeit(lambda: hip_conv.conv_forward(x, w, b, None, st, pad, 1, False, variant=1))
        t256 = timeit(lambda: hip_conv.conv_forward(x, w, b, None, st, pad, 1, False, variant=2)) if K % 256 == 0 \
            else float('nan')
        t1s = timeit(lambda: hip_conv.conv_forward(x, w, b, None, st, pad, 1, False, variant=3))
        bb = b.bfloat16()
        t2 = timeit(lambda: F.conv2d(x, w, bb, st, pad))
        line = f'{name:26s} {gf:8.1f} {t1:9.3f} {gf / t1:7.1f} {t2:9.3f} {gf / t2:7.1f}  {t2 / t1:5.2f}x | 256-tile {t256:7.3f} ms {gf / t256:7.1f} TF/s | 128-tile 1-stage {t1s:7.3f} ms {gf / t1s:7.1f} TF/s'
        if '--wgrad' in sys.argv and C % 128 == 0 and K % 128 == 0:
            gy = torch.randn(N, K, Ho, Wo, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
            t3 = timeit(lambda: hip_conv.conv_wgrad(x, gy, K, R, R, st, pad, 1))
            t4 = timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [st, st], [pad, pad], [1, 1], False,
                                                                    [0, 0], 1, [False, True, False]))
            line += f'   wgrad {t3:6.3f} | torch {t4:6.3f}'
        print(line)


if __name__ == '__main__':
    main()
