"""The 128-tile kernel's variants on the small-map / K = 128 launches of R50-FPN and R101-DC5, COLD (768 MB of other traffic
between timed launches): variant 1 = two LDS stages (round 6: fragment reads software-pipelined), 3 = one stage, 2 = 256-tile.
usage: python tools/probe/tile128_probe.py   (GPU)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oadg_amd  # noqa: F401,E402
from oadg_amd import hip_conv  # noqa: E402

SHAPES = [  # name, N, C, H, W, K, R, stride, pad, dil
    ('layer4 3x3', 8, 512, 32, 64, 512, 3, 1, 1, 1),
    ('layer4 3x3 s2', 8, 512, 64, 128, 512, 3, 2, 1, 1),
    ('P5 3x3', 8, 256, 32, 64, 256, 3, 1, 1, 1),
    ('layer4 1x1 2048->512', 8, 2048, 32, 64, 512, 1, 1, 0, 1),
    ('layer2 3x3', 8, 128, 128, 256, 128, 3, 1, 1, 1),
    ('dc5 1x1 1024->256', 4, 1024, 46, 80, 256, 1, 1, 0, 1),
    ('dc5 3x3 256', 4, 256, 46, 80, 256, 3, 1, 1, 1),
    ('dc5 1x1 2048->512', 4, 2048, 46, 80, 512, 1, 1, 0, 1),
    ('dc5 3x3 512 dil2', 4, 512, 46, 80, 512, 3, 1, 2, 2),
]
dev = torch.device('cuda:0')
flush = torch.empty(768 << 20, dtype=torch.uint8, device=dev)


def cold_time(fn, iters=8):
    fn()
    ts = []
    for _ in range(iters):
        flush.add_(1)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2] * 1e3


for name, N, C, H, W, K, R, st, pad, dil in SHAPES:
    x = torch.randn(N, C, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(K, C, R, R, device=dev) / (C * R * R) ** 0.5).bfloat16().contiguous(memory_format=torch.channels_last)
    b = torch.randn(K, device=dev)
    Ho, Wo = (H + 2 * pad - dil * (R - 1) - 1) // st + 1, (W + 2 * pad - dil * (R - 1) - 1) // st + 1
    gf = 2.0 * N * Ho * Wo * K * C * R * R / 1e9
    ref = hip_conv.conv_forward(x, w, b, None, st, pad, dil, True, variant=3)
    line = f'{name:24s} {gf:7.1f} GF'
    for v in (1, 3, 2):
        if v == 2 and K % 256:
            continue
        y = hip_conv.conv_forward(x, w, b, None, st, pad, dil, True, variant=v)
        t = cold_time(lambda: hip_conv.conv_forward(x, w, b, None, st, pad, dil, False, variant=v))
        line += f' | v{v} {t:6.1f} us {gf / t * 1e3:6.0f} TF/s{"" if torch.equal(y, ref) else " DIFFERENT"}'
    print(line, flush=True)
