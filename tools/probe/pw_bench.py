"""pointwise (1x1 / stride 1) convolution launches of the step: the streaming kernel (variant 4) against the 128-tile
kernel (variant 3) and the 256-tile kernel (variant 2) on the operand combinations their callers use, interleaved   (GPU)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oadg_amd  # noqa: E402,F401
from oadg_amd import hip_conv, _lib  # noqa: E402

SHAPES = [(8, 512, 128, 256, 128), (8, 512, 128, 256, 256), (8, 512, 32, 64, 2048), (8, 512, 64, 128, 1024),
          (8, 128, 128, 256, 512), (8, 256, 64, 128, 1024), (8, 64, 256, 512, 256), (8, 1024, 64, 128, 256)]
dev = torch.device('cuda:0')
L = _lib.lib()
cl = dict(memory_format=torch.channels_last)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


COLD = '--cold' in sys.argv        # every launch after a 1 GiB fill: operands come from HBM, as inside the training step
if COLD:
    _flush = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    _warm = timeit

    def timeit(fn, iters=10):       # noqa: F811
        _warm(fn, 1)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for i, (a, b) in enumerate(ev):
            _flush.fill_(i)
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in ev) / iters * 1e3


for N, C, H, W, K in SHAPES:
    x = torch.randn(N, C, H, W, device=dev).bfloat16().contiguous(**cl)
    w = (torch.randn(K, C, 1, 1, device=dev) / C ** 0.5).bfloat16().contiguous(**cl)
    b = torch.randn(K, device=dev)
    res = torch.randn(N, K, H, W, device=dev).bfloat16().contiguous(**cl)
    bits = (torch.rand(N * H * W * K // 8, device=dev) * 256).to(torch.uint8)
    bo = torch.empty_like(bits)
    M = N * H * W
    combos = [('plain', dict(bias=b), 2.0 * M * (C + K)),
              ('relu+bits_out', dict(bias=b, relu=True, bits_out=bo), 2.0 * M * (C + K) + M * K / 8),
              ('res+relu+bits_out', dict(bias=b, res=res, relu=True, bits_out=bo), 2.0 * M * (C + 2 * K) + M * K / 8),
              ('bits_in+colsum', dict(mask_bits=bits, want_colsum=True), 2.0 * M * (C + K) + M * K / 8),
              ('res+bits_in+colsum', dict(res=res, mask_bits=bits, want_colsum=True), 2.0 * M * (C + 2 * K) + M * K / 8)]
    auto = L.oadg_conv2d_auto_variant(N, H, W, C, K, 1, 1, 1, 0, 1)
    for name, kw, by in combos:
        def run(v):
            k = dict(kw)
            return hip_conv.conv_forward(x, w, k.pop('bias', None), k.pop('res', None), 1, 0, 1, k.pop('relu', False), variant=v, **k)
        ts = {}
        for v in (3, 4, 2):
            try:
                ts[v] = timeit(lambda: run(v))
            except Exception:
                ts[v] = float('nan')
        print(f'C{C:5d} K{K:5d} {H}x{W} {name:20s} auto={auto}  tile128 {ts[3]:7.1f} us {by / ts[3] / 1e6:5.2f} TB/s | stream {ts[4]:7.1f} us '
              f'{by / ts[4] / 1e6:5.2f} TB/s | tile256 {ts[2]:7.1f} us', flush=True)
