"""GB/s of the three 16-channel head kernels (csrc/narrow_head.hip) on the pyramid levels of the benchmark step   (GPU)"""
import os
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oadg_amd  # noqa: F401,E402
from oadg_amd import _lib  # noqa: E402
from oadg_amd.hip_conv import ptr, stream_ptr, check, _zeros  # noqa: E402

L = _lib.lib()
dev = torch.device('cuda:0')
C = 256
for name, M in (('P2', 8 * 256 * 512), ('P3', 8 * 128 * 256), ('P4', 8 * 64 * 128), ('P5', 8 * 32 * 64)):
    x = torch.randn(M, C, device=dev).to(torch.bfloat16)
    w = (torch.randn(16, C, device=dev) * 0.05).to(torch.bfloat16)
    wt = w.t().contiguous()
    b = torch.randn(16, device=dev)
    y = torch.empty(M, 16, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(M, 16, device=dev).to(torch.bfloat16)
    dx = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
    bits = torch.randint(0, 256, (M, C // 8), device=dev, dtype=torch.uint8)
    part = torch.empty(L.oadg_conv1x1_n16_dgrad_rows(M), C, device=dev)
    wp = torch.empty(L.oadg_conv1x1_n16_wgrad_rows(M), 16, C, device=dev)
    bp = torch.empty(L.oadg_conv1x1_n16_wgrad_rows(M), 16, device=dev)
    z = _zeros(dev)
    runs = dict(
        fwd=(lambda: L.oadg_conv1x1_n16_fwd(ptr(x), ptr(w), ptr(b), ptr(y), M, C, stream_ptr()), M * (2 * C + 32)),
        dgrad=(lambda: L.oadg_conv1x1_n16_dgrad(ptr(dy), ptr(wt), ptr(dx), ptr(bits), ptr(part), M, C, stream_ptr()),
               M * (2 * C + 32 + C // 8)),
        wgrad=(lambda: L.oadg_conv1x1_n16_wgrad(ptr(x), ptr(dy), ptr(wp), ptr(bp), ptr(z), M, C, stream_ptr()),
               M * (2 * C + 32)))
    out = []
    for k, (fn, nbytes) in runs.items():
        for _ in range(3):
            check(fn(), k)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        out.append(f'{k} {us:7.1f} us {nbytes / us / 1e6:5.2f} TB/s')
    print(name, M, ' | '.join(out), flush=True)
