import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oadg_amd
from oadg_amd import hip_conv
dev = torch.device('cuda:0')
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
for (N, C, H, W, K) in ((8, 64, 256, 512, 256), (8, 128, 128, 256, 512), (8, 256, 64, 128, 1024), (8, 512, 32, 64, 2048)):
    x = torch.randn(N, C, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(K, C, 1, 1, device=dev) / C ** 0.5).bfloat16().contiguous(memory_format=torch.channels_last)
    b = torch.randn(K, device=dev)
    res = torch.randn(N, K, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
    mask = torch.relu(torch.randn(N, K, H, W, device=dev)).bfloat16().contiguous(memory_format=torch.channels_last)
    byt = 2.0 * (N * H * W * (C + 2 * K))
    for v in (1, 2, 3):
        t_res = timeit(lambda: hip_conv.conv_forward(x, w, b, res, 1, 0, 1, True, variant=v))
        t_rm = timeit(lambda: hip_conv.conv_forward(x, w, None, res, 1, 0, 1, False, variant=v, mask=mask, want_colsum=True))
        t_plain = timeit(lambda: hip_conv.conv_forward(x, w, b, None, 1, 0, 1, True, variant=v))
        print(f'C{C} K{K} {H}x{W} variant {v}: plain {t_plain:.3f} ms  +res {t_res:.3f} ms ({byt / t_res / 1e9:.2f} TB/s)  +res+mask+colsum {t_rm:.3f} ms')
