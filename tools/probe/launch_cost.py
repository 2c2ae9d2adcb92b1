"""host cost of one convolution launch, layer by layer of the call stack (python wrapper / ctypes / HIP runtime)   (GPU)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oadg_amd import _lib, hip_conv
from oadg_amd._lib import ptr, stream_ptr

dev = torch.device('cuda:0')
hip_conv.enable()
L = _lib.lib()
N, C, H, W, K = 1, 64, 16, 16, 64
x = torch.randn(N, C, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
w = torch.randn(K, C, 1, 1, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
y = torch.empty(N, K, H, W, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
z = hip_conv._zeros(dev)
R = 2000


def timeit(name, fn):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(R):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'{name:58s} host {1e6 * (t1 - t0) / R:7.2f} us per call   (drained: {1e6 * (t2 - t0) / R:7.2f})')


args = (ptr(x), ptr(w), None, None, ptr(y), ptr(z), N, H, W, C, K, 1, 1, 1, 0, 1, 0, 3, None, None, None, None, stream_ptr())
timeit('raw ctypes call, arguments prepared', lambda: L.oadg_conv2d_nhwc_bf16_ex(*args))
timeit('ctypes call + ptr() / stream_ptr() per call', lambda: L.oadg_conv2d_nhwc_bf16_ex(
    ptr(x), ptr(w), None, None, ptr(y), ptr(z), N, H, W, C, K, 1, 1, 1, 0, 1, 0, 3, None, None, None, None, stream_ptr()))
timeit('hip_conv.conv_forward (allocates y)', lambda: hip_conv.conv_forward(x, w, None, None, 1, 0, 1, False, variant=3))
timeit('torch.empty of the output', lambda: torch.empty((N, K, H, W), dtype=torch.bfloat16, device=dev, memory_format=torch.channels_last))
timeit('torch.relu_ (an ATen elementwise launch)', lambda: torch.relu_(y))
xx = x.clone().requires_grad_(True)
conv = torch.nn.Conv2d(C, K, 1, bias=False).to(dev).to(memory_format=torch.channels_last)
from oadg_amd import layers
with torch.autocast('cuda', dtype=torch.bfloat16):
    timeit('layers.conv2d under autocast (autograd Function.apply)', lambda: layers.conv2d(xx, conv.weight, None, 1, 0, 1))
e = torch.cuda.Event(enable_timing=True)
timeit('torch.cuda.Event.record', lambda: e.record())
