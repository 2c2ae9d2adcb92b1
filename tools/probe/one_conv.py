import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oadg_amd
from oadg_amd import hip_conv
dev = torch.device('cuda:0')
N, C, H, W, K = 8, 256, 256, 512, 256
x = torch.randn(N, C, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
w = (torch.randn(K, C, 3, 3, device=dev) / 48).bfloat16().contiguous(memory_format=torch.channels_last)
gy = torch.randn(N, K, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
for _ in range(3):
    hip_conv.conv_forward(x, w, None, None, 1, 1, 1, False, variant=2)
    hip_conv.conv_forward(x, w, None, None, 1, 1, 1, False, variant=1)
    hip_conv.conv_wgrad(x, gy, K, 3, 3, 1, 1, 1)
torch.cuda.synchronize()
