import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oadg_amd
from oadg_amd import hip_conv
dev = torch.device('cuda:0')
N, C, H, W, K, R = 8, 256, 128, 256, 256, 3
x = torch.randn(N, C, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
gy = torch.randn(N, K, H, W, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
w = torch.randn(K, C, R, R, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
for _ in range(3):
    hip_conv.conv_wgrad(x, gy, K, R, R, 1, 1, 1)
    hip_conv.conv_forward(x, w, None, None, 1, 1, 1, False)
torch.cuda.synchronize()
