"""What the library GEMM (hipBLASLt through torch.mm, bf16, fp32 accumulate) reaches on the explicit-GEMM shapes of the
step's heaviest convolutions, on this box, with the clock / power it held: the practical MFMA ceiling of a power-capped
MI355X for these shapes (the 256-tile convolution kernel is an IMPLICIT GEMM of the same M x N x K)."""
import sys, time, threading
import torch
sys.path.insert(0, '.')
from bench import ClockSampler
dev = 'cuda'
def run(M, N, K, iters=30, zeros=False):
    a = torch.zeros(M, K, device=dev, dtype=torch.bfloat16) if zeros else torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    b = torch.zeros(K, N, device=dev, dtype=torch.bfloat16) if zeros else torch.randn(K, N, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        a @ b
    torch.cuda.synchronize()
    cs = ClockSampler(0, period=0.02).start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        a @ b
    e1.record()
    torch.cuda.synchronize()
    s = cs.stop()
    ms = e0.elapsed_time(e1) / iters
    print(f'M {M} N {N} K {K} {"zeros" if zeros else "randn"}: {ms * 1e3:8.1f} us  {2.0 * M * N * K / ms / 1e9:7.1f} TF/s  '
          f'sclk {s.get("sclk_mhz")} power {s.get("socket_power_w")}', flush=True)
for zeros in (False, True):
    run(8 * 256 * 512, 256, 2304, zeros=zeros)       # P2 3x3, 256 -> 256
    run(8 * 128 * 256, 256, 2304, zeros=zeros)       # P3 3x3
    run(8 * 64 * 128, 256, 2304, zeros=zeros)        # layer3 3x3
    run(8192, 8192, 8192, zeros=zeros)
    run(256, 256 * 9, 8 * 256 * 512, iters=10, zeros=zeros)    # the P2 weight gradient as a GEMM (K = pixels)
