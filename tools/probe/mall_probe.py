"""Does a consumer launch find its producer's output in the 256 MB memory-side cache?  Times the layer3 conv1 launch
(1024 -> 256, 1x1, 65536 pixels: reads 134 MB) alone (operands cache-resident), after a 1 GiB fill (cold), and directly after
the launch that produces its input (conv3 of the previous block: 256 -> 1024 with / without the residual operand)   (GPU)"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oadg_amd  # noqa: E402,F401
from oadg_amd import hip_conv  # noqa: E402

dev = torch.device('cuda:0')
cl = dict(memory_format=torch.channels_last)
flush = torch.empty(1 << 30, dtype=torch.uint8, device=dev)


def run(N, H, W, Cs, Cb):
    x = torch.randn(N, Cs, H, W, device=dev).bfloat16().contiguous(**cl)
    wA = (torch.randn(Cb, Cs, 1, 1, device=dev) / Cs ** 0.5).bfloat16().contiguous(**cl)
    wB = (torch.randn(Cs, Cb, 1, 1, device=dev) / Cb ** 0.5).bfloat16().contiguous(**cl)
    bA, bB = torch.randn(Cb, device=dev), torch.randn(Cs, device=dev)
    res = torch.randn(N, Cb, H, W, device=dev).bfloat16().contiguous(**cl)
    bo = torch.empty(N * H * W * Cb // 8, dtype=torch.uint8, device=dev)
    y = [None]

    def A(full):
        y[0] = hip_conv.conv_forward(x, wA, bA, res if full else None, 1, 0, 1, True, bits_out=bo)[0] if False else \
            hip_conv.conv_forward(x, wA, bA, res if full else None, 1, 0, 1, True, bits_out=bo)

    def B():
        t = y[0][0] if isinstance(y[0], (tuple, list)) else y[0]
        return hip_conv.conv_forward(t, wB, bB, None, 1, 0, 1, True)

    def timed(pre, iters=10):
        tot = 0.0
        ev = []
        for i in range(iters):
            pre(i)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); B(); b.record()
            ev.append((a, b))
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in ev) / iters * 1e3

    A(True); B(); torch.cuda.synchronize()
    mb = N * H * W * Cb * 2 / 1e6
    print(f'{N}x{H}x{W}: consumer {Cb} -> {Cs} reads {mb:.0f} MB', flush=True)
    print(f'  hot (back to back)            {timed(lambda i: None):7.1f} us')
    print(f'  after a 1 GiB fill            {timed(lambda i: flush.fill_(i)):7.1f} us')
    print(f'  fill, producer (+ residual)   {timed(lambda i: (flush.fill_(i), A(True))):7.1f} us')
    print(f'  fill, producer (no residual)  {timed(lambda i: (flush.fill_(i), A(False))):7.1f} us')
    print(f'  fill, producer x2 (+ residual){timed(lambda i: (flush.fill_(i), A(True), A(True))):7.1f} us', flush=True)


run(8, 64, 128, 256, 1024)
run(8, 32, 64, 512, 2048)
run(8, 128, 256, 128, 512)
