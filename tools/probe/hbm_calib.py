"""What simple streaming kernels reach on this box (cold operands: every timed launch touches tensors that were not used for
> 1 GB of other traffic): the practical HBM ceiling for the read/write mixes of the step's HBM-bound convolution launches."""
import torch
dev = 'cuda'
def timeit(fn, bytes_, name, reps=10):
    big = [torch.empty(64 << 20, dtype=torch.float32, device=dev) for _ in range(6)]      # 6 x 256 MB: flushes the 256 MB MALL
    ts = []
    for i in range(reps):
        big[i % 6].fill_(float(i))
        big[(i + 1) % 6].fill_(float(i))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[len(ts) // 2]
    print(f'{name:45s} {bytes_ / 1e6:8.1f} MB  {ms * 1e3:8.1f} us  {bytes_ / ms / 1e9:6.2f} TB/s', flush=True)
for mb in (33, 134, 537):
    n = mb * 1000 * 1000 // 2
    a = torch.randn(n, device=dev).bfloat16(); b = torch.randn(n, device=dev).bfloat16(); c = torch.empty_like(a)
    timeit(lambda: c.copy_(a), 2 * n * 2, f'copy bf16 {mb} MB (1R 1W)')
    timeit(lambda: torch.add(a, b, out=c), 3 * n * 2, f'add bf16 {mb} MB (2R 1W)')
    timeit(lambda: a.sum(), n * 2, f'sum bf16 {mb} MB (1R)')
    timeit(lambda: c.fill_(1.0), n * 2, f'fill bf16 {mb} MB (1W)')
    q = n // 4
    timeit(lambda: torch.add(a[:q], b[:q], out=c[:q]), (2 * q + q) * 2, f'add bf16 quarter of it')
