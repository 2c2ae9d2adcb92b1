"""host time per GEMM call of the RoI head's linears (the host-bound stretch of the step): F.linear under autocast / on
pre-cast bf16 operands / torch.mm / addmm, hipBLASLt vs rocBLAS backend; and the backward pair.   (GPU)"""
import time
import torch
import torch.nn.functional as F
dev = torch.device('cuda:0')
K = 4198
shapes = [(12544, 1024), (1024, 1024), (1024, 9), (1024, 36), (1024, 256)]


def host_us(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    return t


for lib in ('default', 'cublas', 'cublaslt'):
    if lib != 'default':
        torch.backends.cuda.preferred_blas_library(lib)
    for cin, cout in shapes:
        x = torch.randn(K, cin, device=dev).bfloat16()
        w32 = torch.randn(cout, cin, device=dev)
        b32 = torch.randn(cout, device=dev)
        w, b = w32.bfloat16(), b32.bfloat16()
        wt = w.t().contiguous()
        r = {}
        r['linear_bf16'] = host_us(lambda: F.linear(x, w, b))
        r['addmm_bf16'] = host_us(lambda: torch.addmm(b, x, w.t()))
        r['mm_bf16'] = host_us(lambda: torch.mm(x, wt))
        with torch.autocast('cuda', dtype=torch.bfloat16):
            r['linear_autocast_fp32w'] = host_us(lambda: F.linear(x, w32, b32))
        xg = x.clone().requires_grad_(True)
        wg = w.clone().requires_grad_(True)
        def fb():
            y = F.linear(xg, wg, b)
            y.backward(y)
            xg.grad = None; wg.grad = None
        r['linear_fwd+bwd'] = host_us(fb, 100)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            F.linear(x, w, b)
        e1.record(); torch.cuda.synchronize()
        print(lib, (cin, cout), {k: round(v, 1) for k, v in r.items()}, 'gpu_us', round(e0.elapsed_time(e1) * 50, 1), flush=True)
