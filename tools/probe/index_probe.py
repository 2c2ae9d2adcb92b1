"""Which index-assignment forms block the host while the stream is busy?"""
import time
import torch
dev = torch.device('cuda:0')
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
def busy():
    for _ in range(40):
        a @ a
def t(label, fn, n=5):
    torch.cuda.synchronize(); busy()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    dt = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    print(f'{label:44s} {dt:9.1f} us')
    return r
idx = torch.randint(0, 500000, (256,), device=dev)
fb = torch.zeros(500001, dtype=torch.bool, device=dev)
ff = torch.zeros(500001, device=dev)
fl = torch.zeros(500001, dtype=torch.long, device=dev)
f2 = torch.zeros(500001, 4, device=dev)
v2 = torch.randn(256, 4, device=dev)
vl = torch.randint(0, 8, (256,), device=dev)
one = torch.ones((), device=dev)
for rep in range(2):
    t('bool[idx] = True', lambda: fb.__setitem__(idx, True))
    t('float[idx] = 1.0', lambda: ff.__setitem__(idx, 1.0))
    t('long[idx] = 0', lambda: fl.__setitem__(idx, 0))
    t('long[idx] = tensor', lambda: fl.__setitem__(idx, vl))
    t('f2[idx, :] = 1.0', lambda: f2.__setitem__((idx, slice(None)), 1.0))
    t('f2[idx, :] = tensor', lambda: f2.__setitem__((idx, slice(None)), v2))
    t('f2[idx] = tensor', lambda: f2.__setitem__(idx, v2))
    t('float[idx] = 0-d dev tensor', lambda: ff.__setitem__(idx, one))
    t('index_fill_', lambda: ff.index_fill_(0, idx, 1.0))
    t('index_copy_', lambda: f2.index_copy_(0, idx, v2))
    t('index_put_((idx,), v2)', lambda: f2.index_put_((idx,), v2))
    t('ff[idx] (gather)', lambda: ff[idx])
    t('f2[idx] (gather)', lambda: f2[idx])
    t('f2[idx, :2]', lambda: f2[idx, :2])
    t('torch.where(mask, a, b)', lambda: torch.where(fb, ff, ff))
    t('new_full', lambda: ff.new_full((1000,), 3.0))
    t('new_tensor', lambda: ff.new_tensor([1.0, 2.0]))
    t('tensor * python float', lambda: ff * 2.5)
    t('clamp(min=0)', lambda: ff.clamp(min=0))
    t('tensor.sum()', lambda: ff.sum())
    t('bool.sum()', lambda: fb.sum())
    t('torch.stack scalars', lambda: torch.stack([ff.sum(), ff.sum()]))
    t('nonzero_static', lambda: torch.nonzero_static(fb, size=256))
    t('sort', lambda: ff[:2000].sort(descending=True))
    t('topk', lambda: ff.topk(2000))
    t('cat', lambda: torch.cat([ff, ff]))
    t('F.pad', lambda: torch.nn.functional.pad(f2, (0, 1)))
    t('max dim', lambda: f2.max(dim=1))
    t('unique (sync expected)', lambda: idx.unique())
    print('---')
