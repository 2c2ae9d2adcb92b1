"""Which statement of PendingSampling.choose blocks the host while the stream is busy?"""
import time
import torch
dev = torch.device('cuda:0')
a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
mask = torch.rand(523776, device=dev) > 0.1
perm_cpu = torch.randperm(400000)[:256]
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
    print(f'{label:34s} {dt:9.1f} us')
    return r
for rep in range(2):
    pinned = t('pin_memory', lambda: perm_cpu.pin_memory())
    perm = t('pinned.to(dev, non_blocking)', lambda: pinned.to(dev, non_blocking=True))
    t('pin+to', lambda: perm_cpu.pin_memory().to(dev, non_blocking=True))
    flags = t('zeros', lambda: torch.zeros(mask.numel() + 1, dtype=torch.bool, device=dev))
    t('flags[perm]=True', lambda: flags.__setitem__(perm, True))
    t('index_fill_', lambda: flags.index_fill_(0, perm, True))
    rank = t('cumsum', lambda: torch.cumsum(mask, 0) - 1)
    sel = t('mask & flags[rank]', lambda: mask & flags[rank.clamp(min=0)])
    t('nonzero_static', lambda: torch.nonzero_static(sel, size=256).squeeze(1))
    ev = torch.cuda.Event()
    t('event.record', lambda: ev.record())
    buf = torch.empty(16, dtype=torch.long, pin_memory=True)
    src = torch.arange(16, device=dev)
    t('pinned.copy_(dev, non_blocking)', lambda: buf.copy_(src, non_blocking=True))
    print('---')
