"""The victim of the packed-fp32 hazard (profiles/r06_packed_fp32_hazard.txt): RoIAlign's backward (the library named by
OADG_HIP_LIB, default = the shipped one) on fixed synthetic operands, every result compared bitwise with the first, while a
co-tenant keeps another stream of THIS process busy:  TENANT = none | mfma32 | mfma16 | valu  (micro-kernels of
mfma_tenant.hip: nothing but v_mfma_f32_32x32x16_bf16 / v_mfma_f32_16x16x32_bf16 / scalar-style fp32 FMAs in a loop)
| conv128 (the 128-tile convolution of the library).  Prints one line.  tools/probe, not the product."""
import ctypes
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import oadg_amd  # noqa: E402,F401
from oadg_amd import hip_conv, hip_ops  # noqa: E402

dev = torch.device('cuda:0')
N, C, H, W, K = 4, 256, 384, 768, 2048
g = torch.Generator().manual_seed(1)
feats = [torch.zeros(N, C, H // s, W // s, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
         .requires_grad_() for s in (4, 8, 16, 32)]
# proposals clustered around a few objects, most of them small (FPN level 0), as a detector's are
ctr = torch.rand(24, 2, generator=g) * torch.tensor([W, H])
pick = torch.randint(0, 24, (K,), generator=g)
cx, cy = (ctr[pick] + torch.randn(K, 2, generator=g) * 12).unbind(1)
w = torch.rand(K, generator=g) ** 2 * 160 + 12
h = torch.rand(K, generator=g) ** 2 * 120 + 12
rois = torch.stack([torch.randint(0, N, (K,), generator=g).float(), (cx - w / 2).clamp(0, W - 1), (cy - h / 2).clamp(0, H - 1),
                    (cx + w / 2).clamp(1, W), (cy + h / 2).clamp(1, H)], 1).to(dev)
gout = (torch.randn(K, C, 7, 7, generator=g) * 1e-5).to(dev).bfloat16().contiguous(memory_format=torch.channels_last)
scales = (0.25, 0.125, 0.0625, 0.03125)


def victim():
    return torch.cat([x.flatten() for x in torch.autograd.grad(hip_ops.roi_align_fpn(feats, rois, 7, scales), feats, gout)])


ref = victim().clone()
torch.cuda.synchronize()
tenant = os.environ.get('TENANT', 'none')
stop = False


def tenant_loop():
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        if tenant == 'conv128':
            cl = torch.channels_last
            x3 = torch.randn(8, 256, 32, 64, device=dev).bfloat16().contiguous(memory_format=cl)
            w3 = (torch.randn(256, 256, 3, 3, device=dev) * 0.02).bfloat16().contiguous(memory_format=cl)
            b3 = torch.randn(256, device=dev)
            run = lambda: hip_conv.conv_forward(x3, w3, b3, None, 1, 1, 1, True, variant=3)  # noqa: E731
        else:
            T = ctypes.CDLL(os.environ['TENANT_LIB'])
            buf = torch.empty(1024 * 256, device=dev)
            kind = dict(mfma32=0, mfma16=1, valu=2)[tenant]
            run = lambda: T.launch(kind, ctypes.c_void_p(buf.data_ptr()), 1024, 20000, ctypes.c_void_p(st.cuda_stream))  # noqa: E731
        while not stop:
            for _ in range(20):
                run()
            st.synchronize()


if tenant != 'none':
    threading.Thread(target=tenant_loop, daemon=True).start()
    time.sleep(2)
bad = n = worst = 0
lanes = set()
t0 = time.time()
while time.time() - t0 < float(os.environ.get('SECONDS', '12')):
    r = victim()
    if not torch.equal(r, ref):
        bad += 1
        if bad <= 50:
            d = (r[:feats[0].numel()].view(N, C, H // 4, W // 4) != ref[:feats[0].numel()].view(N, C, H // 4, W // 4))
            lanes |= set((d.any(3).any(2).any(0).nonzero().flatten() % 64).tolist())
            worst = max(worst, int(d.any(1).sum()))
    n += 1
stop = True
print(f'library {os.path.basename(os.environ.get("OADG_HIP_LIB", "shipped")):>22} | co-tenant {tenant:>8} | launches {n:6d} | wrong {bad:6d} | '
      f'most pixels of level 0 wrong in a launch {worst:4d} | channel % 64 of the wrong elements: {(str(min(lanes)) + "-" + str(max(lanes))) if lanes else "-"}', flush=True)
