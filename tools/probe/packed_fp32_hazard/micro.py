"""micro_victims.hip beside the co-tenant micro-kernels (mfma_tenant.hip) on another stream: which INSTRUCTION FORMS return
different bits when matrix instructions share the SIMD?  VICTIM_LIB / TENANT_LIB = the two built libraries."""
import ctypes
import os
import threading
import time

import torch

V = ctypes.CDLL(os.environ['VICTIM_LIB'])
V9 = ctypes.CDLL(os.environ['VICTIM9_LIB'])
T = ctypes.CDLL(os.environ['TENANT_LIB'])
dev = torch.device('cuda:0')
blocks = 2048
x = torch.rand(blocks * 256, device=dev) + 0.5
names = ['scalar v_fma_f32 / v_mul_f32', 'v_pk_fma_f32 / v_pk_mul_f32, plain operands', 'v_pk_mul_f32 op_sel_hi:[1,0] / v_pk_add_f32',
         'v_mov_b32 then v_pk_mul_f32 on its pair', 'v_mov_b32, s_nop 0, v_pk_mul_f32 on its pair',
         'ds_read_b64, waitcnt, v_pk_mul_f32', 'ds_read_b64, waitcnt, scalar v_mul/v_add', 'global_load_dwordx2, waitcnt, v_pk_mul_f32',
         'global_load_dwordx2, waitcnt, scalar v_mul/v_add',
         'C++ loop: LDS float4 weights in flight + packed mul/add']
stop = False


def tenant(kind):
    st = torch.cuda.Stream()
    buf = torch.empty(1024 * 256, device=dev)
    with torch.cuda.stream(st):
        while not stop:
            for _ in range(20):
                T.launch(kind, ctypes.c_void_p(buf.data_ptr()), 1024, 20000, ctypes.c_void_p(st.cuda_stream))
            st.synchronize()


def run(kind):
    out = torch.empty(2 * blocks * 256, device=dev)
    if kind == 9:
        V9.launch_victim9(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.data_ptr()), blocks, 1500,
                          ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        return out
    V.launch_victim(kind, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.data_ptr()), blocks, 4000 if kind < 5 else 1500,
                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    return out


refs = [run(k).clone() for k in range(10)]
torch.cuda.synchronize()
for tname, tk in (('none', None), ('valu', 2), ('mfma16', 1), ('mfma32', 0)):
    stop = False
    th = None
    if tk is not None:
        th = threading.Thread(target=tenant, args=(tk,), daemon=True)
        th.start()
        time.sleep(1.5)
    for k in range(10):
        bad = n = 0
        lanes = set()
        t0 = time.time()
        while time.time() - t0 < 5:
            r = run(k)
            if not torch.equal(r, refs[k]):
                bad += 1
                if bad <= 20:
                    lanes |= set((((r != refs[k]).view(-1, 2).any(1).nonzero().flatten()) % 64).tolist())
            n += 1
        print(f'co-tenant {tname:>7} | victim {names[k]:<46} | launches {n:5d} | differing {bad:5d} | lanes {(str(min(lanes)) + "-" + str(max(lanes))) if lanes else "-"}', flush=True)
    stop = True
    if th is not None:
        th.join()
