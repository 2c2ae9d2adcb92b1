"""a co-tenant PROCESS: one of mfma_tenant.hip's micro-kernels in an endless loop (TENANT_LIB, argv[1] = mfma32 | mfma16 | valu)"""
import ctypes
import os
import sys

import torch

T = ctypes.CDLL(os.environ['TENANT_LIB'])
kind = dict(mfma32=0, mfma16=1, valu=2)[sys.argv[1]]
buf = torch.empty(1024 * 256, device='cuda:0')
st = torch.cuda.current_stream()
while True:
    for _ in range(20):
        T.launch(kind, ctypes.c_void_p(buf.data_ptr()), 1024, 20000, ctypes.c_void_p(st.cuda_stream))
    st.synchronize()
