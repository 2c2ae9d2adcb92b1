"""CPU: the C-ABI library loads and exports every symbol include/oadg_hip.h declares (no compute)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'oadg_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(oadg_[a-z0-9_]+)\s*\(', src)))


def test_header_declares_something():
    assert len(_declared()) >= 10


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    from oadg_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        g.build()
    h = _lib.lib()
    for name in _declared():
        assert hasattr(h, name), name
        assert name in _lib.SIGNATURES, f'{name} has no ctypes signature'
    assert sorted(_lib.SIGNATURES) == _declared()


def test_hot_ops_refuse_cpu_tensors():
    import torch
    from oadg_amd import hip_ops
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        hip_ops.supcon_loss(torch.zeros(4, 64), torch.zeros(4, 1, dtype=torch.long), 2, 0)
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        hip_ops.roi_align_fpn([torch.zeros(1, 4, 8, 8)], torch.zeros(1, 5), 7, [0.25])


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'oa-dg_amd')
    for dp, _, fn in os.walk(pkg):
        for f in fn:
            if f.endswith('.py'):
                s = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', s, flags=re.M), os.path.join(dp, f)
