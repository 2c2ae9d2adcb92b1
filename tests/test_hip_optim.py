"""-m gpu: apis.FusedSGD (csrc/optim.hip oadg_sgd_step_multi, one launch for all parameters) against torch.optim.SGD:
parameters and momentum buffers bit-identical over several steps (first step = buffer initialisation, changing learning
rate), for contiguous, channels_last, tiny and odd-sized tensors and for gradients that are unaligned slices of a flat
buffer (the data-parallel reducer's layout); state_dict interchangeable; unsupported settings take torch's path."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(64, 32, 3, 3), (256,), (1,), (1000, 77), (7, 5, 1, 1), (4099,), (128, 64, 1, 1), (1, 64, 3, 3)]


def _params(dev, seed=0):
    g = torch.Generator(device=dev).manual_seed(seed)
    ps = []
    for i, shp in enumerate(SHAPES):
        t = torch.randn(*shp, device=dev, generator=g)
        if len(shp) == 4 and i % 2 == 0:
            t = t.contiguous(memory_format=torch.channels_last)
        ps.append(torch.nn.Parameter(t))
    return ps


def _set_grads(ps, step, dev, flat_slices=False):
    g = torch.Generator(device=dev).manual_seed(100 + step)
    if flat_slices:      # gradients as slices of one flat buffer at odd element offsets (4-byte aligned only)
        flat = torch.randn(sum(p.numel() for p in ps) + len(ps) + 1, device=dev, generator=g)
        at = 1
        for p in ps:
            p.grad = flat[at:at + p.numel()].as_strided(p.shape, p.stride())
            at += p.numel() + 1
    else:
        for p in ps:
            if p.dim() != 4:
                p.grad = torch.randn(p.shape, device=dev, generator=g).as_strided(p.shape, p.stride())
            else:
                p.grad = torch.empty_like(p).copy_(torch.randn(p.shape, device=dev, generator=g))
                if p.shape[2:] == (1, 1):      # the same memory, other (arbitrary) strides on the size-1 dimensions
                    st = p.grad.stride()
                    p.grad = p.grad.as_strided(p.shape, (st[0], st[1], 1, 1))


@pytest.mark.parametrize('wd', [1e-4, 0.0])
@pytest.mark.parametrize('flat', [False, True])
def test_fused_sgd_is_bit_identical_to_torch_sgd(dev, wd, flat):
    from oadg_amd.apis import FusedSGD
    a, b = _params(dev), _params(dev)
    oa = torch.optim.SGD(a, lr=0.02, momentum=0.9, weight_decay=wd)
    ob = FusedSGD(b, lr=0.02, momentum=0.9, weight_decay=wd)
    for step in range(4):
        for o in (oa, ob):
            o.param_groups[0]['lr'] = 0.02 * (0.5 if step >= 2 else 1.0) * (step + 1) / 4
        _set_grads(a, step, dev, flat)
        _set_grads(b, step, dev, flat)
        v0 = [p._version for p in b]
        oa.step()
        ob.step()
        assert ob._tables, 'the fused path did not run'
        for x, y, v in zip(a, b, v0):
            assert torch.equal(x, y), (step, tuple(x.shape))
            assert torch.equal(oa.state[x]['momentum_buffer'], ob.state[y]['momentum_buffer']), (step, tuple(x.shape))
            assert y._version > v                   # consumers keyed on the version counter see the update
    # the state dicts are interchangeable
    sa, sb = oa.state_dict(), ob.state_dict()
    assert sa['param_groups'] == sb['param_groups']
    oc = FusedSGD(_params(dev), lr=0.1, momentum=0.9, weight_decay=wd)
    oc.load_state_dict(copy.deepcopy(sa))
    c = oc.param_groups[0]['params']
    with torch.no_grad():
        for x, y in zip(a, c):
            y.copy_(x)
    _set_grads(a, 9, dev, flat)
    _set_grads(c, 9, dev, flat)
    oa.step()
    oc.step()
    for x, y in zip(a, c):
        assert torch.equal(x, y)


def test_fused_sgd_leaves_unsupported_settings_to_torch(dev):
    from oadg_amd.apis import FusedSGD
    a, b = _params(dev), _params(dev)
    oa = torch.optim.SGD(a, lr=0.02, momentum=0.9, nesterov=True)
    ob = FusedSGD(b, lr=0.02, momentum=0.9, nesterov=True)
    for step in range(2):
        _set_grads(a, step, dev)
        _set_grads(b, step, dev)
        oa.step()
        ob.step()
    assert not ob._tables
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    # a parameter without a gradient is skipped, as torch does
    oc = FusedSGD(_params(dev), lr=0.02, momentum=0.9)
    ps = oc.param_groups[0]['params']
    _set_grads(ps, 0, dev)
    ps[1].grad = None
    before = ps[1].detach().clone()
    oc.step()
    assert torch.equal(ps[1], before) and 'momentum_buffer' not in oc.state[ps[1]]


@pytest.mark.parametrize('mode', ['absent', 'old_signature'])
def test_fused_sgd_version_bump_fallback(dev, monkeypatch, mode):
    """FusedSGD writes the parameters behind ATen's back and bumps their version counters through a private torch entry
    point (torch._C._autograd._unsafe_set_version_counter with two lists).  A torch without it, or with the older
    (Tensor, int) signature, takes the fallback - an in-place no-op per tensor - and everything keyed on the counters
    (the bank of prepared convolution weights) still sees the update; values stay bit-identical to torch.optim.SGD."""
    from oadg_amd.apis import FusedSGD
    if mode == 'absent':
        monkeypatch.delattr(torch._C._autograd, '_unsafe_set_version_counter', raising=False)
    else:
        def old(t, v):
            if not isinstance(t, torch.Tensor):
                raise TypeError('_unsafe_set_version_counter(): argument "t" must be Tensor, not list')
        monkeypatch.setattr(torch._C._autograd, '_unsafe_set_version_counter', old, raising=False)
    a, b = _params(dev), _params(dev)
    oa = torch.optim.SGD(a, lr=0.02, momentum=0.9, weight_decay=1e-4)
    ob = FusedSGD(b, lr=0.02, momentum=0.9, weight_decay=1e-4)
    for step in range(2):
        _set_grads(a, step, dev)
        _set_grads(b, step, dev)
        v0 = [p._version for p in b]
        oa.step()
        ob.step()
        assert ob._tables, 'the fused path did not run'
        for x, y, v in zip(a, b, v0):
            assert torch.equal(x, y) and y._version > v
