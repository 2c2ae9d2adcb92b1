"""CPU: the O(k) replay of ATen's CPU randperm (csrc/host_rng.hip) returns the same prefix and leaves the
global generator in the same state as torch.randperm."""
import pytest
import torch

from oadg_amd.core.bbox import randperm_prefix


@pytest.mark.parametrize('n,k', [(523776, 256), (4096, 128), (5000, 5000), (6000, 7000), (100000, 1),
                                 (8191, 8190), (1000, 128), (1, 1)])
def test_randperm_prefix_matches_torch(n, k):
    torch.manual_seed(n + k)
    torch.rand(7)                       # move the engine off a block boundary
    ref = torch.randperm(n)[:k]
    after = torch.rand(5)
    torch.manual_seed(n + k)
    torch.rand(7)
    got = randperm_prefix(n, k)
    assert torch.equal(ref, got)
    assert torch.equal(after, torch.rand(5)), 'generator state diverged'


def test_consecutive_calls_cross_reload_boundaries():
    torch.manual_seed(3)
    ref = [torch.randperm(n)[:64] for n in (5000, 7001, 523776, 4999, 90000)]
    tail = torch.rand(3)
    torch.manual_seed(3)
    got = [randperm_prefix(n, 64) for n in (5000, 7001, 523776, 4999, 90000)]
    assert all(torch.equal(a, b) for a, b in zip(ref, got))
    assert torch.equal(tail, torch.rand(3))
