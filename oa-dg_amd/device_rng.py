"""ATen's CPU MT19937 engine handed to the device for the RoI sampler (csrc/roi_sampler.hip) and back.

The reference draws every ``torch.randperm`` of its samplers from the global CPU generator
(mmdet/core/bbox/samplers/random_sampler.py:58).  The RoI head's draws depend on device data (candidate counts after the
NMS); rounds 1-4 read those counts to the host and drew there - the one host wait of a step behind which the device idles.
Now the engine state (624 words + ``left`` + ``next``) travels instead: uploaded right before the sampler launch (an
asynchronous 2.5 KB copy), advanced by the kernel exactly as ATen would, copied back behind it; the host generator is
set to the advanced state when the trainer next needs it (``sync_host``, at the end of the step - by then the event has
long fired).  Seeded runs stay draw-for-draw equal to the reference.
"""
import numpy as np
import torch

_WORDS = 626                      # state[624], left, next
_STATE_BYTES = 5056               # torch.get_rng_state() of the CPU generator (legacy pod layout)


def _pack(state_u8):
    """torch.get_rng_state() bytes -> int32[626] (state words, left, next)"""
    a = state_u8.numpy()
    assert a.size == _STATE_BYTES, 'unexpected CPU generator state layout'
    out = np.empty(_WORDS, np.uint32)
    out[:624] = a[24:24 + 624 * 8].view(np.uint64).astype(np.uint32)
    out[624] = a[8:12].view(np.int32)[0]
    out[625] = a[16:24].view(np.uint64)[0]
    return out


def _unpack_into(state_u8, words):
    a = state_u8.numpy()
    a[24:24 + 624 * 8] = words[:624].astype(np.uint64).view(np.uint8)
    a[8:12] = np.array([int(words[624])], np.int32).view(np.uint8)
    a[16:24] = np.array([int(words[625])], np.uint64).view(np.uint8)


class DeviceGenerator:
    """one per device; the host generator is authoritative between ``sync_host()`` and the next ``upload()``"""

    def __init__(self, device):
        self.device = device
        # [0] = the state handed to the kernel (read-only for its whole launch), [1] = the state it leaves
        self._both = torch.empty((2, _WORDS), dtype=torch.int32, device=device)
        self.state, self.state_out = self._both[0], self._both[1]
        self._up = torch.empty(_WORDS, dtype=torch.int32).pin_memory()
        self._down = torch.empty(_WORDS, dtype=torch.int32).pin_memory()
        self._event = None            # fires when the advanced state has landed in ``_down``
        self._base = None             # the host state the pending download was derived from

    def upload(self):
        """host generator -> device (asynchronous); call right before the kernel that draws"""
        assert self._event is None, 'DeviceGenerator.upload() with a download pending: call sync_host() first'
        st = torch.get_rng_state()
        self._up.numpy()[:] = _pack(st).view(np.int32)
        self.state.copy_(self._up, non_blocking=True)
        self._base = st
        return self.state, self.state_out

    def download_async(self):
        """device -> pinned host buffer behind the kernels enqueued so far"""
        self._down.copy_(self.state_out, non_blocking=True)
        self._event = torch.cuda.Event()
        self._event.record()

    def pending(self):
        return self._event is not None

    def sync_host(self):
        """set the host generator to the state the device left (waits for the download: a no-op wait in the trainer, which
        calls this at the end of the step)"""
        if self._event is None:
            return False
        self._event.synchronize()
        self._event = None
        st = self._base
        _unpack_into(st, self._down.numpy().view(np.uint32))
        torch.set_rng_state(st)
        self._base = None
        return True

    def discard(self):
        """drop a pending download (the step is being repeated from the saved host state)"""
        if self._event is not None:
            self._event.synchronize()
        self._event = self._base = None


_GENERATORS = {}


def generator(device):
    g = _GENERATORS.get(device)
    if g is None:
        g = _GENERATORS[device] = DeviceGenerator(device)
    return g


def sync_all():
    """bring the host generator up to date with every device generator (checkpointing, tests, anything that is about to
    draw from torch's CPU generator outside the trainer)"""
    return any([g.sync_host() for g in _GENERATORS.values()])
