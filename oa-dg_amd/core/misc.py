"""List/level plumbing (mmdet/core/utils/misc.py:11-60,122-155)."""
from functools import partial

import torch


def multi_apply(func, *args, **kwargs):
    """Map func over zipped args; return a tuple of per-output lists (misc.py:11-30)."""
    f = partial(func, **kwargs) if kwargs else func
    results = list(map(f, *args))
    return tuple(map(list, zip(*results)))


def unmap(data, count, inds, fill=0):
    """Scatter a subset back into a tensor of `count` rows (misc.py:33-45); inds is a bool mask."""
    if data.dim() == 1:
        ret = data.new_full((count,), fill)
        ret[inds.type(torch.bool)] = data
    else:
        ret = data.new_full((count,) + data.size()[1:], fill)
        ret[inds.type(torch.bool), :] = data
    return ret


def select_single_mlvl(mlvl_tensors, batch_id, detach=True):
    """Per-level [N,...] tensors -> one image's per-level list (misc.py:122-155)."""
    if detach:
        return [t[batch_id].detach() for t in mlvl_tensors]
    return [t[batch_id] for t in mlvl_tensors]
