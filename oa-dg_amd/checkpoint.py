"""Checkpoint loading with mmcv's semantics (mmcv/runner/checkpoint.py load_checkpoint / load_state_dict /
_load_checkpoint_with_prefix, called from mmdet/apis/train.py:190-198 for ``load_from`` / ``resume_from`` and from
``init_cfg=dict(type='Pretrained', checkpoint=...)`` of the backbone, configs/_base_/models/faster_rcnn_r50_fpn.py:15):

* accepts ``{'state_dict': ...}`` or a bare state dict, strips a ``module.`` prefix;
* parameters whose SHAPE differs from the model's (a COCO head with 81 classes loaded into an 8-class head) are
  skipped and reported instead of raising; missing and unexpected keys are reported;
* ``prefix``: take only the keys under that prefix (``backbone.`` out of a detector checkpoint);
* ``torchvision://`` / ``open-mmlab://`` / ``http(s)://`` names cannot be downloaded here (no network): they are looked
  up in ``$OADG_CHECKPOINT_DIR`` (default ``~/.cache/oadg/checkpoints``) by file name and otherwise raise
  ``FileNotFoundError`` - a configured checkpoint is never silently ignored.
"""
import os
import re

import torch


def checkpoint_dir():
    return os.environ.get('OADG_CHECKPOINT_DIR', os.path.join(os.path.expanduser('~'), '.cache', 'oadg', 'checkpoints'))


def resolve_checkpoint(name):
    """local path of a checkpoint name, or FileNotFoundError with what to put where"""
    name = str(name)
    if os.path.exists(name):
        return name
    m = re.match(r'^(torchvision|open-mmlab|openmmlab|mmcls|https?)://(.*)$', name)
    if m is None:
        raise FileNotFoundError(f'checkpoint {name!r} does not exist')
    stem = os.path.splitext(os.path.basename(m.group(2).rstrip('/')))[0]
    d = checkpoint_dir()
    if os.path.isdir(d):
        # the stem followed by '-', '_' + hash, '.' or the extension: 'resnet50' must not pick up 'resnet50_caffe-*.pth'
        pat = re.compile(r'^' + re.escape(stem) + r'([-.][^/]*)?\.(pth|pt)$')
        hits = sorted(f for f in os.listdir(d) if pat.match(f))
        if hits:
            return os.path.join(d, hits[0])
    raise FileNotFoundError(
        f'checkpoint {name!r} cannot be downloaded (no network); place the file as {os.path.join(d, stem)}*.pth '
        f'(OADG_CHECKPOINT_DIR overrides the directory) or set the entry to a local path / None')


def load_state_dict(module, state_dict, strict=False, logger=print, what='checkpoint'):
    """mmcv load_state_dict + the size-mismatch tolerance of its _load_from_state_dict wrapper; returns a report"""
    own = module.state_dict()
    use, mismatched = {}, []
    for k, v in state_dict.items():
        if k in own and tuple(own[k].shape) != tuple(v.shape):
            mismatched.append((k, tuple(v.shape), tuple(own[k].shape)))
            continue
        use[k] = v
    res = module.load_state_dict(use, strict=False)
    skipped = {m[0] for m in mismatched}
    missing = [k for k in res.missing_keys if 'num_batches_tracked' not in k and k not in skipped]
    unexpected = list(res.unexpected_keys)
    report = dict(missing=missing, unexpected=unexpected, mismatched=mismatched, loaded=len(use) - len(unexpected))
    msgs = []
    if unexpected:
        msgs.append(f'unexpected key in source state_dict: {", ".join(unexpected[:12])}' + (' ...' if len(unexpected) > 12 else ''))
    if missing:
        msgs.append(f'missing keys in source state_dict: {", ".join(missing[:12])}' + (' ...' if len(missing) > 12 else ''))
    for k, a, b in mismatched:
        msgs.append(f'size mismatch for {k}: copying a param with shape {a} from {what}, the shape in current model is {b}.')
    if msgs:
        text = f'The model and loaded state dict do not match exactly ({what})\n' + '\n'.join(msgs)
        if strict:
            raise RuntimeError(text)
        if logger is not None:
            logger(text)
    return report


def load_checkpoint(module, filename, map_location='cpu', strict=False, prefix=None, logger=print):
    path = resolve_checkpoint(filename)
    ck = torch.load(path, map_location=map_location, weights_only=False)
    sd = ck.get('state_dict', ck) if isinstance(ck, dict) else ck
    if not isinstance(sd, dict):
        raise RuntimeError(f'No state_dict found in checkpoint file {path}')
    sd = {re.sub(r'^module\.', '', k): v for k, v in sd.items()}
    if prefix:
        p = prefix if prefix.endswith('.') else prefix + '.'
        sub = {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
        if not sub:
            raise RuntimeError(f'{prefix} is not in the pretrained model {path}')
        sd = sub
    report = load_state_dict(module, sd, strict, logger, what=os.path.basename(path))
    report['path'] = path
    report['meta'] = ck.get('meta', {}) if isinstance(ck, dict) else {}
    report['checkpoint'] = ck
    return report
