"""Registries and ``build_from_cfg``: the reference's plug-in surface.

The reference resolves every component by ``dict(type='Name', ...)`` through mmcv registries declared at
mmdet/models/builder.py:7-15 (MODELS and its aliases), mmdet/datasets/builder.py:27-28 (PIPELINES,
DATASETS), mmdet/core/bbox/builder.py:4-6 (assigners / samplers / coders) and
mmdet/core/anchor/builder.py (PRIOR_GENERATORS).  The type strings registered here are the reference's.
"""
import inspect


class Registry:

    def __init__(self, name, parent=None):
        self.name = name
        self.parent = parent
        self._modules = {}

    def __repr__(self):
        return f'Registry({self.name}, {sorted(self._modules)})'

    def __contains__(self, key):
        return self.get(key) is not None

    def __len__(self):
        return len(self._modules)

    @property
    def module_dict(self):
        return self._modules

    def get(self, key):
        if key in self._modules:
            return self._modules[key]
        return self.parent.get(key) if self.parent is not None else None

    def _add(self, obj, name, force):
        names = [name] if isinstance(name, str) else list(name or [obj.__name__])
        for n in names:
            if n in self._modules and not force and self._modules[n] is not obj:
                raise KeyError(f'{n} is already registered in {self.name}')
            self._modules[n] = obj

    def register_module(self, name=None, force=False, module=None):
        """``@R.register_module()``, ``@R.register_module('alias')`` or ``R.register_module(module=cls)``."""
        if module is not None:
            self._add(module, name, force)
            return module
        if inspect.isclass(name) or inspect.isfunction(name):
            self._add(name, None, force)
            return name

        def deco(obj):
            self._add(obj, name, force)
            return obj
        return deco

    def build(self, cfg, default_args=None):
        return build_from_cfg(cfg, self, default_args)


def build_from_cfg(cfg, registry, default_args=None):
    if not isinstance(cfg, dict) or 'type' not in cfg:
        raise TypeError(f'cfg must be a dict with a "type" key, got {cfg!r}')
    args = dict(cfg)
    for k, v in (default_args or {}).items():
        args.setdefault(k, v)
    t = args.pop('type')
    obj = registry.get(t) if isinstance(t, str) else t
    if obj is None:
        raise KeyError(f'{t} is not in the {registry.name} registry')
    try:
        return obj(**args)
    except Exception as e:  # the reference re-raises with the class name, keep that behaviour
        raise type(e)(f'{getattr(obj, "__name__", obj)}: {e}') from e


MODELS = Registry('models')
BACKBONES = NECKS = ROI_EXTRACTORS = SHARED_HEADS = HEADS = LOSSES = DETECTORS = MODELS
PIPELINES = Registry('pipeline')
DATASETS = Registry('dataset')
BBOX_ASSIGNERS = Registry('bbox_assigner')
BBOX_SAMPLERS = Registry('bbox_sampler')
BBOX_CODERS = Registry('bbox_coder')
PRIOR_GENERATORS = Registry('prior_generator')
ANCHOR_GENERATORS = PRIOR_GENERATORS
IOU_CALCULATORS = Registry('iou_calculator')
ROI_LAYERS = Registry('roi_layer')        # the reference looks these up as attributes of mmcv.ops
OPTIMIZERS = Registry('optimizer')


def build_backbone(cfg):
    return BACKBONES.build(cfg)


def build_neck(cfg):
    return NECKS.build(cfg)


def build_roi_extractor(cfg):
    return ROI_EXTRACTORS.build(cfg)


def build_head(cfg):
    return HEADS.build(cfg)


def build_loss(cfg):
    return LOSSES.build(cfg)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    """mmdet/models/builder.py build_detector: train_cfg/test_cfg travel inside the model dict."""
    assert cfg.get('train_cfg') is None or train_cfg is None
    assert cfg.get('test_cfg') is None or test_cfg is None
    return DETECTORS.build(cfg, default_args=dict(train_cfg=train_cfg, test_cfg=test_cfg))


def build_assigner(cfg, **default_args):
    return BBOX_ASSIGNERS.build(cfg, default_args)


def build_sampler(cfg, **default_args):
    return BBOX_SAMPLERS.build(cfg, default_args)


def build_bbox_coder(cfg, **default_args):
    return BBOX_CODERS.build(cfg, default_args)


def build_prior_generator(cfg, default_args=None):
    return PRIOR_GENERATORS.build(cfg, default_args)


build_anchor_generator = build_prior_generator
