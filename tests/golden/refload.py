"""Load the reference's Python (read-only, /root/reference) by path, in THIS container only.

Used by the ``make_*.py`` fixture generators next to this file; never imported by the test-suite proper,
by ``bench.py`` or by the product, and nothing from /root/reference is copied: the generators run the
reference's own functions on seeded inputs and commit inputs + outputs as small ``.npz`` fixtures.

The reference needs ``mmcv`` and ``cv2``, which are absent here.  This module installs
  * package shells for every ``mmdet`` sub-package (``__path__`` only, their ``__init__`` is NOT run) whose
    attributes resolve lazily to whichever reference file defines the requested name,
  * a minimal ``mmcv`` stand-in (Registry, build_from_cfg, jit, BaseModule, ConvModule, init helpers),
  * ``mmcv.ops.{RoIAlign,batched_nms,nms}`` and a ``cv2`` module supplied by the caller (normally the
    CPU oracle's restatement of those un-vendored dependencies).
"""
import ast
import importlib
import inspect
import os
import sys
import types

import torch
import torch.nn as nn

REF = os.environ.get('OADG_REFERENCE', '/root/reference')


def available():
    return os.path.isdir(os.path.join(REF, 'mmdet'))


# ------------------------------------------------------------------------------------------- mmcv stand-in
class Registry:

    def __init__(self, name, build_func=None, parent=None, scope=None):
        self.name = name
        self._module_dict = {}
        self.parent = parent
        self.build_func = build_func or build_from_cfg
        self.children = {}

    @property
    def module_dict(self):
        return self._module_dict

    def __contains__(self, key):
        return self.get(key) is not None

    def __len__(self):
        return len(self._module_dict)

    def get(self, key):
        if key in self._module_dict:
            return self._module_dict[key]
        if self.parent is not None:
            return self.parent.get(key)
        return None

    def build(self, *args, **kwargs):
        return self.build_func(*args, **kwargs, registry=self)

    def _register(self, cls, name=None, force=False):
        names = [name] if isinstance(name, str) else (name or [cls.__name__])
        for n in names:
            self._module_dict[n] = cls

    def register_module(self, name=None, force=False, module=None):
        if module is not None:
            self._register(module, name, force)
            return module
        if inspect.isclass(name) or inspect.isfunction(name):  # bare decorator
            self._register(name)
            return name

        def deco(cls):
            self._register(cls, name, force)
            return cls
        return deco


def build_from_cfg(cfg, registry, default_args=None):
    args = dict(cfg)
    if default_args:
        for k, v in default_args.items():
            args.setdefault(k, v)
    t = args.pop('type')
    cls = registry.get(t) if isinstance(t, str) else t
    if cls is None:
        raise KeyError(f'{t} is not in the {registry.name} registry')
    return cls(**args)


class ConfigDict(dict):
    """attribute-style dict (mmcv.ConfigDict stand-in)"""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return v

    def __setattr__(self, k, v):
        self[k] = v


def to_cfg(obj):
    if isinstance(obj, dict):
        return ConfigDict({k: to_cfg(v) for k, v in obj.items()})
    if isinstance(obj, (list, tuple)):
        return type(obj)(to_cfg(v) for v in obj)
    return obj


def _identity_decorator(*dargs, **dkw):
    if len(dargs) == 1 and callable(dargs[0]) and not dkw:
        return dargs[0]

    def deco(f):
        return f
    return deco


class BaseModule(nn.Module):

    def __init__(self, init_cfg=None):
        super().__init__()
        self._is_init = False
        self.init_cfg = init_cfg

    def init_weights(self):
        pass


class Sequential(BaseModule, nn.Sequential):

    def __init__(self, *args, init_cfg=None):
        BaseModule.__init__(self, init_cfg)
        nn.Sequential.__init__(self, *args)


class ModuleList(BaseModule, nn.ModuleList):

    def __init__(self, modules=None, init_cfg=None):
        BaseModule.__init__(self, init_cfg)
        nn.ModuleList.__init__(self, modules)


def build_conv_layer(cfg, *args, **kwargs):
    assert cfg is None or cfg.get('type', 'Conv2d') in ('Conv2d', 'Conv'), cfg
    return nn.Conv2d(*args, **kwargs)


def build_norm_layer(cfg, num_features, postfix=''):
    cfg = dict(cfg)
    t = cfg.pop('type')
    requires_grad = cfg.pop('requires_grad', True)
    cfg.setdefault('eps', 1e-5)
    assert t in ('BN', 'BN2d'), t
    layer = nn.BatchNorm2d(num_features, **cfg)
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return 'bn' + str(postfix), layer


def build_plugin_layer(*a, **k):
    raise NotImplementedError('plugins are not on the OA-DG path')


class ConvModule(nn.Module):
    """conv [+ norm] [+ ReLU], attribute names as mmcv's (conv / bn / activate)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias='auto', conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'), inplace=True,
                 with_spectral_norm=False, padding_mode='zeros', order=('conv', 'norm', 'act')):
        super().__init__()
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if bias == 'auto':
            bias = not self.with_norm
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                              bias)
        if self.with_norm:
            _, self.bn = build_norm_layer(norm_cfg, out_channels)
        if self.with_activation:
            assert act_cfg['type'] == 'ReLU'
            self.activate = nn.ReLU(inplace=inplace)
        # mmcv ConvModule.init_weights: kaiming for conv (relu), constant 1 for norm
        nn.init.kaiming_normal_(self.conv.weight, a=0, mode='fan_out', nonlinearity='relu')
        if self.conv.bias is not None:
            nn.init.constant_(self.conv.bias, 0)

    def forward(self, x, activate=True, norm=True):
        x = self.conv(x)
        if norm and self.with_norm:
            x = self.bn(x)
        if activate and self.with_activation:
            x = self.activate(x)
        return x


def constant_init(module, val, bias=0):
    if hasattr(module, 'weight') and module.weight is not None:
        nn.init.constant_(module.weight, val)
    if hasattr(module, 'bias') and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def kaiming_init(module, a=0, mode='fan_out', nonlinearity='relu', bias=0, distribution='normal'):
    nn.init.kaiming_normal_(module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def normal_init(module, mean=0, std=1, bias=0):
    nn.init.normal_(module.weight, mean, std)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def xavier_init(module, gain=1, bias=0, distribution='normal'):
    (nn.init.xavier_uniform_ if distribution == 'uniform' else nn.init.xavier_normal_)(module.weight, gain=gain)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def is_tuple_of(seq, expected_type):
    return isinstance(seq, tuple) and all(isinstance(s, expected_type) for s in seq)


def is_list_of(seq, expected_type):
    return isinstance(seq, list) and all(isinstance(s, expected_type) for s in seq)


class _Missing:
    """placeholder for an mmcv/cv2 name the reference imports but the OA-DG path never calls"""

    def __init__(self, name):
        self._name = name

    def __call__(self, *a, **k):
        raise NotImplementedError(f'{self._name} is not provided by the refload stand-in')

    def __getattr__(self, n):
        if n.startswith('__'):
            raise AttributeError(n)
        return _Missing(self._name + '.' + n)


class _StubModule(types.ModuleType):

    def __getattr__(self, n):
        if n.startswith('__'):
            raise AttributeError(n)
        return _Missing(self.__name__ + '.' + n)


def _mod(name, **attrs):
    m = _StubModule(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_mmcv(ops=None):
    MODELS = Registry('model')
    mm = _mod('mmcv', jit=_identity_decorator, is_tuple_of=is_tuple_of, is_list_of=is_list_of,
              ConfigDict=ConfigDict, __path__=[])
    utils = _mod('mmcv.utils', Registry=Registry, build_from_cfg=build_from_cfg, is_tuple_of=is_tuple_of,
                 print_log=lambda *a, **k: None, ConfigDict=ConfigDict, __path__=[],
                 TORCH_VERSION=torch.__version__, digit_version=lambda v: tuple(int(x) for x in v.split('+')[0].split('.')[:3]),
                 deprecated_api_warning=lambda *a, **k: (lambda f: f))
    _mod('mmcv.utils.parrots_wrapper', _BatchNorm=nn.modules.batchnorm._BatchNorm)
    mm.utils = utils
    cnn = _mod('mmcv.cnn', MODELS=MODELS, ConvModule=ConvModule, build_conv_layer=build_conv_layer,
               build_norm_layer=build_norm_layer, build_plugin_layer=build_plugin_layer,
               constant_init=constant_init, kaiming_init=kaiming_init, normal_init=normal_init,
               xavier_init=xavier_init, __path__=[],
               bricks=None)
    _mod('mmcv.cnn.utils', __path__=[])
    _mod('mmcv.cnn.utils.weight_init', constant_init=constant_init, kaiming_init=kaiming_init,
         normal_init=normal_init, xavier_init=xavier_init)
    _mod('mmcv.cnn.bricks', __path__=[])
    _mod('mmcv.cnn.bricks.transformer', __path__=[])
    mm.cnn = cnn
    runner = _mod('mmcv.runner', BaseModule=BaseModule, Sequential=Sequential, ModuleList=ModuleList,
                  auto_fp16=_identity_decorator, force_fp32=_identity_decorator, __path__=[],
                  get_dist_info=lambda: (0, 1))
    mm.runner = runner
    _mod('mmcv.parallel', DataContainer=object, __path__=[])
    ops = ops or {}
    opsmod = _mod('mmcv.ops', __path__=[], **ops)
    _mod('mmcv.ops.nms', **{k: v for k, v in ops.items() if 'nms' in k})
    mm.ops = opsmod
    return mm


# ------------------------------------------------------------------------------------------- mmdet shells
class _LazyPkg(types.ModuleType):
    """A package object that never runs the reference's ``__init__`` (those pull in every detector).
    Attribute lookups resolve to a sub-module, or to the reference file under this package that defines
    the name (found by an AST scan of top-level defs/classes/assignments/import-aliases)."""

    def __getattr__(self, name):
        if name.startswith('__'):
            raise AttributeError(name)
        full = self.__name__ + '.' + name
        d = self.__path__[0]
        if (self.__name__, name) in _OVERRIDES:
            val = getattr(importlib.import_module(_OVERRIDES[(self.__name__, name)]), name)
            self.__dict__[name] = val
            return val
        if os.path.isdir(os.path.join(d, name)) or os.path.isfile(os.path.join(d, name + '.py')):
            return importlib.import_module(full)
        owner = _OVERRIDES.get((self.__name__, name)) or self._index().get(name)
        if owner is None:
            raise AttributeError(f'{self.__name__}: reference does not define {name}')
        val = getattr(importlib.import_module(owner), name)
        setattr(self, name, val)
        return val

    def _index(self):
        idx = self.__dict__.get('_idx')
        if idx is None:
            idx = {}
            root = self.__path__[0]
            for dp, dn, fn in os.walk(root):
                dn.sort()
                for f in sorted(fn):
                    if not f.endswith('.py') or f == '__init__.py':
                        continue
                    rel = os.path.relpath(os.path.join(dp, f), root)[:-3].replace(os.sep, '.')
                    modname = self.__name__ + '.' + rel
                    try:
                        tree = ast.parse(open(os.path.join(dp, f)).read())
                    except SyntaxError:
                        continue
                    for node in tree.body:
                        names = []
                        if isinstance(node, (ast.FunctionDef, ast.ClassDef)):
                            names = [node.name]
                        elif isinstance(node, ast.Assign):
                            names = [t.id for t in node.targets if isinstance(t, ast.Name)]
                        for n in names:
                            idx.setdefault(n, modname)
            self.__dict__['_idx'] = idx
        return idx


# names defined in more than one reference file: which one the real package exports
_OVERRIDES = {
    ('mmdet.models.losses', 'accuracy'): 'mmdet.models.losses.accuracy',
    ('mmdet.core', 'bbox_overlaps'): 'mmdet.core.bbox.iou_calculators.iou2d_calculator',
    ('mmdet.core.bbox', 'bbox_overlaps'): 'mmdet.core.bbox.iou_calculators.iou2d_calculator',
    ('mmdet.models.losses', 'cross_entropy'): 'mmdet.models.losses.oadg.cross_entropy_loss_plus',
    ('mmdet.models.losses', 'binary_cross_entropy'): 'mmdet.models.losses.oadg.cross_entropy_loss_plus',
    ('mmdet.models.losses', 'smooth_l1_loss'): 'mmdet.models.losses.oadg.smooth_l1_loss_plus',
    ('mmdet.models.losses', 'l1_loss'): 'mmdet.models.losses.oadg.smooth_l1_loss_plus',
}


def install_mmdet_shells():
    root = os.path.join(REF, 'mmdet')
    for dp, dn, fn in os.walk(root):
        if '__init__.py' not in fn:
            dn[:] = []
            continue
        rel = os.path.relpath(dp, root)
        name = 'mmdet' if rel == '.' else 'mmdet.' + rel.replace(os.sep, '.')
        if name in sys.modules:
            continue
        pkg = _LazyPkg(name)
        pkg.__path__ = [dp]
        pkg.__package__ = name
        sys.modules[name] = pkg
    for name, mod in list(sys.modules.items()):
        if name.startswith('mmdet.') and isinstance(mod, _LazyPkg):
            parent, _, leaf = name.rpartition('.')
            setattr(sys.modules[parent], leaf, mod)
    return sys.modules['mmdet']


def install(cv2_module=None, ops=None):
    """Install every stand-in; idempotent per process."""
    if not available():
        raise RuntimeError(f'reference not found at {REF}')
    if 'mmdet' in sys.modules and isinstance(sys.modules['mmdet'], _LazyPkg):
        if cv2_module is not None:
            sys.modules['cv2'] = cv2_module
        if ops:
            for k, v in ops.items():
                setattr(sys.modules['mmcv.ops'], k, v)
                if 'nms' in k:
                    setattr(sys.modules['mmcv.ops.nms'], k, v)
        return
    install_mmcv(ops)
    sys.modules['cv2'] = cv2_module if cv2_module is not None else types.ModuleType('cv2')
    install_mmdet_shells()
    # heavy/irrelevant leaves imported at module import time by files on the path
    class _NoOps(types.ModuleType):
        def __getattr__(self, n):
            if n.startswith('__'):
                raise AttributeError(n)
            return lambda *a, **k: None
    sys.modules['mmdet.utils.visualize'] = _NoOps('mmdet.utils.visualize')
    _mod('mmdet.core.visualization', imshow_det_bboxes=None, __path__=[])
    _mod('mmdet.core.mask.structures', BitmapMasks=type('BitmapMasks', (), {}),
         PolygonMasks=type('PolygonMasks', (), {}))
    sys.modules['mmdet.core.mask'].BitmapMasks = sys.modules['mmdet.core.mask.structures'].BitmapMasks
    sys.modules['mmdet.core.mask'].PolygonMasks = sys.modules['mmdet.core.mask.structures'].PolygonMasks
    sys.modules['mmdet.core'].PolygonMasks = sys.modules['mmdet.core.mask.structures'].PolygonMasks
    sys.modules['mmdet.core'].BitmapMasks = sys.modules['mmdet.core.mask.structures'].BitmapMasks
    sys.modules['mmdet.utils'].get_root_logger = lambda *a, **k: __import__('logging').getLogger('mmdet')


def ref(module, name=None):
    """``ref('mmdet.models.losses.oadg.contrastive_loss', 'supcontrast')``"""
    m = importlib.import_module(module)
    return getattr(m, name) if name else m
