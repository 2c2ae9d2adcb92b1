"""Python-file configs with ``_base_`` inheritance, the reference's config surface.

Stands in for mmcv.Config as used by tools/train.py:95-97: ``Config.fromfile`` executes a ``.py`` file,
merges its ``_base_`` list depth-first (dicts merge key-wise, ``_delete_=True`` replaces), supports
``--cfg-options a.b=c`` overrides and ``custom_imports``.  The OA-DG configs inherit through the authors'
docker mount ``/ws/external/...`` (configs/OA-DG/cityscapes/faster_rcnn_r50_fpn_1x_cityscapes_oadg.py:2);
that prefix is remapped to the tree the config file lives in (or $OADG_CONFIG_ROOT).
"""
import ast
import copy
import importlib
import os

DELETE_KEY = '_delete_'
BASE_KEY = '_base_'
WS_PREFIX = '/ws/external/'


class ConfigDict(dict):
    """dict with attribute access (missing attribute -> AttributeError, like mmcv.ConfigDict)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(f"'ConfigDict' object has no attribute '{name}'")

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        del self[name]

    def __deepcopy__(self, memo):
        return ConfigDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def _wrap(obj):
    if isinstance(obj, dict):
        return ConfigDict({k: _wrap(v) for k, v in obj.items()})
    if isinstance(obj, list):
        return [_wrap(v) for v in obj]
    if isinstance(obj, tuple):
        return tuple(_wrap(v) for v in obj)
    return obj


def _merge(a, b):
    """b into a copy of a (mmcv Config._merge_a_into_b semantics)."""
    out = dict(a)
    for k, v in b.items():
        if isinstance(v, dict):
            v = dict(v)
            if v.pop(DELETE_KEY, False) or not isinstance(out.get(k), dict):
                out[k] = _merge({}, v)
            else:
                out[k] = _merge(out[k], v)
        else:
            out[k] = v
    return out


def _config_root(path):
    env = os.environ.get('OADG_CONFIG_ROOT')
    if env:
        return env
    d = os.path.dirname(os.path.abspath(path))
    while d != os.path.dirname(d):
        if os.path.basename(d) == 'configs':
            return os.path.dirname(d)
        d = os.path.dirname(d)
    return os.getcwd()


def _resolve_base(base, cur_file):
    if base.startswith(WS_PREFIX):
        return os.path.join(_config_root(cur_file), base[len(WS_PREFIX):])
    if os.path.isabs(base):
        return base
    return os.path.join(os.path.dirname(os.path.abspath(cur_file)), base)


def _exec_file(path):
    src = open(path).read()
    ast.parse(src, path)
    ns = {'__file__': path, '__name__': '__oadg_config__'}
    exec(compile(src, path, 'exec'), ns)
    import types
    return {k: v for k, v in ns.items()
            if (not k.startswith('_') or k in (BASE_KEY, DELETE_KEY))
            and not isinstance(v, (types.ModuleType, types.FunctionType, type))}


def _load(path):
    if not os.path.isfile(path):
        raise FileNotFoundError(path)
    cfg = _exec_file(path)
    bases = cfg.pop(BASE_KEY, [])
    if isinstance(bases, str):
        bases = [bases]
    merged = {}
    for b in bases:
        bcfg = _load(_resolve_base(b, path))
        dup = set(merged) & set(bcfg)
        if dup:
            raise KeyError(f'duplicate key(s) {sorted(dup)} in the _base_ files of {path}')
        merged.update(bcfg)
    return _merge(merged, cfg)


def _parse_value(s):
    try:
        return ast.literal_eval(s)
    except (ValueError, SyntaxError):
        return s


class Config:
    """``cfg = Config.fromfile(path); cfg.model.backbone.depth; cfg.merge_from_dict({'a.b': 1})``"""

    def __init__(self, cfg_dict=None, filename=None):
        object.__setattr__(self, '_cfg_dict', _wrap(cfg_dict or {}))
        object.__setattr__(self, 'filename', filename)

    @staticmethod
    def fromfile(filename, import_custom_modules=True):
        cfg = Config(_load(filename), filename)
        ci = cfg.get('custom_imports')
        if import_custom_modules and ci:
            for mod in ci.get('imports', []):
                try:
                    importlib.import_module(_remap_module(mod))
                except ImportError:
                    if not ci.get('allow_failed_imports', False):
                        raise
        return cfg

    def merge_from_dict(self, options):
        """``--cfg-options`` style: {'model.backbone.depth': 101, 'data.samples_per_gpu': '4'}."""
        nested = {}
        for full, v in options.items():
            d = nested
            keys = full.split('.')
            for k in keys[:-1]:
                d = d.setdefault(k, {})
            d[keys[-1]] = _parse_value(v) if isinstance(v, str) else v
        object.__setattr__(self, '_cfg_dict', _wrap(_merge(self._cfg_dict, nested)))

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __setattr__(self, name, value):
        self._cfg_dict[name] = _wrap(value)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __contains__(self, name):
        return name in self._cfg_dict

    def get(self, name, default=None):
        return self._cfg_dict.get(name, default)

    def to_dict(self):
        return copy.deepcopy(dict(self._cfg_dict))


def _remap_module(mod):
    """custom_imports of the reference name mmdet modules; our mirror lives under oadg_amd."""
    table = {'mmdet.datasets.pipelines.oa_mix': 'oadg_amd.pipelines.oa_mix'}
    return table.get(mod, mod)
