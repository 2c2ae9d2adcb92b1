"""Import shim: exposes the package directory ``oa-dg_amd/`` (not a valid Python identifier) as the
importable package ``oadg_amd``.  ``import oadg_amd`` returns the real package object."""
import importlib.util
import os
import sys

_dir = os.environ.get('OADG_PKG_DIR') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'oa-dg_amd')   # (override: A/B probes)
_spec = importlib.util.spec_from_file_location(
    'oadg_amd', os.path.join(_dir, '__init__.py'), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['oadg_amd'] = _mod
_spec.loader.exec_module(_mod)
