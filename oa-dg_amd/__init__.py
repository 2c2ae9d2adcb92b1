"""oadg_amd - MI355X-native implementation of OA-DG's training hot path.

OA-Mix augmentation -> Faster R-CNN forward/backward -> OA-Loss, behind the reference's
registry/config surface.  The arithmetic of the named hot ops lives in
``csrc/liboadg_hip.so`` (hand-written HIP for gfx950, C ABI in ``include/oadg_hip.h``); this
package is the host-side mirror of the reference's Python interface.  There is no CPU fallback:
calling a hot op without the library or off-GPU raises.
"""
__version__ = '0.1.0'

from . import _lib  # noqa: F401
from . import registry  # noqa: F401
from .config import Config, ConfigDict  # noqa: F401
from .registry import (MODELS, PIPELINES, build_detector, build_from_cfg)  # noqa: F401
# importing the modules populates the registries with the reference's type strings
from . import core, losses, backbones, necks, dense_heads, roi_heads, detectors, pipelines, datasets  # noqa: F401,E402
