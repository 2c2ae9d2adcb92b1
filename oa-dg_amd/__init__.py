"""oadg_amd - MI355X-native implementation of OA-DG's training hot path.

OA-Mix augmentation -> Faster R-CNN forward/backward -> OA-Loss, behind the reference's
registry/config surface.  The arithmetic of the named hot ops lives in
``csrc/liboadg_hip.so`` (hand-written HIP for gfx950, C ABI in ``include/oadg_hip.h``); this
package is the host-side mirror of the reference's Python interface.  There is no CPU fallback:
calling a hot op without the library or off-GPU raises.
"""
__version__ = '0.1.0'

from . import _lib  # noqa: F401
