"""object_detectors_amd — MI355X-native hot path of kostas1515/object_detectors.

Python mirror of the reference's call sites over libmi355det.so (HIP, gfx950).  The package
holds only what the hot path needs: csrc/ (kernels + C ABI) and the host-side
mirror of the reference interfaces (yolo/, tvision/).
"""
from . import _lib  # noqa: F401
from ._lib import Mi355detError, lib  # noqa: F401

__version__ = "0.1.0"
