"""samplenet_amd -- MI355X-native implementation of the SampleNet differentiable-sampling hot path.

Public surface mirrors `registration/src/__init__.py` of itailang/SampleNet for this path:
    from samplenet_amd import ChamferDistance, SoftProjection, SampleNet, sputils
Importing the package loads libsamplenet_hip.so (hand-written HIP for gfx950) and fails loudly if it
is missing -- there is no CPU or eager-PyTorch fallback.
"""
from . import _lib  # noqa: F401  (loads the HIP library or raises)
from . import ops, sputils  # noqa: F401
from .chamfer_distance import ChamferDistance, ChamferDistanceFunction  # noqa: F401
from .progressive import SampleNetProgressive, progressive_sizes  # noqa: F401
from .samplenet import SampleNet  # noqa: F401
from .soft_projection import SoftProjection  # noqa: F401

__all__ = ["ChamferDistance", "ChamferDistanceFunction", "SoftProjection", "SampleNet", "SampleNetProgressive",
           "progressive_sizes", "sputils", "ops"]
