"""MI355X-native FastSLAM-1.0 particle update behind parakeet_slam's class surface.

    from parakeet_slam_amd import FastSLAM, FilterParticle, Feature

mirrors ``from prkt_core_v2 import FastSLAM, FilterParticle, Feature`` of the reference
(buckbaskin/parakeet_slam, src/prkt_core_v2.py).  The arithmetic runs in hand-written HIP
kernels for gfx950 reached through the C ABI in ``include/parakeet_slam.h``.
"""
from ._lib import DeviceFilter, HostRng, PkError, probe  # noqa: F401
from .core import FastSLAM, Feature, FilterParticle  # noqa: F401
from .multi import ShardedFastSLAM  # noqa: F401  (FastSLAM(..., devices=[...]): one child process per GPU)

__all__ = ["FastSLAM", "FilterParticle", "Feature", "DeviceFilter", "PkError", "probe", "HostRng", "ShardedFastSLAM"]
