"""MI355X-native implementation of the NVSF volumetric-rendering hot path.

Mirrors the module layout of the reference for the path it replaces
(`nvsf.nerf.raymarching.raymarching`, `nvsf.nerf.models.*`, `nvsf.nerf.activation`); unlike the
reference's `nvsf/__init__.py:11` nothing heavy is imported eagerly.
"""
__version__ = "0.1.0"
