"""Occupancy-grid ray marching and volume compositing on MI355X (libsdfx_hip.so).

`import raymarching` exposes the operators nerf/renderer.py calls — near_far_from_aabb, sph_from_ray, morton3D,
morton3D_invert, packbits, flatten_rays, march_rays_train, composite_rays_train, march_rays, composite_rays — plus
this repository's extensions (compact_rays, march_rays_train_count / march_rays_train_write / march_rays_train_stage_write).
"""
from . import raymarching as _ops
from .raymarching import __all__ as _exported

globals().update({name: getattr(_ops, name) for name in _exported})
__all__ = list(_exported)
