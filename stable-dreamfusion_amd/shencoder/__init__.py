"""Real spherical-harmonics direction encoding on MI355X (libsdfx_hip.so): `from shencoder import SHEncoder`."""
from .sphere_harmonics import SHEncoder, sh_encode

__all__ = ["SHEncoder", "sh_encode"]
