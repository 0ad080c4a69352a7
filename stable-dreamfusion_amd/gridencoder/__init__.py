"""Multi-resolution hash / tiled grid encoder on MI355X (libsdfx_hip.so): `from gridencoder import GridEncoder`."""
from .grid import GridEncoder, grid_encode

__all__ = ["GridEncoder", "grid_encode"]
