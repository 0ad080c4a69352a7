from .sphere_harmonics import SHEncoder  # noqa: F401
