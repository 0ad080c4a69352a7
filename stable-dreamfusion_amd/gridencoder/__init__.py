from .grid import GridEncoder  # noqa: F401
