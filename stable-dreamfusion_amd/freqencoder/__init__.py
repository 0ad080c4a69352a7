from .freq import FreqEncoder  # noqa: F401
