"""Sin/cos positional encoding on MI355X (libsdfx_hip.so): `from freqencoder import FreqEncoder`."""
from .freq import FreqEncoder, freq_encode

__all__ = ["FreqEncoder", "freq_encode"]
