"""MI355X-native hot path of stable-dreamfusion: the `raymarching`, `gridencoder`,
`freqencoder` and `shencoder` operator packages (same Python surface as the reference's),
their pybind-level twins (`_raymarching`, ...), and libsdfx_hip.so underneath.

The directory name contains a hyphen, so import it through `importlib`:

    import importlib, sys
    sys.path.insert(0, "<repo root>")
    importlib.import_module("stable-dreamfusion_amd")      # puts the operator packages on sys.path
    import raymarching, gridencoder                        # exactly what the reference imports

Putting this directory itself on sys.path (or running from inside it) works as well; that is
how the reference resolves its extensions (cwd on sys.path, nerf/renderer.py:14).
"""
import os as _os
import sys as _sys

PACKAGE_DIR = _os.path.dirname(_os.path.abspath(__file__))
if PACKAGE_DIR not in _sys.path:
    _sys.path.insert(0, PACKAGE_DIR)

OPERATOR_PACKAGES = ("raymarching", "gridencoder", "freqencoder", "shencoder")


def build(force: bool = False, verbose: bool = False, devtools: bool = False) -> str:
    """Compile libsdfx_hip.so for gfx950 (hipcc). Returns its path."""
    import importlib.util as _u
    spec = _u.spec_from_file_location("_sdfx_build", _os.path.join(PACKAGE_DIR, "build.py"))
    mod = _u.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build(force=force, verbose=verbose, devtools=devtools)
