"""The ONE place where the Python half of the package looks at A/B switches.

Product behaviour: every switch has the default written at its use; the environment is NOT read. In a devtools session —
`SDFX_DEV=1`, or `SDFX_LIB` selecting another build of the library (libsdfx_hip_dev.so: include/sdfx_devtools.h) — an
`SDFX_<NAME>` environment variable overrides that default, read ONCE when the module that owns the switch is imported (the value
lands in a module attribute, e.g. `sdfx_nerf.conv._FUSED`; tests and bench.py flip those attributes directly). The switches of
the LIBRARY are the devtools library's own business (`sdfx_dev_set`, csrc/sdfx_common.h: `dev_switch`)."""
from __future__ import annotations

import os

DEV = os.environ.get("SDFX_DEV") == "1" or bool(os.environ.get("SDFX_LIB"))


_warned = set()


def get(name: str, default):
    """The switch `name`: `default` in the product, the environment's value (converted to the default's type) in a devtools session.
    A variable of that name set OUTSIDE a devtools session is ignored — with one warning per name, so that an A/B script that forgot
    `SDFX_DEV=1` does not silently compare a configuration with itself."""
    if not DEV:
        if os.environ.get(name) not in (None, "") and name not in _warned:
            _warned.add(name)
            import warnings
            warnings.warn(f"{name}={os.environ[name]} is ignored: the package reads SDFX_* switches only in a devtools session "
                          f"(SDFX_DEV=1, or SDFX_LIB selecting another build)", stacklevel=2)
        return default
    v = os.environ.get(name)
    if v is None or v == "":
        return default
    return type(default)(v)
