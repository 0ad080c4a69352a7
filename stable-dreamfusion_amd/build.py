"""Build libsdfx_hip.so (the C-ABI HIP library) in-tree for gfx950 with hipcc.

    python stable-dreamfusion_amd/build.py [--force] [--verbose] [--devtools]

--devtools builds libsdfx_hip_dev.so instead: the same sources with -DSDFX_DEVTOOLS (implementation switches, measurement
knobs, superseded kernels; include/sdfx_devtools.h). The product library has none of them.

hipcc cross-compiles without a GPU; the resulting .so sits next to the sources (git-ignored,
but shipped to the GPU box with the repository snapshot).
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.normpath(os.path.join(HERE, "..", "include"))
# superseded / measurement-only kernels that only the devtools library compiles in (field_dot2.inc.h, attention_pipe.inc.h): they
# live outside the product source tree
DEVTOOLS_KERNELS = os.path.normpath(os.path.join(HERE, "..", "tools", "devtools_kernels"))
LIB = os.path.join(CSRC, "libsdfx_hip.so")
DEV_LIB = os.path.join(CSRC, "libsdfx_hip_dev.so")
SOURCES = ["sdfx_core.hip", "raymarching.hip", "gridencoder.hip", "gridencoder_fwd.hip", "gridencoder_bwd_binned.hip", "encoders.hip", "field.hip", "optim.hip", "shade.hip", "render.hip", "occupancy.hip", "infer.hip", "head.hip", "sds.hip", "dmtet.hip", "raster.hip", "groupnorm.hip", "conv.hip", "attention.hip"]
ARCH = "gfx950"
# -ffp-contract=off: the march / encode arithmetic must not gain FMAs the source does not spell
# out (bit-exact ray counts and fp32 features against the CPU oracle).
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-Wno-pass-failed", "-I", INCLUDE]


def hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built")
    return exe


def _deps_mtime(devtools: bool = False) -> float:
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(INCLUDE, "sdfx.h")]
    if devtools and os.path.isdir(DEVTOOLS_KERNELS):
        hdrs += [os.path.join(DEVTOOLS_KERNELS, f) for f in os.listdir(DEVTOOLS_KERNELS) if f.endswith(".h")]
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src: str, force: bool, verbose: bool, devtools: bool = False) -> str:
    path = os.path.join(CSRC, src)
    obj = os.path.join(CSRC, "build_dev" if devtools else "build", src.replace(".hip", ".o"))
    os.makedirs(os.path.dirname(obj), exist_ok=True)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(path), _deps_mtime(devtools)):
        return obj
    cmd = [hipcc()] + FLAGS + (["-DSDFX_DEVTOOLS", "-I", DEVTOOLS_KERNELS] if devtools else []) + ["-c", path, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return obj


def build(force: bool = False, verbose: bool = False, devtools: bool = False) -> str:
    sources = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    lib = DEV_LIB if devtools else LIB
    with cf.ThreadPoolExecutor(max_workers=min(len(sources), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, verbose, devtools), sources))
    if force or not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(o) for o in objs):
        cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv or "-v" in sys.argv, devtools="--devtools" in sys.argv))
