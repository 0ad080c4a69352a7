#!/usr/bin/env python3
"""Build a VARIANT of the devtools library for a same-box A/B: one translation unit recompiled with extra -D flags, linked with the
regular devtools objects into ab/libsdfx_hip_<name>.so (git-ignored, shipped to the GPU box; select with SDFX_LIB).

    python tools/build_variant.py b12t1024 gridencoder_bwd_binned.hip -DSDFX_BUCKET_LOG2=12 -DSDFX_BIN_THREADS=1024
"""
import importlib.util, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("_b", os.path.join(ROOT, "stable-dreamfusion_amd", "build.py"))
B = importlib.util.module_from_spec(spec); spec.loader.exec_module(B)
name, unit, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build(devtools=True)                                   # the regular devtools objects
objs = [os.path.join(B.CSRC, "build_dev", s.replace(".hip", ".o")) for s in B.SOURCES if s != unit]
os.makedirs(os.path.join(ROOT, "ab"), exist_ok=True)
obj = os.path.join(ROOT, "ab", f"{name}_{unit.replace('.hip', '.o')}")
subprocess.check_call([B.hipcc()] + B.FLAGS + ["-DSDFX_DEVTOOLS", "-I", B.DEVTOOLS_KERNELS] + flags + ["-c", os.path.join(B.CSRC, unit), "-o", obj])
lib = os.path.join(ROOT, "ab", f"libsdfx_hip_{name}.so")
subprocess.check_call([B.hipcc(), "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", lib] + objs + [obj])
print(lib)
