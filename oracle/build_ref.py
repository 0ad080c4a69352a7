#!/usr/bin/env python3
"""Build oracle/_ref/: the REFERENCE's own CUDA extensions, compiled in place for gfx950.

Test infrastructure only (cross-check of the oracle and of the HIP kernels against the reference's
kernels running on the same MI355X). Sources are read where they lie under /root/reference — nothing
is copied or translated (no hipify): hipcc compiles the .cu files as HIP with an include-path shim
(oracle/ref_shim/) that maps <cuda.h>, <cuda_runtime.h>, <cuda_fp16.h> onto the HIP headers. Outputs go
only to oracle/_ref/ (git-ignored, shipped to the GPU box with the snapshot).

Two variants per extension:
    _ref_<name>.so      default device flags (-ffp-contract=fast, like nvcc's -fmad=true)
    _refnc_<name>.so    -ffp-contract=off (to separate algorithmic differences from FMA contraction)
"""
import os
import shutil
import subprocess
import sys
import sysconfig

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
SHIM = os.path.join(HERE, "ref_shim")
EXTS = {
    "raymarching": ["raymarching/src/raymarching.cu", "raymarching/src/bindings.cpp"],
    "gridencoder": ["gridencoder/src/gridencoder.cu", "gridencoder/src/bindings.cpp"],
    "shencoder": ["shencoder/src/shencoder.cu", "shencoder/src/bindings.cpp"],
    "freqencoder": ["freqencoder/src/freqencoder.cu", "freqencoder/src/bindings.cpp"],
}


def build_one(name, srcs, variant, extra):
    import pybind11
    import torch
    from torch.utils import cpp_extension as ce
    modname = f"{variant}_{name}"
    out = os.path.join(OUT, modname + ".so")
    newest = max(os.path.getmtime(os.path.join(REF, s)) for s in srcs)
    if os.path.exists(out) and os.path.getmtime(out) > newest:
        return out
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    inc = [SHIM, os.path.join(REF, os.path.dirname(srcs[0]))] + ce.include_paths() + [pybind11.get_include(),
                                                                                    sysconfig.get_paths()["include"]]
    cmd = [shutil.which("hipcc") or "/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DHIPBLAS_V2", "-DCUDA_HAS_FP16=1", "-D_GLIBCXX_USE_CXX11_ABI=1",
           "-DTORCH_API_INCLUDE_EXTENSION_H", f"-DTORCH_EXTENSION_NAME={modname}", "-w"] + extra
    for i in inc:
        cmd += ["-I", i]
    for s in srcs:
        cmd += ["-x", "hip", os.path.join(REF, s)]
    cmd += ["-o", out, "-L", tlib, "-lc10", "-lc10_hip", "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-ltorch_python",
            f"-Wl,-rpath,{tlib}"]
    print("[build_ref]", modname, flush=True)
    subprocess.check_call(cmd)
    return out


def main():
    if not os.path.isdir(REF):
        print("[build_ref] /root/reference not mounted; nothing to do")
        return 0
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[1:] or list(EXTS)
    rc = 0
    for name in only:
        for variant, extra in (("_ref", []), ("_refnc", ["-ffp-contract=off"])):
            try:
                build_one(name, EXTS[name], variant, extra)
            except subprocess.CalledProcessError as exc:
                print(f"[build_ref] {variant}_{name} FAILED ({exc.returncode}); this cross-check will be skipped")
                rc = 1
    return rc


if __name__ == "__main__":
    sys.exit(main())
