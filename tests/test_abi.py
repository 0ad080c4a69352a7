"""CPU: the C-ABI shared library loads and exports every symbol include/sdfx.h declares; the
Python operator packages keep the reference's call surface. No compute (no GPU here)."""
import ctypes
import inspect
import os
import re

import pytest
import torch

from conftest import ROOT


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "sdfx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sdfx_[a-zA-Z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import _sdfx
    if not os.path.exists(_sdfx.LIB_PATH):
        from importlib import import_module
        import_module("stable-dreamfusion_amd").build()
    lib = ctypes.CDLL(_sdfx.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 24
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/sdfx.h but not exported"
    # and the binding table covers the header
    missing = set(syms) - set(_sdfx.exported_symbols())
    assert not missing, missing
    assert b"gfx950" in _sdfx.lib().sdfx_build_info()


def test_ctypes_signatures_match_header_prototypes():
    """Every argtypes list in _sdfx.py agrees with the prototype in include/sdfx.h, parameter by parameter."""
    import ctypes as C
    import _sdfx
    text = open(os.path.join(ROOT, "include", "sdfx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = re.findall(r"\b(?:int|void|uint64_t|uint32_t|const char\*)\s+(sdfx_[a-zA-Z0-9_]+)\s*\(([^)]*)\)\s*;", text)
    assert len(protos) >= 24
    def ctype(decl):
        decl = decl.strip()
        if decl in ("void", ""):
            return None
        if "*" in decl or decl.startswith("sdfx_stream_t"):
            return C.c_void_p
        base = decl.split()[0] if not decl.startswith("const") else decl.split()[1]
        return {"uint32_t": C.c_uint32, "float": C.c_float, "double": C.c_double, "int": C.c_int, "uint64_t": C.c_uint64}[base]
    for name, params in protos:
        want = [c for c in (ctype(d) for d in params.split(",")) if c is not None]
        if name in ("sdfx_last_error", "sdfx_build_info"):
            continue
        assert name in _sdfx._SIGNATURES, name
        assert _sdfx._SIGNATURES[name] == want, (name, _sdfx._SIGNATURES[name], want)


def test_error_reporting_without_gpu():
    """Argument validation happens on the host before any launch."""
    import _sdfx
    lib = _sdfx.lib()
    rc = lib.sdfx_morton3D(None, 10, None, None)
    assert rc == -1 and b"null pointer" in lib.sdfx_last_error()
    with pytest.raises(RuntimeError, match="null pointer"):
        _sdfx.call("sdfx_packbits", None, 8, 0.5, None, None)
    rc = lib.sdfx_sh_encode_forward(ctypes.c_void_p(16), ctypes.c_void_p(16), 4, 3, 9, None, None)
    assert rc == -1 and b"degree in [1, 8]" in lib.sdfx_last_error()
    rc = lib.sdfx_grid_encode_forward(ctypes.c_void_p(16), ctypes.c_void_p(16), None, (ctypes.c_int32 * 3)(0, 8, 16),
                                      ctypes.c_void_p(16), 4, 3, 3, 2, 2, 1.0, 16, None, 0, 0, 0, 0, 0, None)
    assert rc == -3 and b"C must be" in lib.sdfx_last_error()


def test_ops_fail_loudly_on_cpu_tensors():
    import _raymarching, _gridencoder, _freqencoder, _shencoder
    x = torch.zeros(4, 3)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        _raymarching.morton3D(x.int(), 4, torch.zeros(4, dtype=torch.int32))
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        _freqencoder.freq_encode_forward(x, 4, 3, 1, 9, torch.zeros(4, 9))
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        _shencoder.sh_encode_forward(x, torch.zeros(4, 16), 4, 3, 4, None)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        _gridencoder.grid_encode_forward(x, torch.zeros(8, 2), torch.zeros(2, dtype=torch.int32), torch.zeros(1, 4, 2), 4, 3,
                                         2, 1, 1, 1.0, 16, None, 0, False, 0)


REFERENCE_SIGNATURES = {
    # raymarching/raymarching.py:34,67,97,120,144,172,200,264,326,377 (forward(ctx, ...) minus ctx)
    "near_far_from_aabb": ["rays_o", "rays_d", "aabb", "min_near=0.2"],
    "sph_from_ray": ["rays_o", "rays_d", "radius"],
    "morton3D": ["coords"],
    "morton3D_invert": ["indices"],
    "packbits": ["grid", "thresh", "bitfield=None"],
    "flatten_rays": ["rays", "M"],
    "march_rays_train": ["rays_o", "rays_d", "bound", "density_bitfield", "C", "H", "nears", "fars", "perturb=False",
                         "dt_gamma=0", "max_steps=1024", "contract=False"],
    "composite_rays_train": ["sigmas", "rgbs", "ts", "rays", "T_thresh=0.0001", "binarize=False"],
    "march_rays": ["n_alive", "n_step", "rays_alive", "rays_t", "rays_o", "rays_d", "bound", "density_bitfield", "C", "H",
                   "near", "far", "perturb=False", "dt_gamma=0", "max_steps=1024", "contract=False"],
    "composite_rays": ["n_alive", "n_step", "rays_alive", "rays_t", "sigmas", "rgbs", "ts", "weights_sum", "depth", "image",
                       "T_thresh=0.01", "binarize=False"],
}


def test_raymarching_surface_matches_reference():
    import raymarching
    import raymarching.raymarching as rm
    for name, want in REFERENCE_SIGNATURES.items():
        assert callable(getattr(raymarching, name))
        fn = getattr(rm, "_" + name).forward
        params = list(inspect.signature(fn).parameters.values())[1:]
        got = [p.name if p.default is inspect._empty else f"{p.name}={p.default}" for p in params]
        assert got[:len(want)] == want, (name, got)
        # anything beyond the reference's list must be optional (extensions)
        assert all(p.default is not inspect._empty for p in params[len(want):])


def test_backend_twins_expose_the_pybind_names():
    import _raymarching, _gridencoder, _freqencoder, _shencoder
    for n in ["flatten_rays", "packbits", "near_far_from_aabb", "sph_from_ray", "morton3D", "morton3D_invert",
              "march_rays_train", "composite_rays_train_forward", "composite_rays_train_backward", "march_rays",
              "composite_rays"]:
        assert callable(getattr(_raymarching, n))
    for n in ["grid_encode_forward", "grid_encode_backward", "grad_total_variation", "grad_weight_decay"]:
        assert callable(getattr(_gridencoder, n))
    assert callable(_freqencoder.freq_encode_forward) and callable(_freqencoder.freq_encode_backward)
    assert callable(_shencoder.sh_encode_forward) and callable(_shencoder.sh_encode_backward)


def test_encoder_modules_match_reference_layout(oracle):
    from gridencoder import GridEncoder
    from freqencoder import FreqEncoder
    from shencoder import SHEncoder
    g = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19,
                    desired_resolution=2048, gridtype="hash", align_corners=False, interpolation="smoothstep")
    offsets, pls = oracle.grid_offsets(desired_resolution=2048)
    assert g.offsets.dtype == torch.int32 and g.offsets.tolist() == offsets.tolist()
    assert g.embeddings.shape == (6098120, 2) and g.output_dim == 32 and int(g.n_params) == 12196240
    assert abs(g.per_level_scale - pls) < 1e-12 and float(g.embeddings.abs().max()) <= 1e-4
    assert set(dict(g.named_buffers())) == {"offsets"} and set(dict(g.named_parameters())) == {"embeddings"}
    assert FreqEncoder(3, 6).output_dim == 39 and SHEncoder(3, 4).output_dim == 16
    sig = inspect.signature(GridEncoder.__init__)
    assert list(sig.parameters)[1:] == ["input_dim", "num_levels", "level_dim", "per_level_scale", "base_resolution",
                                        "log2_hashmap_size", "desired_resolution", "gridtype", "align_corners",
                                        "interpolation"]
    assert list(inspect.signature(GridEncoder.forward).parameters)[1:] == ["inputs", "bound", "max_level"]


def test_row_limit_setter_is_host_only_state():
    """sdfx_set_row_limit only records a device pointer and a period for the calling thread: callable without a GPU, void."""
    import importlib
    importlib.import_module("stable-dreamfusion_amd")
    import _sdfx
    lib = _sdfx.lib()
    lib.sdfx_set_row_limit(ctypes.c_void_p(64), 4096)
    lib.sdfx_set_row_limit(None, 0)
    with _sdfx.row_limit(None, 123):          # total None: the context manager is a no-op
        pass


def test_product_library_has_no_switches_and_the_devtools_library_declares_its_own():
    """include/sdfx.h (product) declares no implementation switch and libsdfx_hip.so neither exports one nor reads the environment;
    libsdfx_hip_dev.so exports exactly what include/sdfx_devtools.h adds."""
    import subprocess
    import _sdfx
    from importlib import import_module
    pkg = import_module("stable-dreamfusion_amd")
    if not os.path.exists(_sdfx.DEV_LIB_PATH):
        pkg.build(devtools=True)
    product = os.path.join(os.path.dirname(_sdfx.DEV_LIB_PATH), "libsdfx_hip.so")
    if not os.path.exists(product):
        pkg.build()
    header = open(os.path.join(ROOT, "include", "sdfx.h")).read()
    assert "set_impl" not in header and "sdfx_dev_" not in header
    dev_header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "sdfx_devtools.h")).read(), flags=re.S)
    dev_syms = sorted(set(re.findall(r"\b(sdfx_[a-zA-Z0-9_]+)\s*\(", dev_header)))
    assert dev_syms == ["sdfx_dev_set", "sdfx_dev_stamps", "sdfx_dev_unset"]
    nm = lambda path: subprocess.run(["nm", "-D", path], capture_output=True, text=True, check=True).stdout
    prod, dev = nm(product), nm(_sdfx.DEV_LIB_PATH)
    assert " U getenv" not in prod, "the product library must not read the environment"
    for s in dev_syms:
        assert f" T {s}" in dev and f" T {s}" not in prod
    assert "set_impl" not in prod and "set_impl" not in dev
    exported = lambda text: set(re.findall(r" T (sdfx_[a-zA-Z0-9_]+)", text))
    assert exported(dev) - exported(prod) == set(dev_syms) and exported(prod) <= exported(dev)
