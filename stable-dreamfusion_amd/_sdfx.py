"""ctypes binding of libsdfx_hip.so (C ABI declared in include/sdfx.h).

This is the only place the shared library is opened.  There is no fallback: if the HIP
library has not been built, or an entry point reports an error, a RuntimeError is raised —
nothing in the shipped path routes around the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.realpath(__file__))  # realpath: backends_only/ holds symlinks to these files
# (SDFX_LIB: a differently configured build of the same library, for A/B measurements)
LIB_PATH = os.environ.get("SDFX_LIB") or os.path.join(_HERE, "csrc", "libsdfx_hip.so")

_u32, _f32, _int, _ptr, _u64 = C.c_uint32, C.c_float, C.c_int, C.c_void_p, C.c_uint64

# name -> argtypes (every entry point returns int unless listed in _RESTYPES)
_SIGNATURES = {
    "sdfx_xcd_round_robin": [],
    "sdfx_near_far_from_aabb": [_ptr, _ptr, _ptr, _u32, _f32, _ptr, _ptr, _ptr],
    "sdfx_sph_from_ray": [_ptr, _ptr, _f32, _u32, _ptr, _ptr],
    "sdfx_morton3D": [_ptr, _u32, _ptr, _ptr],
    "sdfx_morton3D_invert": [_ptr, _u32, _ptr, _ptr],
    "sdfx_packbits": [_ptr, _u32, _f32, _ptr, _ptr],
    "sdfx_flatten_rays": [_ptr, _u32, _u32, _ptr, _ptr],
    "sdfx_march_rays_train": [_ptr, _ptr, _ptr, _f32, _int, _f32, _u32, _u32, _u32, _u32, _ptr, _ptr, _ptr, _ptr, _ptr,
                              _ptr, _ptr, _ptr, _ptr, _ptr],
    "sdfx_march_rays_train_stage_write": [_ptr, _ptr, _f32, _int, _f32, _u32, _u32, _u32, _u32, _ptr, _ptr, _ptr, _u32, _ptr, _ptr,
                                          _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    "sdfx_march_rays_train_scratch_bytes": [_u32, _u32],
    "sdfx_composite_rays_train_forward": [_ptr, _ptr, _ptr, _ptr, _u32, _u32, _f32, _int, _ptr, _ptr, _ptr, _ptr, _ptr],
    "sdfx_composite_rays_train_backward": [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _u32, _u32,
                                           _f32, _int, _ptr, _ptr, _ptr],
    "sdfx_march_rays": [_u32, _u32, _ptr, _ptr, _ptr, _ptr, _f32, _int, _f32, _u32, _u32, _u32, _ptr, _ptr, _ptr, _ptr,
                        _ptr, _ptr, _ptr, _ptr],
    "sdfx_composite_rays": [_u32, _u32, _f32, _int, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    "sdfx_compact_rays": [_ptr, _u32, _ptr, _ptr, _ptr, _ptr],
    "sdfx_compact_rays_scratch_bytes": [_u32],
    "sdfx_grid_encode_forward": [_ptr, _ptr, _ptr, _ptr, _ptr, _u32, _u32, _u32, _u32, _u32, _f32, _u32, _ptr, _u32,
                                 _int, _u32, _int, _int, _ptr],
    "sdfx_grid_encode_forward_hint": [_ptr, _ptr, _ptr, _ptr, _ptr, _u32, _u32, _u32, _u32, _u32, _f32, _u32, _ptr, _u32,
                                 _int, _u32, _int, _int, _u32, _f32, _ptr],
    "sdfx_grid_encode_backward": [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _u32, _u32, _u32, _u32, _u32, _f32, _u32, _ptr,
                                  _ptr, _u32, _int, _u32, _int, _int, _ptr],
    "sdfx_grid_encode_backward_binned": [_ptr, _ptr, _ptr, _ptr, _u32, _u32, _u32, _u32, _u32, _f32, _u32, _u32, _int, _u32,
                                         _int, _int, _ptr, _u64, _ptr],
    "sdfx_grid_encode_backward_binned_scratch_bytes": [_ptr, _u32, _u32, _f32, _u32, _u32, _int],
    "sdfx_grid_encode_backward_binned_stats": [_ptr, _ptr, _ptr],
    "sdfx_grad_total_variation": [_ptr, _ptr, _ptr, _ptr, _ptr, _f32, _u32, _u32, _u32, _u32, _f32, _u32, _u32, _int,
                                  _int, _ptr],
    "sdfx_grad_weight_decay": [_ptr, _ptr, _ptr, _f32, _u32, _u32, _u32, _int, _ptr],
    "sdfx_freq_encode_forward": [_ptr, _u32, _u32, _u32, _u32, _ptr, _ptr],
    "sdfx_freq_encode_backward": [_ptr, _ptr, _u32, _u32, _u32, _u32, _ptr, _ptr],
    "sdfx_sh_encode_forward": [_ptr, _ptr, _u32, _u32, _u32, _ptr, _ptr],
    "sdfx_sh_encode_backward": [_ptr, _ptr, _u32, _u32, _u32, _ptr, _ptr, _ptr],
    "sdfx_field_packed_words": [],
    "sdfx_field_backward_scratch_bytes": [_u32],
    "sdfx_field_stencil_points": [_ptr, _u32, _f32, _f32, C.c_double, _ptr, _ptr, _ptr],
    "sdfx_field_pack": [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    "sdfx_field_forward": [_ptr, _int, _ptr, _ptr, _u32, _f32, _f32, _ptr, _ptr, _ptr],
    "sdfx_field_backward": [_ptr, _int, _ptr, _ptr, _u32, _f32, _f32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr,
                            _ptr, _ptr],
    "sdfx_shade_forward": [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _int, _f32, _u32, _u32, _ptr, _ptr, _ptr, _ptr, _ptr],
    "sdfx_shade_backward": [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _int, _f32, _u32, _u32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr,
                            _ptr],
    "sdfx_head_scratch_bytes": [_u32],
    "sdfx_head_forward": [_ptr] * 11 + [_f32, _f32, _u32, _u32, _ptr, _ptr, _ptr, _ptr],
    "sdfx_head_backward": [_ptr] * 11 + [_f32, _f32, _u32, _u32] + [_ptr] * 11,
    "sdfx_sds_add_noise": [_ptr, _int, _int, _ptr, _ptr, _ptr, _u32, _u32, _ptr, _ptr, _ptr, _ptr],
    "sdfx_sds_loss": [_ptr, _ptr, _ptr, _int, _ptr, _ptr, _f32, _f32, _f32, _u32, _u32, _u32, _ptr, _ptr, _ptr],
    "sdfx_sds_upsample_forward": [_ptr, _u32, _u32, _u32, _u32, _u32, _int, _int, _ptr, _ptr],
    "sdfx_sds_upsample_backward": [_ptr, _int, _u32, _u32, _u32, _u32, _u32, _int, _ptr, _ptr],
    "sdfx_sds_text_mix": [_ptr] * 7 + [_u32, _ptr, _ptr],
    "sdfx_render_infer": [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _f32, _int, _f32, _u32, _u32, _u32, _u32, _ptr, _ptr, _u32, _f32, _u32,
                          _u32, _int, _u32, _ptr, _f32, _f32, _f32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    "sdfx_occupancy_points": [_u32, C.c_double, _ptr, _u64, _u32, _ptr, _ptr],
    "sdfx_occupancy_update": [_ptr, _ptr, _u32, _f32, _ptr, _int, _ptr],
    "sdfx_occupancy_pack": [_ptr, _u32, _ptr, _f32, _ptr, _ptr, _ptr],
    "sdfx_render_train_forward": [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _int, _f32, _f32, _u32, _u32, _ptr,
                                  _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    "sdfx_render_train_backward": [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _int, _f32, _f32, _u32, _u32, _ptr,
                                   _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    "sdfx_entropy_forward": [_ptr, _u32, _ptr, _ptr, _ptr],
    "sdfx_entropy_backward": [_ptr, _u32, _ptr, _ptr, _ptr, _ptr],
    "sdfx_set_row_limit": [_ptr, _u32],
    "sdfx_set_albedo_rows": [_u32],
    "sdfx_field_albedo_rows_ok": [_u32, _int],
    "sdfx_set_stencil_source": [_ptr, _u32, _f32, _f32, C.c_double],
    "sdfx_grid_backward_plan": [_ptr, _u32, _f32, _u32, _u32, _int, _ptr, _ptr],
    "sdfx_grid_forward_plan": [_ptr, _u32, _f32, _u32, _int, _u32, _u32, _f32, _ptr, _u32, _ptr],
    "sdfx_grid_forward_level_costs": [_ptr, _u32, _f32, _u32, _u32, _f32, _ptr],
    "sdfx_marching_tets_scratch_bytes": [_u32, _u32],
    "sdfx_marching_tets_count": [_ptr, _ptr, _u32, _ptr, _u32, _ptr, _ptr, _ptr],
    "sdfx_marching_tets_emit": [_ptr, _ptr, _ptr, _u32, _ptr, _ptr, _u32, _ptr, _ptr, _ptr, _ptr, _ptr, _u32, _ptr, _u32, _ptr],
    "sdfx_marching_tets_backward": [_ptr, _u32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr],
    "sdfx_rasterize_scratch_bytes": [_u32, _u32],
    "sdfx_rasterize_forward": [_ptr, _ptr, _u32, _u32, _u32, _u32, _ptr, _ptr, _ptr],
    "sdfx_rasterize_backward": [_ptr, _ptr, _u32, _u32, _u32, _ptr, _ptr, _ptr, _ptr],
    "sdfx_interpolate_forward": [_ptr, _ptr, _u32, _u32, _u32, _ptr, _ptr, _ptr],
    "sdfx_interpolate_backward": [_ptr, _ptr, _u32, _u32, _u32, _ptr, _ptr, _ptr, _ptr, _ptr],
    "sdfx_antialias_forward": [_ptr, _ptr, _ptr, _ptr, _ptr, _u32, _u32, _u32, _u32, _ptr, _ptr],
    "sdfx_antialias_backward": [_ptr, _ptr, _ptr, _ptr, _ptr, _u32, _u32, _u32, _u32, _ptr, _ptr, _ptr, _ptr],
    "sdfx_group_norm_scratch_bytes": [_u32, _u32, _u32, _u32],
    "sdfx_group_norm_forward": [_ptr, _ptr, _ptr, _ptr, _u32, _u32, _u32, _u32, _f32, _int, _ptr, _ptr, _ptr, _ptr],
    "sdfx_group_norm_backward": [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _u32, _u32, _u32, _u32, _int, _ptr, _ptr, _ptr],
    "sdfx_add_bias_residual": [_ptr, _ptr, _ptr, _u32, _u32, _u32, _ptr, _ptr],
    "sdfx_geglu": [_ptr, _u64, _u32, _ptr, _ptr],
    "sdfx_attention_forward": [_ptr, _ptr, _ptr, _u32, _u32, _u32, _u32, _u32, _ptr, _ptr, _ptr, _f32, _int, _ptr, _ptr],
    "sdfx_conv3x3_packed_ok": [_u32, _u32, _u32, _u32, _u32, _u32],
    "sdfx_conv3x3_packed_scratch_bytes": [_u32, _u32, _u32, _u32, _u32, _u32, _int],
    "sdfx_conv3x3_pack_weights": [_ptr, _u32, _u32, _ptr, _ptr],
    "sdfx_conv3x3_packed_forward": [_ptr, _ptr, _ptr, _ptr, _u32, _u32, _u32, _u32, _u32, _u32, _int, _ptr, _ptr, _ptr],
    "sdfx_linear_scratch_bytes": [_u32, _u32, _u32, _int, _int],
    "sdfx_linear_forward": [_ptr, _ptr, _ptr, _ptr, _u32, _u32, _u32, _int, _int, _ptr, _ptr, _ptr],
    "sdfx_conv3x3_scratch_bytes": [_u32, _u32, _u32, _u32, _u32, _u32, _u32, _int, _int],
    "sdfx_conv3x3_forward": [_ptr, _ptr, _ptr, _ptr, _u32, _u32, _u32, _u32, _u32, _u32, _u32, _int, _int, _ptr, _ptr, _ptr],
    "sdfx_adan_ctl_words": [],
    "sdfx_amp_grad_stats_doubles": [],
    "sdfx_occupancy_stats_doubles": [],
    "sdfx_amp_grad_stats": [_ptr, _ptr, _ptr, _u32, _ptr, _ptr],
    "sdfx_adan_prepare": [_ptr, _ptr, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _u32, _ptr],
    "sdfx_adan_update": [_ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _u32, _ptr, _f32, _f32, _f32, _f32, _int, _ptr],
}
_RESTYPES = {
    "sdfx_march_rays_train_scratch_bytes": _u64,
    "sdfx_compact_rays_scratch_bytes": _u64,
    "sdfx_head_scratch_bytes": _u64,
    "sdfx_grid_encode_backward_binned_scratch_bytes": _u64,
    "sdfx_marching_tets_scratch_bytes": _u64,
    "sdfx_rasterize_scratch_bytes": _u64,
    "sdfx_field_packed_words": _u32,
    "sdfx_field_backward_scratch_bytes": _u64,
    "sdfx_group_norm_scratch_bytes": _u64,
    "sdfx_conv3x3_scratch_bytes": _u64,
    "sdfx_linear_scratch_bytes": _u64,
    "sdfx_conv3x3_packed_scratch_bytes": _u64,
    "sdfx_adan_ctl_words": _u32,
    "sdfx_amp_grad_stats_doubles": _u32,
    "sdfx_occupancy_stats_doubles": _u32,
}

_LIB = None


def lib() -> C.CDLL:
    """Open libsdfx_hip.so (once). Fails loudly when it has not been built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"libsdfx_hip.so not found at {LIB_PATH}: build it with "
                "`python stable-dreamfusion_amd/build.py` (or __graft_entry__.build()). "
                "There is no CPU or PyTorch fallback for these operators.")
        handle = C.CDLL(LIB_PATH)
        handle.sdfx_last_error.restype = C.c_char_p
        handle.sdfx_last_error.argtypes = []
        handle.sdfx_build_info.restype = C.c_char_p
        handle.sdfx_build_info.argtypes = []
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(handle, name, None)
            if fn is None:
                continue  # optional entry points are checked where they are used
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, _int)
        _LIB = handle
    return _LIB


class row_limit:
    """`with row_limit(total, period): ...` — sdfx_set_row_limit around the calls inside (include/sdfx.h): rows whose index modulo
    `period` is >= total[0] (an int32 device tensor) are padding for the hinted encoder forward, the field kernels and the binned
    table-gradient scatter. `total` None: no-op."""

    def __init__(self, total, period):
        self.total, self.period = total, int(period)

    def __enter__(self):
        if self.total is not None:
            check_tensor(self.total, "row limit", torch.int32)
            lib().sdfx_set_row_limit(ptr(self.total), self.period)
        return self

    def __exit__(self, *exc):
        if self.total is not None:
            lib().sdfx_set_row_limit(None, 0)
        return False


class stencil_source:
    """`with stencil_source(xyzs, epsilon, bound): ...` — sdfx_set_stencil_source around the calls inside (include/sdfx.h): the
    grid-encoder and field kernels form row r of the [7, M, 3] finite-difference stencil batch of `xyzs` [M, 3] themselves and
    ignore their `inputs` / `x` arguments (pass None). `xyzs` None: no-op."""

    def __init__(self, xyzs, epsilon, bound):
        self.xyzs, self.epsilon, self.bound = xyzs, float(epsilon), float(bound)

    def __enter__(self):
        if self.xyzs is not None:
            check_tensor(self.xyzs, "stencil source", torch.float32)
            lib().sdfx_set_stencil_source(ptr(self.xyzs), self.xyzs.shape[0], self.epsilon, self.bound, 2.0 * self.bound)
        return self

    def __exit__(self, *exc):
        if self.xyzs is not None:
            lib().sdfx_set_stencil_source(None, 0, 0.0, 0.0, 0.0)
        return False


class albedo_rows:
    """`with albedo_rows(n): ...` — sdfx_set_albedo_rows around the field calls inside (include/sdfx.h): their albedo / d-albedo buffers
    hold the first n rows of the batch only. n = 0 / None: no-op."""

    def __init__(self, rows):
        self.rows = int(rows or 0)

    def __enter__(self):
        if self.rows:
            lib().sdfx_set_albedo_rows(self.rows)
        return self

    def __exit__(self, *exc):
        if self.rows:
            lib().sdfx_set_albedo_rows(0)
        return False


DEV_LIB_PATH = os.path.join(_HERE, "csrc", "libsdfx_hip_dev.so")


def is_devtools() -> bool:
    """True when the loaded library is the devtools build (include/sdfx_devtools.h; SDFX_LIB=<...>/libsdfx_hip_dev.so)."""
    return hasattr(lib(), "sdfx_dev_set")


class dev_switch:
    """`with dev_switch(NAME=value, ...): ...` — sdfx_dev_set around the calls inside; the devtools library only."""

    def __init__(self, **values):
        if not is_devtools():
            raise RuntimeError("implementation switches exist only in libsdfx_hip_dev.so (SDFX_LIB=" + DEV_LIB_PATH + ")")
        self.values = values

    def __enter__(self):
        for k, v in self.values.items():
            lib().sdfx_dev_set(k.encode(), int(v))
        return self

    def __exit__(self, *exc):
        for k in self.values:
            lib().sdfx_dev_unset(k.encode())
        return False


def exported_symbols():
    """Names declared for binding (used by the CPU test that checks the ABI surface)."""
    return ["sdfx_last_error", "sdfx_build_info"] + list(_SIGNATURES)


def check(rc: int) -> None:
    if rc != 0:
        msg = lib().sdfx_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"sdfx error {rc}: {msg}")


def stream() -> C.c_void_p:
    """hipStream_t of torch's current stream on the current device."""
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t) -> C.c_void_p | None:
    return None if t is None else C.c_void_p(t.data_ptr())


# ---- argument checks: a superset of the reference's CHECK_CUDA / CHECK_CONTIGUOUS / CHECK_IS_* ----
def check_cuda(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")


def check_contiguous(t: torch.Tensor, name: str) -> None:
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")


def check_dtype(t: torch.Tensor, name: str, *dtypes) -> None:
    if t.dtype not in dtypes:
        raise RuntimeError(f"{name} must be one of {tuple(str(d) for d in dtypes)}, got {t.dtype}")


def check_tensor(t: torch.Tensor, name: str, *dtypes) -> torch.Tensor:
    check_cuda(t, name)
    check_contiguous(t, name)
    if dtypes:
        check_dtype(t, name, *dtypes)
    return t


def call(name: str, *args) -> None:
    check(getattr(lib(), name)(*args))


def half_image(p: torch.Tensor, create: bool = True):
    """The float16 image of a float32 parameter `p` that the forward kernels gather from — `p.to(torch.half)` of
    gridencoder/grid.py:46-47 — kept on the tensor object and reused while it is current: (buffer, p._version at the time it was
    formed). Every PyTorch in-place write to `p` (an optimiser's foreach ops, load_state_dict) bumps `p._version`, so the image is
    re-formed (into the SAME buffer: captured graphs keep reading it) at the next call; DeviceAdan writes parameters through a raw
    pointer, which does not touch the version — and rewrites the image in the same kernel (sdfx_adan_update: half_copies), so the
    two stay in step without a cast launch per iteration. `create` False: only return a current image (None otherwise)."""
    hit = getattr(p, "_sdfx_half", None)
    if hit is not None and hit[1] == p._version and hit[0].shape == p.shape and hit[0].device == p.device:
        return hit[0]
    if not create:
        return None
    buf = hit[0] if (hit is not None and hit[0].shape == p.shape and hit[0].device == p.device) else torch.empty_like(p, dtype=torch.float16)
    with torch.no_grad():
        buf.copy_(p.detach())
    p._sdfx_half = (buf, p._version)
    return buf


class StreamScratch:
    """Grow-on-demand float32 scratch per (device, stream), for kernels whose every call rewrites what it reads (split-K partials,
    GroupNorm partial moments). Streams get their own buffer because nothing orders two streams' calls against each other. A buffer
    that a HIP-graph capture was handed has its address baked into that graph and must outlive it: such a buffer is kept when it is
    outgrown; one that no capture ever saw is released then. A request the current buffer cannot hold DURING a capture is served
    from the capturing graph's own pool (it lives and dies with that graph) and is not remembered."""

    def __init__(self):
        self._bufs = {}    # (device index, stream handle) -> [[buffer, handed to a capture?], ...], the last one current

    def get(self, device, nbytes):
        key = (device.index, torch.cuda.current_stream(device).cuda_stream)
        bufs = self._bufs.setdefault(key, [])
        capturing = torch.cuda.is_current_stream_capturing()
        if not bufs or bufs[-1][0].numel() * 4 < nbytes:
            buf = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
            if capturing:
                return buf
            bufs[:] = [b for b in bufs if b[1]]      # superseded buffers no graph refers to: released
            bufs.append([buf, False])
        if capturing:
            bufs[-1][1] = True
        return bufs[-1][0]

    def held_bytes(self):
        return sum(b[0].numel() * 4 for bufs in self._bufs.values() for b in bufs)
