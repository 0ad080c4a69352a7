"""`_raymarching` — drop-in for the reference's pybind module of the same name
(raymarching/src/bindings.cpp:5-18; the reference's raymarching.py tries
`import _raymarching as _backend` first, raymarching/raymarching.py:18-21).

Same 11 free functions, same argument order, tensors by value, all outputs caller-allocated;
each call checks device / contiguity / dtype and forwards raw pointers to libsdfx_hip.so on
torch's current stream.
"""
from __future__ import annotations

import torch

import _sdfx as S

_F32 = torch.float32
_I32 = torch.int32


def _f(t, name):
    return S.check_tensor(t, name, _F32)


def near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars):
    S.call("sdfx_near_far_from_aabb", S.ptr(_f(rays_o, "rays_o")), S.ptr(_f(rays_d, "rays_d")), S.ptr(_f(aabb, "aabb")),
           N, float(min_near), S.ptr(_f(nears, "nears")), S.ptr(_f(fars, "fars")), S.stream())


def sph_from_ray(rays_o, rays_d, radius, N, coords):
    S.call("sdfx_sph_from_ray", S.ptr(_f(rays_o, "rays_o")), S.ptr(_f(rays_d, "rays_d")), float(radius), N,
           S.ptr(_f(coords, "coords")), S.stream())


def morton3D(coords, N, indices):
    S.call("sdfx_morton3D", S.ptr(S.check_tensor(coords, "coords", _I32)), N,
           S.ptr(S.check_tensor(indices, "indices", _I32)), S.stream())


def morton3D_invert(indices, N, coords):
    S.call("sdfx_morton3D_invert", S.ptr(S.check_tensor(indices, "indices", _I32)), N,
           S.ptr(S.check_tensor(coords, "coords", _I32)), S.stream())


def packbits(grid, N, density_thresh, bitfield):
    S.call("sdfx_packbits", S.ptr(_f(grid, "grid")), N, float(density_thresh),
           S.ptr(S.check_tensor(bitfield, "bitfield", torch.uint8)), S.stream())


def flatten_rays(rays, N, M, res):
    S.call("sdfx_flatten_rays", S.ptr(S.check_tensor(rays, "rays", _I32)), N, M,
           S.ptr(S.check_tensor(res, "res", _I32)), S.stream())


# scratch handed from the counting pass to the writing pass of the same `rays` tensor
_MARCH_SCRATCH = {}


def march_rays_train(rays_o, rays_d, grid, bound, contract, dt_gamma, max_steps, N, C, H, nears, fars, xyzs, dirs, ts,
                     rays, counter, noises, scratch=None):
    """Two-call protocol of raymarching/raymarching.py:240-254: xyzs/dirs/ts None = counting pass.

    `scratch` (extension, optional float32 [N * max_steps]) lets the writing pass reuse the
    sample times recorded by the counting pass. When the caller does not manage it, the
    counting pass allocates one and parks it until the writing pass for the same `rays`.
    """
    S.check_tensor(rays, "rays", _I32)
    key = (rays.data_ptr(), N, max_steps)
    if xyzs is None:
        if scratch is None:
            scratch = torch.empty(N * max_steps, dtype=_F32, device=rays_o.device)
            _MARCH_SCRATCH.clear()  # at most one parked buffer
            _MARCH_SCRATCH[key] = scratch
    elif scratch is None:
        scratch = _MARCH_SCRATCH.pop(key, None)
    S.call("sdfx_march_rays_train", S.ptr(_f(rays_o, "rays_o")), S.ptr(_f(rays_d, "rays_d")),
           S.ptr(S.check_tensor(grid, "grid", torch.uint8)), float(bound), int(bool(contract)), float(dt_gamma),
           max_steps, N, C, H, S.ptr(_f(nears, "nears")), S.ptr(_f(fars, "fars")),
           S.ptr(None if xyzs is None else _f(xyzs, "xyzs")), S.ptr(None if dirs is None else _f(dirs, "dirs")),
           S.ptr(None if ts is None else _f(ts, "ts")), S.ptr(rays), S.ptr(S.check_tensor(counter, "counter", _I32)),
           S.ptr(_f(noises, "noises")), S.ptr(scratch), S.stream())


def march_rays_train_stage_write(rays_o, rays_d, bound, contract, dt_gamma, max_steps, N, C, H, rays, counter, scratch, capacity,
                                 xyzs, dirs, ts, out_rays_o, out_rays_d, out_rays, out_total, out_n_valid):
    """Pass 2 into fixed-capacity buffers + the staging copies of the iteration, one launch (include/sdfx.h)."""
    i32 = lambda t, n: S.ptr(S.check_tensor(t, n, _I32))
    S.call("sdfx_march_rays_train_stage_write", S.ptr(_f(rays_o, "rays_o")), S.ptr(_f(rays_d, "rays_d")), float(bound),
           int(bool(contract)), float(dt_gamma), max_steps, N, C, H, i32(rays, "rays"), i32(counter, "counter"),
           S.ptr(_f(scratch, "scratch")), capacity, S.ptr(_f(xyzs, "xyzs")), S.ptr(_f(dirs, "dirs")), S.ptr(_f(ts, "ts")),
           S.ptr(_f(out_rays_o, "out_rays_o")), S.ptr(_f(out_rays_d, "out_rays_d")), i32(out_rays, "out_rays"),
           i32(out_total, "out_total"), S.ptr(_f(out_n_valid, "out_n_valid")), S.stream())


def composite_rays_train_forward(sigmas, rgbs, ts, rays, M, N, T_thresh, binarize, weights, weights_sum, depth, image):
    S.call("sdfx_composite_rays_train_forward", S.ptr(_f(sigmas, "sigmas")), S.ptr(_f(rgbs, "rgbs")), S.ptr(_f(ts, "ts")),
           S.ptr(S.check_tensor(rays, "rays", _I32)), M, N, float(T_thresh), int(bool(binarize)),
           S.ptr(_f(weights, "weights")), S.ptr(_f(weights_sum, "weights_sum")), S.ptr(_f(depth, "depth")),
           S.ptr(_f(image, "image")), S.stream())


def composite_rays_train_backward(grad_weights, grad_weights_sum, grad_depth, grad_image, sigmas, rgbs, ts, rays,
                                  weights_sum, depth, image, M, N, T_thresh, binarize, grad_sigmas, grad_rgbs):
    S.call("sdfx_composite_rays_train_backward", S.ptr(_f(grad_weights, "grad_weights")),
           S.ptr(_f(grad_weights_sum, "grad_weights_sum")), S.ptr(_f(grad_depth, "grad_depth")),
           S.ptr(_f(grad_image, "grad_image")), S.ptr(_f(sigmas, "sigmas")), S.ptr(_f(rgbs, "rgbs")), S.ptr(_f(ts, "ts")),
           S.ptr(S.check_tensor(rays, "rays", _I32)), S.ptr(_f(weights_sum, "weights_sum")), S.ptr(_f(depth, "depth")),
           S.ptr(_f(image, "image")), M, N, float(T_thresh), int(bool(binarize)), S.ptr(_f(grad_sigmas, "grad_sigmas")),
           S.ptr(_f(grad_rgbs, "grad_rgbs")), S.stream())


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, contract, dt_gamma, max_steps, C, H, grid,
               nears, fars, xyzs, dirs, ts, noises):
    S.call("sdfx_march_rays", n_alive, n_step, S.ptr(S.check_tensor(rays_alive, "rays_alive", _I32)),
           S.ptr(_f(rays_t, "rays_t")), S.ptr(_f(rays_o, "rays_o")), S.ptr(_f(rays_d, "rays_d")), float(bound),
           int(bool(contract)), float(dt_gamma), max_steps, C, H, S.ptr(S.check_tensor(grid, "grid", torch.uint8)),
           S.ptr(_f(nears, "nears")), S.ptr(_f(fars, "fars")), S.ptr(_f(xyzs, "xyzs")), S.ptr(_f(dirs, "dirs")),
           S.ptr(_f(ts, "ts")), S.ptr(_f(noises, "noises")), S.stream())


def composite_rays(n_alive, n_step, T_thresh, binarize, rays_alive, rays_t, sigmas, rgbs, ts, weights_sum, depth, image):
    S.call("sdfx_composite_rays", n_alive, n_step, float(T_thresh), int(bool(binarize)),
           S.ptr(S.check_tensor(rays_alive, "rays_alive", _I32)), S.ptr(_f(rays_t, "rays_t")), S.ptr(_f(sigmas, "sigmas")),
           S.ptr(_f(rgbs, "rgbs")), S.ptr(_f(ts, "ts")), S.ptr(_f(weights_sum, "weights_sum")), S.ptr(_f(depth, "depth")),
           S.ptr(_f(image, "image")), S.stream())


# ---- extension: stable compaction of the alive list (nerf/renderer.py:791 does it with a mask) ----
def compact_rays(rays_alive, n, out, count, scratch=None):
    if scratch is None:
        nbytes = int(S.lib().sdfx_compact_rays_scratch_bytes(n))
        scratch = torch.empty(max(nbytes, 4), dtype=torch.uint8, device=rays_alive.device)
    S.call("sdfx_compact_rays", S.ptr(S.check_tensor(rays_alive, "rays_alive", _I32)), n,
           S.ptr(S.check_tensor(out, "out", _I32)), S.ptr(S.check_tensor(count, "count", _I32)), S.ptr(scratch),
           S.stream())
