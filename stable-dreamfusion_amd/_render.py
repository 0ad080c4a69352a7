"""`_render` — tensor-level binding of the fused shading + compositing kernels (include/sdfx.h, sdfx_render_train_*)."""
from __future__ import annotations

import torch

import _sdfx as S

_F32 = torch.float32


def _f(t, name):
    return S.ptr(S.check_tensor(t, name, _F32))


def train_forward(sigma7, albedo, dirs, ts, rays, rays_o, light_offset, ratio, mode_dev, mode, epsilon, T_thresh, total, weights,
                  weights_sum, depth, image, ray_sums):
    cap, n_rays = dirs.shape[0], rays.shape[0]
    S.call("sdfx_render_train_forward", _f(sigma7, "sigma7"), _f(albedo, "albedo"), _f(dirs, "dirs"), _f(ts, "ts"),
           S.ptr(S.check_tensor(rays, "rays", torch.int32)), _f(rays_o, "rays_o"), _f(light_offset, "light_offset"), _f(ratio, "ratio"),
           None if mode_dev is None else _f(mode_dev, "mode"), int(mode), float(epsilon), float(T_thresh), cap, n_rays,
           S.ptr(S.check_tensor(total, "total", torch.int32)), _f(weights, "weights"), _f(weights_sum, "weights_sum"),
           _f(depth, "depth"), _f(image, "image"), _f(ray_sums, "ray_sums"), S.stream())


def train_backward(sigma7, albedo, dirs, ts, rays, rays_o, light_offset, ratio, mode_dev, mode, epsilon, T_thresh, total, weights_sum,
                   depth, image, g_weights_sum, g_depth, g_image, g_ray_sums, dsigma7, dalbedo):
    cap, n_rays = dirs.shape[0], rays.shape[0]
    S.call("sdfx_render_train_backward", _f(sigma7, "sigma7"), _f(albedo, "albedo"), _f(dirs, "dirs"), _f(ts, "ts"), S.ptr(rays),
           _f(rays_o, "rays_o"), _f(light_offset, "light_offset"), _f(ratio, "ratio"), None if mode_dev is None else _f(mode_dev, "mode"),
           int(mode), float(epsilon), float(T_thresh), cap, n_rays, S.ptr(total), _f(weights_sum, "weights_sum"), _f(depth, "depth"),
           _f(image, "image"), _f(g_weights_sum, "grad_weights_sum"), None if g_depth is None else _f(g_depth, "grad_depth"),
           _f(g_image, "grad_image"), None if g_ray_sums is None else _f(g_ray_sums, "grad_ray_sums"), _f(dsigma7, "dsigma7"),
           _f(dalbedo, "dalbedo"), S.stream())
