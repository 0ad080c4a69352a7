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


# ---- image head (csrc/head.hip): background + mix + [1, C, H, W] layout + regulariser sum ----
_HEAD_SCRATCH = {}


def _head_scratch(device, N):
    # one buffer per (device, stream, N): the kernels of one stream are ordered, two trainers on different streams must not share it
    key = (device.index, torch.cuda.current_stream(device).cuda_stream, N)
    buf = _HEAD_SCRATCH.get(key)
    if buf is None:
        buf = _HEAD_SCRATCH[key] = torch.empty(int(S.lib().sdfx_head_scratch_bytes(N)), dtype=torch.uint8, device=device)
    return buf


def _opt(t, name):
    return None if t is None else _f(t, name)


def head_forward(image_raw, ws, ray_sums, rays_d, net, bg_color, lam_entropy, n_valid, lam_opacity, lam_orient, C_, pred, loss_reg):
    N = ws.shape[0]
    W1, b1, W2, b2 = net if net is not None else (None,) * 4
    S.call("sdfx_head_forward", _f(image_raw, "image_raw"), _f(ws, "weights_sum"), _opt(ray_sums, "ray_sums"), _opt(rays_d, "rays_d"),
           _opt(W1, "W1"), _opt(b1, "b1"), _opt(W2, "W2"), _opt(b2, "b2"), _opt(bg_color, "bg_color"), _f(lam_entropy, "lambda_entropy"),
           _f(n_valid, "n_valid"), float(lam_opacity), float(lam_orient), N, C_, _f(pred, "pred"), _f(loss_reg, "loss_reg"),
           S.ptr(_head_scratch(ws.device, N)), S.stream())


def head_backward(image_raw, ws, ray_sums, rays_d, net, bg_color, lam_entropy, n_valid, lam_opacity, lam_orient, C_, g_pred, g_reg, g_image,
                  g_ws, g_sums, dnet):
    N = ws.shape[0]
    W1, b1, W2, b2 = net if net is not None else (None,) * 4
    dW1, db1, dW2, db2 = dnet if dnet is not None else (None,) * 4
    S.call("sdfx_head_backward", _f(image_raw, "image_raw"), _f(ws, "weights_sum"), _opt(ray_sums, "ray_sums"), _opt(rays_d, "rays_d"),
           _opt(W1, "W1"), _opt(b1, "b1"), _opt(W2, "W2"), _opt(b2, "b2"), _opt(bg_color, "bg_color"), _f(lam_entropy, "lambda_entropy"),
           _f(n_valid, "n_valid"), float(lam_opacity), float(lam_orient), N, C_, _f(g_pred, "grad_pred"), _opt(g_reg, "grad_loss_reg"),
           _f(g_image, "grad_image"), _f(g_ws, "grad_weights_sum"), _opt(g_sums, "grad_ray_sums"), _opt(dW1, "dW1"), _opt(db1, "db1"),
           _opt(dW2, "dW2"), _opt(db2, "db2"), S.ptr(_head_scratch(ws.device, N)), S.stream())
