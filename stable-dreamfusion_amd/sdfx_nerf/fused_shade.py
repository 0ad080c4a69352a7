"""Fused normal / shading / orientation term (csrc/shade.hip) behind an autograd Function.

    color, normal, orient = fused_shade(sigma7, albedo0, dirs, rays, rays_o, light_offset, ratio, total, shading)

== the tail of NeRFNetwork.forward (nerf/network_grid.py:98-130) for shading in {'lambertian', 'textureless',
'normal'} after the seven densities of the finite-difference stencil are known, plus `dirs = safe_normalize(dirs)`
(renderer.py:734) and the per-sample factor `clamp(normal . dirs, 0)^2` of loss_orient (renderer.py:744-746).

  sigma7        [7 * cap] float32: densities at x, x+e_x, x-e_x, x+e_y, x-e_y, x+e_z, x-e_z (stencil-major)
  albedo0       [cap, 3] float32 albedo at x (only read for 'lambertian')
  dirs          [cap, 3] un-normalised view directions as the march wrote them
  rays          [N, 2] int32 (offset, count); rays_o [N, 3]; light_offset [3] (the reference's torch.randn(3))
  ratio         0-dim float32 tensor (ambient ratio); total int32 [1] = number of valid rows (rest is padding)
"""
from __future__ import annotations

import torch
from torch.amp import custom_bwd, custom_fwd
from torch.autograd import Function

import _render
import _sdfx as S

MODES = {"lambertian": 1, "textureless": 2, "normal": 3}
_F32 = torch.float32


class _fused_shade(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, sigma7, albedo0, dirs, rays, rays_o, light_offset, ratio, total, mode, epsilon):
        sigma7 = sigma7.contiguous()
        dirs = dirs.contiguous()
        cap = dirs.shape[0]
        assert sigma7.numel() == 7 * cap
        if mode == 1:
            albedo0 = albedo0.contiguous()
        rays_o = rays_o.contiguous().view(-1, 3)
        n_rays = rays.shape[0]
        dev = sigma7.device
        color = torch.empty(cap, 3, dtype=_F32, device=dev)
        normal = torch.empty(cap, 3, dtype=_F32, device=dev)
        orient = torch.empty(cap, dtype=_F32, device=dev)
        S.call("sdfx_shade_forward", S.ptr(S.check_tensor(sigma7, "sigma7", _F32)), S.ptr(albedo0 if mode == 1 else None),
               S.ptr(S.check_tensor(dirs, "dirs", _F32)), S.ptr(S.check_tensor(rays, "rays", torch.int32)),
               S.ptr(S.check_tensor(rays_o, "rays_o", _F32)), S.ptr(S.check_tensor(light_offset, "light_offset", _F32)),
               S.ptr(S.check_tensor(ratio, "ratio", _F32)), mode, float(epsilon), cap, n_rays,
               S.ptr(S.check_tensor(total, "total", torch.int32)), S.ptr(color), S.ptr(normal), S.ptr(orient), S.stream())
        ctx.save_for_backward(sigma7, albedo0 if mode == 1 else None, dirs, rays, rays_o, light_offset, ratio, total)
        ctx.meta = (mode, float(epsilon), cap, n_rays)
        ctx.set_materialize_grads(False)
        return color, normal, orient

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, dcolor, dnormal, dorient):
        sigma7, albedo0, dirs, rays, rays_o, light_offset, ratio, total = ctx.saved_tensors
        mode, epsilon, cap, n_rays = ctx.meta
        dev = sigma7.device
        dcolor = torch.zeros(cap, 3, dtype=_F32, device=dev) if dcolor is None else dcolor.float().contiguous()
        dorient = torch.zeros(cap, dtype=_F32, device=dev) if dorient is None else dorient.float().contiguous()
        dnormal = None if dnormal is None else dnormal.float().contiguous()
        dsigma7 = torch.empty(7 * cap, dtype=_F32, device=dev)
        dalbedo = torch.empty(cap, 3, dtype=_F32, device=dev) if mode == 1 else None
        S.call("sdfx_shade_backward", S.ptr(sigma7), S.ptr(albedo0), S.ptr(dirs), S.ptr(rays), S.ptr(rays_o), S.ptr(light_offset),
               S.ptr(ratio), mode, epsilon, cap, n_rays, S.ptr(total), S.ptr(dcolor), S.ptr(dnormal), S.ptr(dorient),
               S.ptr(dsigma7), S.ptr(dalbedo), S.stream())
        return dsigma7, dalbedo, None, None, None, None, None, None, None, None


def fused_shade(sigma7, albedo0, dirs, rays, rays_o, light_offset, ratio, total, shading, epsilon=1e-2):
    if not torch.is_tensor(ratio):
        ratio = torch.tensor(float(ratio), dtype=_F32, device=sigma7.device)
    return _fused_shade.apply(sigma7.reshape(-1), albedo0, dirs, rays, rays_o, light_offset, ratio.to(_F32), total,
                              MODES[shading], epsilon)


class _fused_render(Function):
    """Shading + compositing + regulariser sums (csrc/render.hip; include/sdfx.h sdfx_render_train_*).

        weights, weights_sum, depth, image, ray_sums = fused_render(sigma7, albedo0, dirs, ts, rays, rays_o, light_offset,
                                                                     ratio, mode, total, T_thresh)

    `mode`: a shading name, or a 0-dim float32 device tensor holding 1 / 2 / 3 (so that a captured graph serves every mode).
    `weights` [cap] is returned detached (its consumers — the entropy and orientation terms — are inside: ray_sums [N, 2])."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, sigma7, albedo0, dirs, ts, rays, rays_o, light_offset, ratio, mode, total, T_thresh, epsilon):
        sigma7, albedo0, dirs, ts = sigma7.contiguous(), albedo0.contiguous(), dirs.contiguous(), ts.contiguous()
        rays_o = rays_o.contiguous().view(-1, 3)
        cap, n_rays, dev = dirs.shape[0], rays.shape[0], sigma7.device
        assert sigma7.numel() == 7 * cap and albedo0.shape[0] == cap
        mode_dev = mode if torch.is_tensor(mode) else None
        mode_int = 0 if mode_dev is not None else MODES[mode]
        f = dict(dtype=_F32, device=dev)
        weights = torch.empty(cap, **f)
        weights_sum, depth = torch.empty(n_rays, **f), torch.empty(n_rays, **f)
        image, ray_sums = torch.empty(n_rays, 3, **f), torch.empty(n_rays, 2, **f)
        _render.train_forward(sigma7, albedo0, dirs, ts, rays, rays_o, light_offset, ratio, mode_dev, mode_int, epsilon, T_thresh, total,
                              weights, weights_sum, depth, image, ray_sums)
        ctx.save_for_backward(sigma7, albedo0, dirs, ts, rays, rays_o, light_offset, ratio, mode_dev, total, weights_sum, depth, image)
        ctx.meta = (mode_int, float(epsilon), float(T_thresh), cap, n_rays)
        ctx.mark_non_differentiable(weights)
        ctx.set_materialize_grads(False)
        return weights, weights_sum, depth, image, ray_sums

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, _gw, g_ws, g_depth, g_image, g_sums):
        sigma7, albedo0, dirs, ts, rays, rays_o, light_offset, ratio, mode_dev, total, weights_sum, depth, image = ctx.saved_tensors
        mode_int, epsilon, T_thresh, cap, n_rays = ctx.meta
        dev = sigma7.device
        z = lambda g, shape: torch.zeros(shape, dtype=_F32, device=dev) if g is None else g.float().contiguous()
        g_ws, g_image = z(g_ws, (n_rays,)), z(g_image, (n_rays, 3))
        g_depth = None if g_depth is None else g_depth.float().contiguous()
        g_sums = None if g_sums is None else g_sums.float().contiguous()
        dsigma7 = torch.empty(7 * cap, dtype=_F32, device=dev)
        dalbedo = torch.empty(cap, 3, dtype=_F32, device=dev)
        _render.train_backward(sigma7, albedo0, dirs, ts, rays, rays_o, light_offset, ratio, mode_dev, mode_int, epsilon, T_thresh, total,
                               weights_sum, depth, image, g_ws, g_depth, g_image, g_sums, dsigma7, dalbedo)
        return (dsigma7, dalbedo) + (None,) * 10


def fused_render(sigma7, albedo0, dirs, ts, rays, rays_o, light_offset, ratio, mode, total, T_thresh=1e-4, epsilon=1e-2):
    if not torch.is_tensor(ratio):
        ratio = torch.tensor(float(ratio), dtype=_F32, device=sigma7.device)
    return _fused_render.apply(sigma7.reshape(-1), albedo0, dirs, ts, rays, rays_o, light_offset, ratio.to(_F32), mode, total,
                               T_thresh, epsilon)


class _image_head(Function):
    """Background (network or colour) + `image + (1 - weights_sum) bg` + the [1, C, H, W] layout the guidance wants + the sum of
    the three per-ray regularisers, one kernel each way (csrc/head.hip; include/sdfx.h sdfx_head_*).

        pred, loss_reg = image_head(image_raw, weights_sum, ray_sums, rays_d, bg_net, bg_color, lam_entropy, n_valid,
                                    lam_opacity, lam_orient, C, H, W)

    bg_net: the model's 2-layer background MLP (sdfx_nerf.network_grid.MLP(39, 3, 32, 2)) or None -> bg_color [3] (device).
    lam_entropy, n_valid: 0-dim float32 device tensors (read at execution time: replay-safe)."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, image_raw, ws, ray_sums, rays_d, W1, b1, W2, b2, bg_color, lam_entropy, n_valid, lam_opacity, lam_orient, C, H, W):
        image_raw, ws = image_raw.contiguous().view(-1, 3), ws.contiguous().view(-1)
        N, dev = ws.shape[0], ws.device
        assert N == H * W
        net = None if W1 is None else tuple(t.contiguous() for t in (W1, b1, W2, b2))
        rays_d = None if rays_d is None else rays_d.contiguous().view(-1, 3)
        ray_sums = None if ray_sums is None else ray_sums.contiguous()
        bg_color = None if bg_color is None else bg_color.contiguous()
        pred = torch.empty(1, C, H, W, dtype=_F32, device=dev)
        loss_reg = torch.empty((), dtype=_F32, device=dev)
        _render.head_forward(image_raw, ws, ray_sums, rays_d, net, bg_color, lam_entropy, n_valid, lam_opacity, lam_orient, C, pred, loss_reg)
        ctx.save_for_backward(image_raw, ws, ray_sums, rays_d, W1 if net is None else net[0], None if net is None else net[1],
                              None if net is None else net[2], None if net is None else net[3], bg_color, lam_entropy, n_valid)
        ctx.meta = (float(lam_opacity), float(lam_orient), C, N)
        ctx.set_materialize_grads(False)
        return pred, loss_reg

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, g_pred, g_reg):
        image_raw, ws, ray_sums, rays_d, W1, b1, W2, b2, bg_color, lam_entropy, n_valid = ctx.saved_tensors
        lam_opacity, lam_orient, C, N = ctx.meta
        dev = ws.device
        f = dict(dtype=_F32, device=dev)
        g_pred = torch.zeros(1, C, N, **f) if g_pred is None else g_pred.float().contiguous()
        g_reg = None if g_reg is None else g_reg.float().contiguous()
        g_image, g_ws = torch.empty(N, 3, **f), torch.empty(N, **f)
        g_sums = None if ray_sums is None else torch.empty(N, 2, **f)
        net = None if W1 is None else (W1, b1, W2, b2)
        dnet = None if net is None else tuple(torch.empty_like(t) for t in net)
        _render.head_backward(image_raw, ws, ray_sums, rays_d, net, bg_color, lam_entropy, n_valid, lam_opacity, lam_orient, C, g_pred, g_reg,
                              g_image, g_ws, g_sums, dnet)
        dW1, db1, dW2, db2 = dnet if dnet is not None else (None,) * 4
        return (g_image, g_ws, g_sums, None, dW1, db1, dW2, db2) + (None,) * 8


def image_head(image_raw, ws, ray_sums, rays_d, bg_net, bg_color, lam_entropy, n_valid, lam_opacity, lam_orient, C, H, W):
    if bg_net is not None:
        n = bg_net.net
        W1, b1, W2, b2 = n[0].weight, n[0].bias, n[1].weight, n[1].bias
    else:
        W1 = b1 = W2 = b2 = None
    return _image_head.apply(image_raw, ws, ray_sums, rays_d, W1, b1, W2, b2, bg_color, lam_entropy, n_valid, lam_opacity, lam_orient,
                             C, H, W)


class _weights_entropy(Function):
    """sum_{i < total} H(clamp(w_i, 1e-5, 1 - 1e-5)) in bits — the un-normalised lambda_entropy term (nerf/utils.py:571-575)."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, weights, total):
        weights = weights.contiguous()
        out = torch.empty(1, dtype=torch.float64, device=weights.device)
        S.call("sdfx_entropy_forward", S.ptr(S.check_tensor(weights, "weights", _F32)), weights.numel(),
               S.ptr(S.check_tensor(total, "total", torch.int32)), S.ptr(out), S.stream())
        ctx.save_for_backward(weights, total)
        return out.to(_F32).reshape(())

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, g):
        weights, total = ctx.saved_tensors
        dw = torch.empty_like(weights)
        S.call("sdfx_entropy_backward", S.ptr(weights), weights.numel(), S.ptr(total), S.ptr(g.float().contiguous()), S.ptr(dw),
               S.stream())
        return dw, None


weights_entropy_sum = _weights_entropy.apply
