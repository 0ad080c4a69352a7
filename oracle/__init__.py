"""CPU ORACLE — test infrastructure, never part of the shipped path.

numpy/ctypes front end of ``oracle/sdfx_oracle.c`` (a scalar C restatement of the
reference's four CUDA extensions) plus numpy restatements of the small pieces of Python
glue that define semantics (zero-initialisation, two-pass march protocol, field MLP).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package, and only as the checker.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    """Compile the C restatement (gcc, -ffp-contract=off). Returns the .so path."""
    so = os.path.join(_HERE, "libsdfx_oracle.so")
    src = os.path.join(_HERE, "sdfx_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libsdfx_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_grid_resolution.restype = C.c_uint32
        _LIB.orc_grid_resolution.argtypes = [C.c_uint32, C.c_float, C.c_uint32]
        _LIB.orc_half_to_float.restype = C.c_float
        _LIB.orc_half_to_float.argtypes = [C.c_uint16]
        _LIB.orc_float_to_half.restype = C.c_uint16
        _LIB.orc_float_to_half.argtypes = [C.c_float]
    return _LIB


def _p(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "oracle arrays must be C-contiguous"
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


u32, f32, i32 = C.c_uint32, C.c_float, C.c_int


# --------------------------------------------------------------------------- raymarching

def near_far_from_aabb(rays_o, rays_d, aabb, min_near=0.2):
    """raymarching/raymarching.py:31-61"""
    rays_o, rays_d, aabb = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3), _f32(aabb)
    N = rays_o.shape[0]
    nears = np.empty(N, np.float32)
    fars = np.empty(N, np.float32)
    lib().orc_near_far_from_aabb(_p(rays_o), _p(rays_d), _p(aabb), u32(N), f32(min_near), _p(nears), _p(fars))
    return nears, fars


def sph_from_ray(rays_o, rays_d, radius):
    """raymarching/raymarching.py:64-92"""
    rays_o, rays_d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    N = rays_o.shape[0]
    coords = np.empty((N, 2), np.float32)
    lib().orc_sph_from_ray(_p(rays_o), _p(rays_d), f32(radius), u32(N), _p(coords))
    return coords


def morton3D(coords):
    """raymarching/raymarching.py:95-116"""
    coords = _i32(coords)
    N = coords.shape[0]
    out = np.empty(N, np.int32)
    lib().orc_morton3D(_p(coords), u32(N), _p(out))
    return out


def morton3D_invert(indices):
    """raymarching/raymarching.py:118-138"""
    indices = _i32(indices)
    N = indices.shape[0]
    out = np.empty((N, 3), np.int32)
    lib().orc_morton3D_invert(_p(indices), u32(N), _p(out))
    return out


def packbits(grid, thresh, bitfield=None):
    """raymarching/raymarching.py:141-167 ; grid [C, H^3]"""
    grid = _f32(grid)
    N = grid.shape[0] * grid.shape[1] // 8
    if bitfield is None:
        bitfield = np.empty(N, np.uint8)
    lib().orc_packbits(_p(grid), u32(N), f32(thresh), _p(bitfield))
    return bitfield


def flatten_rays(rays, M):
    """raymarching/raymarching.py:170-191"""
    rays = _i32(rays)
    res = np.zeros(M, np.int32)
    lib().orc_flatten_rays(_p(rays), u32(rays.shape[0]), u32(M), _p(res))
    return res


def march_rays_train(rays_o, rays_d, bound, density_bitfield, C_, H, nears, fars, noises, dt_gamma=0.0,
                     max_steps=1024, contract=False):
    """raymarching/raymarching.py:197-258 — two-pass protocol; `noises` is passed in
    (the reference draws torch.rand inside) so that runs are reproducible."""
    rays_o, rays_d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    grid = np.ascontiguousarray(density_bitfield, dtype=np.uint8)
    nears, fars, noises = _f32(nears), _f32(fars), _f32(noises)
    N = rays_o.shape[0]
    counter = np.zeros(1, np.int32)
    rays = np.empty((N, 2), np.int32)
    args = (_p(rays_o), _p(rays_d), _p(grid), f32(bound), i32(int(contract)), f32(dt_gamma), u32(max_steps), u32(N),
            u32(C_), u32(H), _p(nears), _p(fars))
    lib().orc_march_rays_train(*args, None, None, None, _p(rays), _p(counter), _p(noises))
    M = int(counter[0])
    xyzs = np.zeros((M, 3), np.float32)
    dirs = np.zeros((M, 3), np.float32)
    ts = np.zeros((M, 2), np.float32)
    lib().orc_march_rays_train(*args, _p(xyzs), _p(dirs), _p(ts), _p(rays), _p(counter), _p(noises))
    return xyzs, dirs, ts, rays


def composite_rays_train_forward(sigmas, rgbs, ts, rays, T_thresh=1e-4, binarize=False):
    """raymarching/raymarching.py:261-295"""
    sigmas, rgbs, ts, rays = _f32(sigmas), _f32(rgbs), _f32(ts), _i32(rays)
    M, N = sigmas.shape[0], rays.shape[0]
    weights = np.zeros(M, np.float32)
    weights_sum = np.empty(N, np.float32)
    depth = np.empty(N, np.float32)
    image = np.empty((N, 3), np.float32)
    lib().orc_composite_rays_train_forward(_p(sigmas), _p(rgbs), _p(ts), _p(rays), u32(M), u32(N), f32(T_thresh),
                                           i32(int(binarize)), _p(weights), _p(weights_sum), _p(depth), _p(image))
    return weights, weights_sum, depth, image


def composite_rays_train_backward(grad_weights, grad_weights_sum, grad_depth, grad_image, sigmas, rgbs, ts, rays,
                                  weights_sum, depth, image, T_thresh=1e-4, binarize=False):
    """raymarching/raymarching.py:297-314"""
    sigmas, rgbs, ts, rays = _f32(sigmas), _f32(rgbs), _f32(ts), _i32(rays)
    M, N = sigmas.shape[0], rays.shape[0]
    grad_sigmas = np.zeros(M, np.float32)
    grad_rgbs = np.zeros((M, 3), np.float32)
    lib().orc_composite_rays_train_backward(_p(_f32(grad_weights)), _p(_f32(grad_weights_sum)), _p(_f32(grad_depth)),
                                            _p(_f32(grad_image)), _p(sigmas), _p(rgbs), _p(ts), _p(rays),
                                            _p(_f32(weights_sum)), _p(_f32(depth)), _p(_f32(image)), u32(M), u32(N),
                                            f32(T_thresh), i32(int(binarize)), _p(grad_sigmas), _p(grad_rgbs))
    return grad_sigmas, grad_rgbs


def march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C_, H, nears, fars, noises,
               dt_gamma=0.0, max_steps=1024, contract=False):
    """raymarching/raymarching.py:323-371"""
    rays_o, rays_d = _f32(rays_o).reshape(-1, 3), _f32(rays_d).reshape(-1, 3)
    grid = np.ascontiguousarray(density_bitfield, dtype=np.uint8)
    M = n_alive * n_step
    xyzs = np.zeros((M, 3), np.float32)
    dirs = np.zeros((M, 3), np.float32)
    ts = np.zeros((M, 2), np.float32)
    lib().orc_march_rays(u32(n_alive), u32(n_step), _p(_i32(rays_alive)), _p(_f32(rays_t)), _p(rays_o), _p(rays_d),
                         f32(bound), i32(int(contract)), f32(dt_gamma), u32(max_steps), u32(C_), u32(H), _p(grid),
                         _p(_f32(nears)), _p(_f32(fars)), _p(xyzs), _p(dirs), _p(ts), _p(_f32(noises)))
    return xyzs, dirs, ts


def composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, ts, weights_sum, depth, image, T_thresh=1e-2,
                   binarize=False):
    """raymarching/raymarching.py:374-398 — mutates rays_alive, rays_t, weights_sum, depth, image in place."""
    for a, dt in ((rays_alive, np.int32), (rays_t, np.float32), (weights_sum, np.float32), (depth, np.float32),
                  (image, np.float32)):
        assert a.dtype == dt and a.flags["C_CONTIGUOUS"]
    lib().orc_composite_rays(u32(n_alive), u32(n_step), f32(T_thresh), i32(int(binarize)), _p(rays_alive), _p(rays_t),
                             _p(_f32(sigmas)), _p(_f32(rgbs)), _p(_f32(ts)), _p(weights_sum), _p(depth), _p(image))


# --------------------------------------------------------------------------- gridencoder

def grid_offsets(input_dim=3, num_levels=16, level_dim=2, per_level_scale=2.0, base_resolution=16,
                 log2_hashmap_size=19, desired_resolution=None):
    """gridencoder/grid.py:104-138 — table layout (float64 host arithmetic, as the reference)."""
    if desired_resolution is not None:
        per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
    max_params = 2 ** log2_hashmap_size
    offsets, offset = [], 0
    for i in range(num_levels):
        resolution = int(np.ceil(base_resolution * per_level_scale ** i))
        params_in_level = min(max_params, resolution ** input_dim)
        params_in_level = int(np.ceil(params_in_level / 8) * 8)
        offsets.append(offset)
        offset += params_in_level
    offsets.append(offset)
    return np.array(offsets, dtype=np.int32), float(per_level_scale)


def grid_resolution(level, S, H):
    """gridencoder.cu:133"""
    return int(lib().orc_grid_resolution(u32(level), f32(S), u32(H)))


def _tab(a):
    a = np.ascontiguousarray(a)
    assert a.dtype in (np.float32, np.float16)
    return a, int(a.dtype == np.float16)


def grid_encode_forward(inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False,
                        gridtype=0, align_corners=False, interpolation=0, max_level=None):
    """gridencoder/grid.py:25-70 — returns (outputs [B, L*C], outputs_LBC, dy_dx)."""
    inputs = _f32(inputs)
    embeddings, is_half = _tab(embeddings)
    offsets = _i32(offsets)
    B, D = inputs.shape
    L = offsets.shape[0] - 1
    Cc = embeddings.shape[1]
    S = np.log2(per_level_scale)
    H = base_resolution
    import math
    max_level = L if max_level is None else max(min(int(math.ceil(max_level * L)), L), 1)
    outputs = np.zeros((L, B, Cc), embeddings.dtype)
    dy_dx = np.zeros((B, L * D * Cc), embeddings.dtype) if calc_grad_inputs else None
    lib().orc_grid_encode_forward(_p(inputs), _p(embeddings), _p(offsets), _p(outputs), u32(B), u32(D), u32(Cc), u32(L),
                                  u32(max_level), f32(S), u32(H), _p(dy_dx), u32(gridtype), i32(int(align_corners)),
                                  u32(interpolation), i32(is_half))
    return outputs.transpose(1, 0, 2).reshape(B, L * Cc).copy(), outputs, dy_dx


def grid_encode_backward(grad, inputs, embeddings, offsets, per_level_scale, base_resolution, dy_dx=None, gridtype=0,
                         align_corners=False, interpolation=0, max_level=None):
    """gridencoder/grid.py:72-96 — grad [B, L*C] → (grad_inputs | None, grad_embeddings)."""
    inputs = _f32(inputs)
    embeddings, is_half = _tab(embeddings)
    offsets = _i32(offsets)
    B, D = inputs.shape
    L = offsets.shape[0] - 1
    Cc = embeddings.shape[1]
    S = np.log2(per_level_scale)
    H = base_resolution
    import math
    max_level = L if max_level is None else max(min(int(math.ceil(max_level * L)), L), 1)
    grad = np.ascontiguousarray(np.asarray(grad, dtype=embeddings.dtype).reshape(B, L, Cc).transpose(1, 0, 2))
    grad_embeddings = np.zeros_like(embeddings)
    grad_inputs = np.zeros((B, D), embeddings.dtype) if dy_dx is not None else None
    lib().orc_grid_encode_backward(_p(grad), _p(inputs), _p(embeddings), _p(offsets), _p(grad_embeddings), u32(B), u32(D),
                                   u32(Cc), u32(L), u32(max_level), f32(S), u32(H), _p(dy_dx), _p(grad_inputs),
                                   u32(gridtype), i32(int(align_corners)), u32(interpolation), i32(is_half))
    return grad_inputs, grad_embeddings


def grad_total_variation(inputs, embeddings, grad, offsets, weight, per_level_scale, base_resolution, gridtype=0,
                         align_corners=False):
    """gridencoder/grid.py:172-193 — adds into `grad` in place (float32 tables)."""
    inputs, embeddings, offsets = _f32(inputs), _f32(embeddings), _i32(offsets)
    assert grad.dtype == np.float32 and grad.flags["C_CONTIGUOUS"]
    B, D = inputs.shape
    L = offsets.shape[0] - 1
    lib().orc_grad_total_variation(_p(inputs), _p(embeddings), _p(grad), _p(offsets), f32(weight), u32(B), u32(D),
                                   u32(embeddings.shape[1]), u32(L), f32(np.log2(per_level_scale)), u32(base_resolution),
                                   u32(gridtype), i32(int(align_corners)))


def grad_weight_decay(embeddings, grad, offsets, weight):
    """gridencoder/grid.py:195-206 — adds into `grad` in place."""
    embeddings, offsets = _f32(embeddings), _i32(offsets)
    assert grad.dtype == np.float32 and grad.flags["C_CONTIGUOUS"]
    lib().orc_grad_weight_decay(_p(embeddings), _p(grad), _p(offsets), f32(weight), u32(embeddings.shape[0]),
                                u32(embeddings.shape[1]), u32(offsets.shape[0] - 1))


# --------------------------------------------------------------------------- freq / SH

def freq_encode_forward(inputs, degree):
    """freqencoder/freq.py:15-35"""
    inputs = _f32(inputs)
    B, D = inputs.shape
    Cc = D + D * 2 * degree
    out = np.empty((B, Cc), np.float32)
    lib().orc_freq_encode_forward(_p(inputs), u32(B), u32(D), u32(degree), u32(Cc), _p(out))
    return out


def freq_encode_backward(grad, outputs, input_dim, degree):
    """freqencoder/freq.py:37-50"""
    grad, outputs = _f32(grad), _f32(outputs)
    B, Cc = outputs.shape
    gi = np.zeros((B, input_dim), np.float32)
    lib().orc_freq_encode_backward(_p(grad), _p(outputs), u32(B), u32(input_dim), u32(degree), u32(Cc), _p(gi))
    return gi


def sh_encode_forward(inputs, degree, calc_grad_inputs=False):
    """shencoder/sphere_harmonics.py:14-38"""
    inputs = _f32(inputs)
    B, D = inputs.shape
    out = np.empty((B, degree ** 2), np.float32)
    dy_dx = np.empty((B, D * degree ** 2), np.float32) if calc_grad_inputs else None
    lib().orc_sh_encode_forward(_p(inputs), _p(out), u32(B), u32(D), u32(degree), _p(dy_dx))
    return out, dy_dx


def sh_encode_backward(grad, inputs, degree, dy_dx):
    """shencoder/sphere_harmonics.py:40-55"""
    grad, inputs, dy_dx = _f32(grad), _f32(inputs), _f32(dy_dx)
    B, D = inputs.shape
    gi = np.zeros((B, D), np.float32)
    lib().orc_sh_encode_backward(_p(grad), _p(inputs), u32(B), u32(D), u32(degree), _p(dy_dx), _p(gi))
    return gi


# --------------------------------------------------------------------------- field (numpy)

def density_blob(x, blob_density=5.0, blob_radius=0.2):
    """nerf/renderer.py:338-349 ('exp' activation branch)."""
    d = (x.astype(np.float32) ** 2).sum(-1)
    return (blob_density * np.exp(-d / np.float32(2 * blob_radius ** 2))).astype(np.float32)


def mlp_forward(x, weights, biases):
    """nerf/network_grid.py:13-32 — Linear/ReLU stack, float32."""
    h = x.astype(np.float32)
    for i, (w, b) in enumerate(zip(weights, biases)):
        h = h @ w.T.astype(np.float32) + b.astype(np.float32)
        if i != len(weights) - 1:
            h = np.maximum(h, 0)
    return h


def field_forward(enc, x, weights, biases, blob_density=5.0, blob_radius=0.2):
    """nerf/network_grid.py:68-78 common_forward: sigma = exp(h0 + blob(x)), albedo = sigmoid(h1..3)."""
    h = mlp_forward(enc, weights, biases)
    sigma = np.exp(h[..., 0] + density_blob(x, blob_density, blob_radius))
    albedo = 1.0 / (1.0 + np.exp(-h[..., 1:]))
    return sigma.astype(np.float32), albedo.astype(np.float32)


# ---- glue between the field and the compositor (numpy float32 restatement; csrc/shade.hip) -----------------------------
_SHADE_MODES = ("lambertian", "textureless", "normal")
_F = np.float32


def _safe_normalize(x, eps=1e-20):
    """nerf/utils.py:109-110: x / sqrt(clamp(sum(x * x, -1), min=eps)); returns (y, q, s)."""
    q = (x * x).sum(-1, dtype=_F)
    s = np.sqrt(np.maximum(q, _F(eps)), dtype=_F)
    with np.errstate(invalid="ignore", divide="ignore"):
        return (x / s[..., None]).astype(_F), q, s


def _ray_ids(rays):
    return np.repeat(np.arange(rays.shape[0]), rays[:, 1].astype(np.int64))


def shade_forward(sigma7, albedo, dirs, rays, rays_o, light_offset, ratio, shading, epsilon=1e-2):
    """NeRFNetwork.forward for shading != 'albedo' once the seven stencil densities are known
    (nerf/network_grid.py:81-96 finite_difference_normal, :98-104 normal, :117-130 shading), with the light
    direction of nerf/renderer.py:727 (safe_normalize(rays_o + offset), gathered per sample :736-737), the
    view-direction normalisation of :734 and the per-sample factor of loss_orient (:744-746).
    sigma7 [7, M] = densities at x, x+e_x, x-e_x, x+e_y, x-e_y, x+e_z, x-e_z. Returns color [M,3], normal [M,3], orient [M]."""
    assert shading in _SHADE_MODES
    s7 = np.asarray(sigma7, _F)
    e = _F(epsilon)
    with np.errstate(invalid="ignore", over="ignore"):
        raw = -np.stack([_F(0.5) * (s7[1] - s7[2]) / e, _F(0.5) * (s7[3] - s7[4]) / e, _F(0.5) * (s7[5] - s7[6]) / e], -1)
        y, _, _ = _safe_normalize(raw)
    n = np.nan_to_num(y, nan=0.0, posinf=np.finfo(_F).max, neginf=np.finfo(_F).min).astype(_F)
    light = _safe_normalize(np.asarray(rays_o, _F) + np.asarray(light_offset, _F))[0][_ray_ids(rays)]
    d = _safe_normalize(np.asarray(dirs, _F))[0]
    ratio = _F(ratio)
    lambert = ratio + (_F(1) - ratio) * np.maximum((n * light).sum(-1, dtype=_F), _F(0))
    if shading == "textureless":
        color = np.repeat(lambert[:, None], 3, 1)
    elif shading == "normal":
        color = (n + _F(1)) / _F(2)
    else:
        color = np.asarray(albedo, _F) * lambert[:, None]
    orient = np.maximum((n * d).sum(-1, dtype=_F), _F(0)) ** 2
    return color.astype(_F), n, orient.astype(_F)


def shade_backward(sigma7, albedo, dirs, rays, rays_o, light_offset, ratio, shading, dcolor, dorient, epsilon=1e-2):
    """Gradient of shade_forward's (color, orient) w.r.t. sigma7 [7, M] (row 0 is zero) and albedo [M, 3], with
    torch.autograd's conventions for the pieces involved: clamp(min=0) passes the gradient where its input is >= 0,
    nan_to_num blocks it where it replaced a value, the normaliser's clamp(min=1e-20) passes it where |x|^2 >= 1e-20."""
    s7 = np.asarray(sigma7, _F)
    e = _F(epsilon)
    with np.errstate(invalid="ignore", over="ignore", divide="ignore"):
        raw = -np.stack([_F(0.5) * (s7[1] - s7[2]) / e, _F(0.5) * (s7[3] - s7[4]) / e, _F(0.5) * (s7[5] - s7[6]) / e], -1)
        y, q, s = _safe_normalize(raw)
        n = np.nan_to_num(y, nan=0.0, posinf=np.finfo(_F).max, neginf=np.finfo(_F).min).astype(_F)
        light = _safe_normalize(np.asarray(rays_o, _F) + np.asarray(light_offset, _F))[0][_ray_ids(rays)]
        d = _safe_normalize(np.asarray(dirs, _F))[0]
        ratio = _F(ratio)
        ndl, ndd = (n * light).sum(-1, dtype=_F), (n * d).sum(-1, dtype=_F)
        lambert = ratio + (_F(1) - ratio) * np.maximum(ndl, _F(0))
        gc = np.asarray(dcolor, _F)
        dn = np.zeros_like(n)
        dalbedo = np.zeros_like(n)
        if shading == "normal":
            dn += gc / _F(2)
            dlambert = np.zeros_like(ndl)
        elif shading == "textureless":
            dlambert = gc.sum(-1, dtype=_F)
        else:
            alb = np.asarray(albedo, _F)
            dlambert = (gc * alb).sum(-1, dtype=_F)
            dalbedo = gc * lambert[:, None]
        dn += np.where(ndl >= 0, dlambert * (_F(1) - ratio), _F(0))[:, None] * light
        dn += np.where(ndd > 0, np.asarray(dorient, _F) * _F(2) * ndd, _F(0))[:, None] * d
        dy = np.where(y == n, dn, _F(0))
        dr = dy / s[:, None]
        dot = (dy * raw).sum(-1, dtype=_F)
        dr = dr - np.where(q >= _F(1e-20), dot / (s * s * s), _F(0))[:, None] * raw
        h = _F(0.5) / e
        ds7 = np.zeros_like(s7)
        ds7[1], ds7[2] = -h * dr[:, 0], h * dr[:, 0]
        ds7[3], ds7[4] = -h * dr[:, 1], h * dr[:, 1]
        ds7[5], ds7[6] = -h * dr[:, 2], h * dr[:, 2]
    return ds7.astype(_F), dalbedo.astype(_F)


def weights_entropy(weights, total):
    """Sum over the first `total` weights of the binary entropy (bits) of clamp(w, 1e-5, 1 - 1e-5) and its gradient
    (Trainer.train_step's lambda_entropy term before the mean, nerf/utils.py:571-575)."""
    w = np.asarray(weights, _F)
    a = np.clip(w[:total], _F(1e-5), _F(1) - _F(1e-5))
    ent = (-a * np.log2(a) - (_F(1) - a) * np.log2(_F(1) - a)).astype(np.float64).sum()
    g = np.zeros_like(w)
    inside = (w[:total] >= _F(1e-5)) & (w[:total] <= _F(1) - _F(1e-5))
    g[:total] = np.where(inside, np.log2(_F(1) - a) - np.log2(a), _F(0))
    return float(ent), g
