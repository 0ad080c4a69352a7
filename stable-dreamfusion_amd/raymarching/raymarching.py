"""`raymarching` — occupancy-grid ray marching and volume compositing operators.

Python surface of the reference's raymarching/raymarching.py (same callables, positional
orders and defaults; citations per function), re-implemented over libsdfx_hip.so. The
autograd / autocast behaviour is the reference's: every op runs in float32 regardless of
autocast (`custom_fwd(cast_inputs=float32)`), only `composite_rays_train` is differentiable.
"""
from __future__ import annotations

import torch
from torch.autograd import Function
from torch.amp import custom_bwd, custom_fwd

import _raymarching as _backend

__all__ = ["near_far_from_aabb", "sph_from_ray", "morton3D", "morton3D_invert", "packbits", "flatten_rays",
           "march_rays_train", "composite_rays_train", "march_rays", "composite_rays", "compact_rays",
           "march_rays_train_count", "march_rays_train_write", "march_rays_train_stage_write"]


def _cuda(t):
    # the reference silently moves CPU inputs to the GPU (raymarching/raymarching.py:46-47)
    return t if t.is_cuda else t.cuda()


# ---------------------------------------------------------------------------------- utils

class _near_far_from_aabb(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, rays_o, rays_d, aabb, min_near=0.2):
        """Ray/AABB intersection times. rays_o/d [N,3], aabb [6] -> nears, fars [N]
        (raymarching/raymarching.py:31-61; note the 0.2 default the renderer relies on)."""
        rays_o = _cuda(rays_o).contiguous().view(-1, 3)
        rays_d = _cuda(rays_d).contiguous().view(-1, 3)
        aabb = _cuda(aabb).contiguous()
        N = rays_o.shape[0]
        nears = torch.empty(N, dtype=rays_o.dtype, device=rays_o.device)
        fars = torch.empty(N, dtype=rays_o.dtype, device=rays_o.device)
        _backend.near_far_from_aabb(rays_o, rays_d, aabb, N, min_near, nears, fars)
        return nears, fars


near_far_from_aabb = _near_far_from_aabb.apply


class _sph_from_ray(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, rays_o, rays_d, radius):
        """(theta, phi) in [-1,1] of the far hit with sphere(radius) (raymarching/raymarching.py:64-92)."""
        rays_o = _cuda(rays_o).contiguous().view(-1, 3)
        rays_d = _cuda(rays_d).contiguous().view(-1, 3)
        N = rays_o.shape[0]
        coords = torch.empty(N, 2, dtype=rays_o.dtype, device=rays_o.device)
        _backend.sph_from_ray(rays_o, rays_d, radius, N, coords)
        return coords


sph_from_ray = _sph_from_ray.apply


class _morton3D(Function):
    @staticmethod
    def forward(ctx, coords):
        """coords [N,3] int in [0,1024) -> Morton code [N] int32 (raymarching/raymarching.py:95-116)."""
        coords = _cuda(coords)
        N = coords.shape[0]
        indices = torch.empty(N, dtype=torch.int32, device=coords.device)
        _backend.morton3D(coords.int().contiguous(), N, indices)
        return indices


morton3D = _morton3D.apply


class _morton3D_invert(Function):
    @staticmethod
    def forward(ctx, indices):
        """Morton code [N] -> coords [N,3] int32 (raymarching/raymarching.py:118-138)."""
        indices = _cuda(indices)
        N = indices.shape[0]
        coords = torch.empty(N, 3, dtype=torch.int32, device=indices.device)
        _backend.morton3D_invert(indices.int().contiguous(), N, coords)
        return coords


morton3D_invert = _morton3D_invert.apply


class _packbits(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, grid, thresh, bitfield=None):
        """grid [C, H^3] float -> bitfield [C*H^3/8] uint8, bit i of byte n = grid[8n+i] > thresh
        (raymarching/raymarching.py:141-167)."""
        grid = _cuda(grid).contiguous()
        C = grid.shape[0]
        H3 = grid.shape[1]
        N = C * H3 // 8
        if bitfield is None:
            bitfield = torch.empty(N, dtype=torch.uint8, device=grid.device)
        _backend.packbits(grid, N, thresh, bitfield)
        return bitfield


packbits = _packbits.apply


class _flatten_rays(Function):
    @staticmethod
    def forward(ctx, rays, M):
        """rays [N,2] (offset, count) -> res [M] ray id per sample (raymarching/raymarching.py:170-191)."""
        rays = _cuda(rays).contiguous()
        N = rays.shape[0]
        res = torch.zeros(M, dtype=torch.int, device=rays.device)
        _backend.flatten_rays(rays, N, M, res)
        return res


flatten_rays = _flatten_rays.apply


# ------------------------------------------------------------------------------- training

class _march_rays_train(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, perturb=False, dt_gamma=0,
                max_steps=1024, contract=False, noises=None):
        """March rays through the occupancy bitfield (raymarching/raymarching.py:197-258).

        Returns xyzs [M,3], dirs [M,3], ts [M,2] = (t after the step, dt), rays [N,2] int32 =
        (offset, count). Offsets are the prefix sum of the counts in ray order. `noises`
        (extension, optional float32 [N] in [0,1)) replaces the internally drawn start jitter so
        that runs are reproducible; `perturb=False` uses zeros as the reference does.
        """
        rays_o = _cuda(rays_o).float().contiguous().view(-1, 3)
        rays_d = _cuda(rays_d).float().contiguous().view(-1, 3)
        density_bitfield = _cuda(density_bitfield).contiguous()
        nears = nears.contiguous()
        fars = fars.contiguous()
        N = rays_o.shape[0]
        device = rays_o.device

        step_counter = torch.zeros(1, dtype=torch.int32, device=device)
        if noises is not None:
            noises = _cuda(noises).float().contiguous()
        elif perturb:
            noises = torch.rand(N, dtype=rays_o.dtype, device=device)
        else:
            noises = torch.zeros(N, dtype=rays_o.dtype, device=device)

        # pass 1: counts, offsets, total — and the per-sample ray times in `scratch`
        rays = torch.empty(N, 2, dtype=torch.int32, device=device)
        scratch = torch.empty(N * max_steps, dtype=torch.float32, device=device)
        _backend.march_rays_train(rays_o, rays_d, density_bitfield, bound, contract, dt_gamma, max_steps, N, C, H, nears,
                                  fars, None, None, None, rays, step_counter, noises, scratch)

        M = step_counter.item()  # the one host sync of the renderer (output shapes depend on it)

        # pass 2 writes every element of the M samples, so no zero fill is needed here
        xyzs = torch.empty(M, 3, dtype=rays_o.dtype, device=device)
        dirs = torch.empty(M, 3, dtype=rays_o.dtype, device=device)
        ts = torch.empty(M, 2, dtype=rays_o.dtype, device=device)
        if M > 0:
            _backend.march_rays_train(rays_o, rays_d, density_bitfield, bound, contract, dt_gamma, max_steps, N, C, H,
                                      nears, fars, xyzs, dirs, ts, rays, step_counter, noises, scratch)
        return xyzs, dirs, ts, rays


march_rays_train = _march_rays_train.apply


# Extension — the two passes of march_rays_train as separate calls, so that a caller can put the one host
# synchronisation of an iteration (reading the sample total) wherever it wants and give the writing pass a
# CAPACITY instead of the exact total: output shapes then no longer depend on the scene, which is what a HIP-graph
# replay of the rest of the iteration needs. Same kernels, same values as march_rays_train.
@torch.no_grad()
def march_rays_train_count(rays_o, rays_d, bound, density_bitfield, C, H, nears, fars, perturb=False, dt_gamma=0,
                           max_steps=1024, contract=False, noises=None, state=None):
    """Counting pass. Returns a state dict {rays [N,2] i32 (offset, count), counter [1] i32 (device), noises,
    scratch, ...}; pass `state` back in to reuse its buffers (their addresses then stay fixed)."""
    rays_o = _cuda(rays_o).float().contiguous().view(-1, 3)
    rays_d = _cuda(rays_d).float().contiguous().view(-1, 3)
    N = rays_o.shape[0]
    device = rays_o.device
    if state is None or state["rays"].shape[0] != N or state["max_steps"] != max_steps:
        state = {"rays": torch.empty(N, 2, dtype=torch.int32, device=device),
                 "counter": torch.zeros(1, dtype=torch.int32, device=device),
                 "scratch": torch.empty(N * max_steps, dtype=torch.float32, device=device),
                 "noises": torch.zeros(N, dtype=torch.float32, device=device), "max_steps": max_steps}
    state["counter"].zero_()  # offsets are handed out starting from counter[0] (raymarching.cu:470-474)
    if noises is not None:
        state["noises"].copy_(_cuda(noises).float().view(-1))
    elif perturb:
        state["noises"].uniform_(0, 1)
    else:
        state["noises"].zero_()
    state.update(rays_o=rays_o, rays_d=rays_d, grid=_cuda(density_bitfield).contiguous(), nears=nears.contiguous(),
                 fars=fars.contiguous(), args=(bound, contract, dt_gamma, max_steps, N, C, H))
    bound, contract, dt_gamma, max_steps, N, C, H = state["args"]
    _backend.march_rays_train(rays_o, rays_d, state["grid"], bound, contract, dt_gamma, max_steps, N, C, H, state["nears"],
                              state["fars"], None, None, None, state["rays"], state["counter"], state["noises"],
                              state["scratch"])
    return state


@torch.no_grad()
def march_rays_train_write(state, capacity):
    """Writing pass into xyzs/dirs [capacity,3], ts [capacity,2]; rows past the sample total stay zero.
    `capacity` must be at least the total the counting pass found (the caller read it, or bounds it)."""
    device = state["rays"].device
    xyzs = torch.zeros(capacity, 3, dtype=torch.float32, device=device)
    dirs = torch.zeros(capacity, 3, dtype=torch.float32, device=device)
    ts = torch.zeros(capacity, 2, dtype=torch.float32, device=device)
    bound, contract, dt_gamma, max_steps, N, C, H = state["args"]
    if capacity > 0:
        _backend.march_rays_train(state["rays_o"], state["rays_d"], state["grid"], bound, contract, dt_gamma, max_steps, N,
                                  C, H, state["nears"], state["fars"], xyzs, dirs, ts, state["rays"], state["counter"],
                                  state["noises"], state["scratch"])
    return xyzs, dirs, ts, state["rays"]


def march_rays_train_stage_write(state, capacity, out_rays_o, out_rays_d, out_rays, out_total, out_n_valid):
    """march_rays_train_write for a replayed iteration: the same xyzs / dirs / ts (padding rows zero) from ONE launch that also
    copies the counting pass's rays, origins, directions and total into the iteration's own buffers (out_*), freeing the
    staging buffers. Needs the scratch the counting pass recorded the sample times in."""
    device = state["rays"].device
    f = dict(dtype=torch.float32, device=device)
    xyzs, dirs, ts = torch.empty(capacity, 3, **f), torch.empty(capacity, 3, **f), torch.empty(capacity, 2, **f)
    bound, contract, dt_gamma, max_steps, N, C, H = state["args"]
    _backend.march_rays_train_stage_write(state["rays_o"], state["rays_d"], bound, contract, dt_gamma, max_steps, N, C, H,
                                          state["rays"], state["counter"], state["scratch"], capacity, xyzs, dirs, ts,
                                          out_rays_o, out_rays_d, out_rays, out_total, out_n_valid)
    return xyzs, dirs, ts, out_rays


class _composite_rays_train(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, sigmas, rgbs, ts, rays, T_thresh=1e-4, binarize=False):
        """Front-to-back compositing with early termination (raymarching/raymarching.py:261-295).
        Returns weights [M], weights_sum [N], depth [N], image [N,3]."""
        sigmas = sigmas.float().contiguous()
        rgbs = rgbs.float().contiguous()
        ts = ts.contiguous()
        rays = rays.contiguous()
        M = sigmas.shape[0]
        N = rays.shape[0]
        weights = torch.zeros(M, dtype=sigmas.dtype, device=sigmas.device)
        weights_sum = torch.empty(N, dtype=sigmas.dtype, device=sigmas.device)
        depth = torch.empty(N, dtype=sigmas.dtype, device=sigmas.device)
        image = torch.empty(N, 3, dtype=sigmas.dtype, device=sigmas.device)
        _backend.composite_rays_train_forward(sigmas, rgbs, ts, rays, M, N, T_thresh, binarize, weights, weights_sum,
                                              depth, image)
        ctx.save_for_backward(sigmas, rgbs, ts, rays, weights_sum, depth, image)
        ctx.dims = [M, N, T_thresh, binarize]
        return weights, weights_sum, depth, image

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, grad_weights, grad_weights_sum, grad_depth, grad_image):
        """raymarching/raymarching.py:297-314"""
        grad_weights = grad_weights.contiguous()
        grad_weights_sum = grad_weights_sum.contiguous()
        grad_depth = grad_depth.contiguous()
        grad_image = grad_image.contiguous()
        sigmas, rgbs, ts, rays, weights_sum, depth, image = ctx.saved_tensors
        M, N, T_thresh, binarize = ctx.dims
        grad_sigmas = torch.zeros_like(sigmas)
        grad_rgbs = torch.zeros_like(rgbs)
        _backend.composite_rays_train_backward(grad_weights, grad_weights_sum, grad_depth, grad_image, sigmas, rgbs, ts,
                                               rays, weights_sum, depth, image, M, N, T_thresh, binarize, grad_sigmas,
                                               grad_rgbs)
        return grad_sigmas, grad_rgbs, None, None, None, None


composite_rays_train = _composite_rays_train.apply


# ------------------------------------------------------------------------------ inference

class _march_rays(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, bound, density_bitfield, C, H, near, far,
                perturb=False, dt_gamma=0, max_steps=1024, contract=False, noises=None):
        """March the alive rays by at most n_step samples (raymarching/raymarching.py:323-371).
        Returns xyzs/dirs [n_alive*n_step,3], ts [n_alive*n_step,2]; unwritten slots stay 0."""
        rays_o = _cuda(rays_o).float().contiguous().view(-1, 3)
        rays_d = _cuda(rays_d).float().contiguous().view(-1, 3)
        device = rays_o.device
        M = n_alive * n_step
        xyzs = torch.zeros(M, 3, dtype=rays_o.dtype, device=device)
        dirs = torch.zeros(M, 3, dtype=rays_o.dtype, device=device)
        ts = torch.zeros(M, 2, dtype=rays_o.dtype, device=device)
        if noises is not None:
            noises = _cuda(noises).float().contiguous()
        elif perturb:
            noises = torch.rand(n_alive, dtype=rays_o.dtype, device=device)
        else:
            noises = torch.zeros(n_alive, dtype=rays_o.dtype, device=device)
        _backend.march_rays(n_alive, n_step, rays_alive.contiguous(), rays_t, rays_o, rays_d, bound, contract, dt_gamma,
                            max_steps, C, H, density_bitfield.contiguous(), near.contiguous(), far.contiguous(), xyzs,
                            dirs, ts, noises)
        return xyzs, dirs, ts


march_rays = _march_rays.apply


class _composite_rays(Function):
    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, ts, weights_sum, depth, image, T_thresh=1e-2,
                binarize=False):
        """Accumulate n_step samples per alive ray in place; rays_alive[n] = -1 marks a finished
        ray, rays_t carries the ray time forward (raymarching/raymarching.py:374-398)."""
        sigmas = sigmas.float().contiguous()
        rgbs = rgbs.float().contiguous()
        _backend.composite_rays(n_alive, n_step, T_thresh, binarize, rays_alive, rays_t, sigmas, rgbs, ts.contiguous(),
                                weights_sum, depth, image)
        return tuple()


composite_rays = _composite_rays.apply


def compact_rays(rays_alive: torch.Tensor) -> torch.Tensor:
    """Stable compaction `rays_alive[rays_alive >= 0]` (nerf/renderer.py:791) as a ballot /
    prefix-sum kernel. One host read-back of the survivor count sizes the result."""
    rays_alive = rays_alive.contiguous()
    n = rays_alive.shape[0]
    out = torch.empty_like(rays_alive)
    count = torch.zeros(1, dtype=torch.int32, device=rays_alive.device)
    _backend.compact_rays(rays_alive, n, out, count)
    return out[: int(count.item())]
