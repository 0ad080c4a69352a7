"""NeRFRenderer — the `cuda_ray` renderer of nerf/renderer.py driven by the HIP operators.

Restates the control flow of `run_cuda` (nerf/renderer.py:709-816), `update_extra_state`
(:1102-1149), `density_blob` (:338-349) and `render` (:1154-1190, cuda_ray branch). Buffer names,
shapes and dtypes (`aabb_train`, `aabb_infer`, `density_grid`, `density_bitfield`) are the
reference's, so its checkpoints load.
"""
from __future__ import annotations

import math
import os

import torch
import torch.nn as nn

import raymarching
import _devswitch

# occupancy refresh through csrc/occupancy.hip (Morton-ordered sample points generated on the device, EMA + mean +
# packbits without a host read); SDFX_FUSED_OCCUPANCY=0 keeps the reference's tensor-by-tensor flow
_FUSED_OCC = _devswitch.get("SDFX_FUSED_OCCUPANCY", 1)


def safe_normalize(x, eps=1e-20):
    """nerf/utils.py:109-110"""
    return x / torch.sqrt(torch.clamp(torch.sum(x * x, -1, keepdim=True), min=eps))


class NeRFRenderer(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.bound = opt.bound
        self.cascade = 1 + math.ceil(math.log2(opt.bound))
        self.grid_size = 128
        self.max_level = None
        self.cuda_ray = opt.cuda_ray
        self.min_near = opt.min_near
        self.density_thresh = opt.density_thresh

        aabb_train = torch.FloatTensor([-opt.bound, -opt.bound, -opt.bound, opt.bound, opt.bound, opt.bound])
        self.register_buffer("aabb_train", aabb_train)
        self.register_buffer("aabb_infer", aabb_train.clone())

        density_grid = torch.zeros([self.cascade, self.grid_size ** 3])
        density_bitfield = torch.zeros(self.cascade * self.grid_size ** 3 // 8, dtype=torch.uint8)
        self.register_buffer("density_grid", density_grid)
        self.register_buffer("density_bitfield", density_bitfield)
        self._mean_density, self._mean_density_dev = 0, None
        self.iter_density = 0
        self._grid_coords = None  # (morton indices, cell-centre coords), built once per device (unfused refresh only)
        self._occ_buffers = None  # (sample points, [sum, count], mean) of the fused refresh
        self.dmtet = bool(getattr(opt, "dmtet", False))
        self.glctx = None
        if self.dmtet:
            self._init_dmtet()

    # --------------------------------------------------------------------------- DMTet fine-tune stage (BASELINE configs[4])
    def _init_dmtet(self):
        """nerf/renderer.py:291-311: tetrahedral grid, sdf / deform parameters. The grid comes from `tets/{N}_tets.npz` when that
        file exists (the reference's quartet grids; its 128 grid is a large blob missing from the checkout), else from this
        repository's generator at the matching size (Kuhn n = N / 2: 274 625 vertices for N = 128 against 277 410; dmtet.py)."""
        import numpy as np
        from . import dmtet as D
        N = int(self.opt.tet_grid_size)
        # `opt.tets_dir` if given, else ./tets (where the reference looks, nerf/renderer.py:296), else <repository>/tets
        here = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        dirs = [d for d in (getattr(self.opt, "tets_dir", None), "tets", os.path.join(here, "tets")) if d]
        path = next((os.path.join(d, f"{N}_tets.npz") for d in dirs if os.path.exists(os.path.join(d, f"{N}_tets.npz"))), None)
        if path is not None:
            tets, self.tet_grid_source = np.load(path), path
        else:
            tets = D.kuhn_tet_grid(max(N // 2, 1))
            self.tet_grid_source = f"kuhn_tet_grid({max(N // 2, 1)})"
            import warnings
            warnings.warn(f"DMTet: tets/{N}_tets.npz not found in {dirs}; using the generated Kuhn grid "
                          f"({tets['vertices'].shape[0]} vertices). `sdf` / `deform` then have that many rows: a checkpoint trained "
                          f"on the reference's grid (277 410 vertices for N = 128) will not load into it.", stacklevel=2)
        self.register_buffer("verts", -torch.tensor(tets["vertices"], dtype=torch.float32) * 2, persistent=False)   # covers [-1, 1]
        self.register_buffer("indices", torch.tensor(tets["indices"], dtype=torch.long), persistent=False)
        self.register_buffer("tet_scale", torch.tensor([1, 1, 1], dtype=torch.float32), persistent=False)
        self.dmtet_model = D.DMTet(None)
        self.sdf = nn.Parameter(torch.zeros_like(self.verts[..., 0]), requires_grad=True)
        self.deform = nn.Parameter(torch.zeros_like(self.verts), requires_grad=True)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # a DMTet checkpoint made on another tetrahedral grid: say which, instead of PyTorch's bare size-mismatch line
        if getattr(self, "dmtet", False):
            for name in ("sdf", "deform"):
                t = state_dict.get(prefix + name)
                if t is not None and t.shape[0] != getattr(self, name).shape[0]:
                    raise RuntimeError(f"DMTet checkpoint has {t.shape[0]} grid vertices ({prefix}{name}), this model's grid "
                                       f"({self.tet_grid_source}) has {getattr(self, name).shape[0]}: load it with the same "
                                       f"tets/{int(self.opt.tet_grid_size)}_tets.npz the checkpoint was trained on")
        return super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def run_dmtet(self, rays_o, rays_d, mvp, h, w, **kwargs):
        """One frame of the DMTet stage (what nerf/renderer.py:862-964 returns): sdfx_nerf/dmtet_stage.py."""
        from . import dmtet_stage
        return dmtet_stage.run_dmtet(self, rays_o, rays_d, mvp, h, w, **kwargs)

    def init_tet(self, mesh=None):
        """sdf / tet_scale from the density field (nerf/renderer.py:818-859): sdfx_nerf/dmtet_stage.py."""
        from . import dmtet_stage
        return dmtet_stage.init_tet(self, mesh)

    @torch.no_grad()
    def density_blob(self, x):
        d = (x ** 2).sum(-1)
        if self.opt.density_activation == "exp":
            return self.opt.blob_density * torch.exp(-d / (2 * self.opt.blob_radius ** 2))
        return self.opt.blob_density * (1 - torch.sqrt(d) / self.opt.blob_radius)

    def forward(self, x, d):
        raise NotImplementedError()

    def density(self, x):
        raise NotImplementedError()

    def reset_extra_state(self):
        self.density_grid.zero_()
        self.mean_density = 0
        self.iter_density = 0

    # --------------------------------------------------------------------------- run_cuda
    def run_cuda(self, rays_o, rays_d, light_d=None, ambient_ratio=1.0, shading="albedo", bg_color=None, perturb=False,
                 T_thresh=1e-4, binarize=False, marched=None, shading_dev=None, defer_head=False, **kwargs):
        """`marched` (extension): (xyzs, dirs, ts, rays, n_valid) from raymarching.march_rays_train_count/_write when
        the caller has already marched into fixed-capacity buffers; n_valid is the device-side sample total and
        replaces the buffer length wherever the reference averages over samples."""
        prefix = rays_o.shape[:-1]
        rays_o = rays_o.contiguous().view(-1, 3)
        rays_d = rays_d.contiguous().view(-1, 3)
        N = rays_o.shape[0]
        device = rays_o.device

        # NOTE: min_near is not forwarded (the reference does not either), so the op's 0.2 default applies
        if marched is None:
            nears, fars = raymarching.near_far_from_aabb(rays_o, rays_d, self.aabb_train if self.training else self.aabb_infer)

        # one kernel for normal + shading + orientation term when nothing else needs the intermediate tensors
        fused = (self.training and light_d is None and hasattr(self, "fused_shade_available")
                 and self.fused_shade_available(shading) and self.opt.lambda_3d_normal_smooth <= 0
                 and self.opt.lambda_2d_normal_smooth <= 0 and self.opt.lambda_normal <= 0)
        if light_d is None:
            light_offset = torch.randn(3, device=device)
            if not fused:
                light_d = safe_normalize(rays_o + light_offset)

        results = {}
        if self.training:
            n_valid = total = None
            if marched is not None:
                xyzs, dirs, ts, rays, n_valid, total = marched
            else:
                xyzs, dirs, ts, rays = raymarching.march_rays_train(rays_o, rays_d, self.bound, self.density_bitfield,
                                                                    self.cascade, self.grid_size, nears, fars, perturb,
                                                                    self.opt.dt_gamma, self.opt.max_steps)
            orient = None
            rendered = (fused and not binarize and hasattr(self, "fused_render_available") and self.fused_render_available(shading))
            if rendered:   # field -> [normal, shading, compositing, entropy and orientation sums] in one kernel
                if total is None:
                    total = (rays[-1, 0] + rays[-1, 1]).reshape(1).to(torch.int32)
                weights, weights_sum, depth, image, ray_sums = self.forward_render(
                    xyzs, dirs, ts, rays, rays_o, light_offset, total, ratio=ambient_ratio,
                    shading=shading if shading_dev is None else shading_dev, T_thresh=T_thresh)
                if defer_head:   # background, [1, C, H, W] layout and the regulariser sum are the caller's one kernel
                    results.update(deferred=True, image_raw=image, ray_sums=ray_sums, weights=weights, weights_sum=weights_sum,
                                   depth=depth, num_samples=xyzs.shape[0], num_valid=n_valid, num_total=total)
                    return results
                sums = ray_sums.sum(0)
                denom = n_valid if n_valid is not None else float(max(xyzs.shape[0], 1))
                results["entropy_sum"] = sums[0]          # sum over samples of H(clamp(w)); the trainer divides by the sample count
                if self.opt.lambda_orient > 0:
                    results["loss_orient"] = sums[1] / denom
                normals = None
            elif fused:
                if total is None:  # offsets are the exclusive prefix sum of the counts in ray order
                    total = (rays[-1, 0] + rays[-1, 1]).reshape(1).to(torch.int32)
                sigmas, rgbs, normals, orient = self.forward_fused(xyzs, dirs, rays, rays_o, light_offset, total,
                                                                   ratio=ambient_ratio, shading=shading)
            else:
                dirs = safe_normalize(dirs)
                if light_d.shape[0] > 1:
                    flatten_rays = raymarching.flatten_rays(rays, xyzs.shape[0]).long()
                    light_d = light_d[flatten_rays]
                sigmas, rgbs, normals = self(xyzs, dirs, light_d, ratio=ambient_ratio, shading=shading)
            if not rendered:
                weights, weights_sum, depth, image = raymarching.composite_rays_train(sigmas, rgbs, ts, rays, T_thresh, binarize)

            if self.opt.lambda_orient > 0 and normals is not None:
                if orient is None:
                    orient = (normals * dirs).sum(-1).clamp(min=0) ** 2
                loss_orient = weights.detach() * orient
                # padded rows have weight 0, so only the divisor differs from .mean()
                results["loss_orient"] = loss_orient.mean() if n_valid is None else loss_orient.sum() / n_valid
            if self.opt.lambda_3d_normal_smooth > 0 and normals is not None:
                normals_perturb = self.normal(xyzs + torch.randn_like(xyzs) * 1e-2)
                results["loss_normal_perturb"] = (normals - normals_perturb).abs().mean()
            if (self.opt.lambda_2d_normal_smooth > 0 or self.opt.lambda_normal > 0) and normals is not None:
                _, _, _, normal_image = raymarching.composite_rays_train(sigmas.detach(), (normals + 1) / 2, ts, rays,
                                                                         T_thresh, binarize)
                results["normal_image"] = normal_image
            results["weights"] = weights
            results["num_samples"] = xyzs.shape[0]
            results["num_valid"] = n_valid
            results["num_total"] = total
        elif (hasattr(self, "infer_fused_available") and self.infer_fused_available(shading) and not binarize):
            # one persistent kernel instead of the host-paced loop below (csrc/infer.hip): same per-ray operations
            noises = torch.rand(N, dtype=torch.float32, device=device) if perturb else None
            weights_sum, depth, image = self.render_infer_fused(rays_o, rays_d, nears, fars, noises, T_thresh)
        else:
            dtype = torch.float32
            weights_sum = torch.zeros(N, dtype=dtype, device=device)
            depth = torch.zeros(N, dtype=dtype, device=device)
            image = torch.zeros(N, 3, dtype=dtype, device=device)
            n_alive = N
            rays_alive = torch.arange(n_alive, dtype=torch.int32, device=device)
            rays_t = nears.clone()
            step = 0
            while step < self.opt.max_steps:
                n_alive = rays_alive.shape[0]
                if n_alive <= 0:
                    break
                n_step = max(min(N // n_alive, 8), 1)
                xyzs, dirs, ts = raymarching.march_rays(n_alive, n_step, rays_alive, rays_t, rays_o, rays_d, self.bound,
                                                        self.density_bitfield, self.cascade, self.grid_size, nears, fars,
                                                        perturb if step == 0 else False, self.opt.dt_gamma,
                                                        self.opt.max_steps)
                dirs = safe_normalize(dirs)
                sigmas, rgbs, normals = self(xyzs, dirs, light_d, ratio=ambient_ratio, shading=shading)
                raymarching.composite_rays(n_alive, n_step, rays_alive, rays_t, sigmas, rgbs, ts, weights_sum, depth, image,
                                           T_thresh, binarize)
                rays_alive = raymarching.compact_rays(rays_alive)  # the reference: rays_alive[rays_alive >= 0]
                step += n_step

        if bg_color is None:
            bg_color = self.background(rays_d) if self.opt.bg_radius > 0 else 1
        image = image + (1 - weights_sum).unsqueeze(-1) * bg_color
        results["image"] = image.view(*prefix, 3)
        results["depth"] = depth.view(*prefix)
        results["weights_sum"] = weights_sum.reshape(*prefix)
        return results

    # ----------------------------------------------------------------- update_extra_state
    @property
    def mean_density(self):
        """Mean of the valid density-grid cells after the last refresh. The fused refresh leaves it on the device (nothing in
        the training loop needs it on the host); reading it here synchronises."""
        if self._mean_density_dev is not None:
            self._mean_density = float(self._mean_density_dev.item())
            self._mean_density_dev = None
        return self._mean_density

    @mean_density.setter
    def mean_density(self, v):
        self._mean_density, self._mean_density_dev = v, None

    @torch.no_grad()
    def update_extra_state(self, decay=0.95, S=128, noise=None):
        """Refresh density_grid (EMA-max of jittered field samples, Morton order) and repack the occupancy bitfield; called
        every opt.update_extra_interval iterations (nerf/renderer.py:1102-1149).
        `noise` (extension, for tests): [cascade, H^3, 3] uniform numbers in the order of the reference's
        torch.rand_like(cas_xyzs) (meshgrid index), instead of drawing them."""
        device = self.aabb_train.device
        if _FUSED_OCC and device.type == "cuda":
            return self._update_extra_state_fused(decay, noise)
        tmp_grid = -torch.ones_like(self.density_grid)
        if self._grid_coords is None or self._grid_coords[0].device != device:
            ar = torch.arange(self.grid_size, dtype=torch.int32, device=device)
            xx, yy, zz = torch.meshgrid(ar, ar, ar, indexing="ij")
            coords = torch.cat([xx.reshape(-1, 1), yy.reshape(-1, 1), zz.reshape(-1, 1)], dim=-1)
            indices = raymarching.morton3D(coords).long()
            xyzs = 2 * coords.float() / (self.grid_size - 1) - 1
            self._grid_coords = (indices, xyzs)
        indices, xyzs = self._grid_coords

        for cas in range(self.cascade):
            bound = min(2 ** cas, self.bound)
            half_grid_size = bound / self.grid_size
            cas_xyzs = xyzs * (bound - half_grid_size)
            u = torch.rand_like(cas_xyzs) if noise is None else noise[cas].to(cas_xyzs)
            cas_xyzs = cas_xyzs + (u * 2 - 1) * half_grid_size
            sigmas = self.density(cas_xyzs)["sigma"].reshape(-1).detach()
            tmp_grid[cas, indices] = sigmas.to(tmp_grid.dtype)

        valid_mask = self.density_grid >= 0
        self.density_grid[valid_mask] = torch.maximum(self.density_grid[valid_mask] * decay, tmp_grid[valid_mask])
        self.mean_density = torch.mean(self.density_grid[valid_mask]).item()
        self.iter_density += 1

        density_thresh = min(self.mean_density, self.density_thresh)
        self.density_bitfield = raymarching.packbits(self.density_grid, density_thresh, self.density_bitfield)

    def _update_extra_state_fused(self, decay, noise):
        """The same refresh on csrc/occupancy.hip: per cascade [points in Morton order -> density -> EMA-max + sum/count],
        then the threshold min(mean, density_thresh) and the bit packing on the device. No coords / indices tensors, no
        scatter, no host read (the reference reads the mean with .item())."""
        import _sdfx as S
        device = self.density_grid.device
        H, n = self.grid_size, self.grid_size ** 3
        if self._occ_buffers is None or self._occ_buffers[0].device != device:
            self._occ_buffers = (torch.empty(n, 3, dtype=torch.float32, device=device),
                                 torch.zeros(int(S.lib().sdfx_occupancy_stats_doubles()), dtype=torch.float64, device=device),
                                 torch.zeros(1, dtype=torch.float32, device=device))
        pts, stats, mean = self._occ_buffers
        grid = S.check_tensor(self.density_grid, "density_grid", torch.float32)
        # the seed comes from torch's CPU generator: reproducible under torch.manual_seed, no device synchronisation
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        for cas in range(self.cascade):
            bound = min(2 ** cas, self.bound)
            nz = None if noise is None else S.check_tensor(noise[cas].to(device=device, dtype=torch.float32).contiguous(), "noise",
                                                           torch.float32)
            S.call("sdfx_occupancy_points", H, float(bound), S.ptr(nz), seed, cas, S.ptr(pts), S.stream())
            # (the points are the grid's cell centres in Morton order, 2 bound_c / H apart: told to a field that takes the hint)
            hint = {"curve_step": 2.0 * bound / H} if getattr(self, "accepts_curve_step", False) else {}
            sigmas = self.density(pts, **hint)["sigma"].reshape(-1).detach().float().contiguous()
            S.call("sdfx_occupancy_update", grid.data_ptr() + cas * n * 4, S.ptr(sigmas), n, float(decay), S.ptr(stats),
                   int(cas == 0), S.stream())
        S.call("sdfx_occupancy_pack", S.ptr(grid), self.cascade * n, S.ptr(stats), float(self.density_thresh),
               S.ptr(S.check_tensor(self.density_bitfield, "density_bitfield", torch.uint8)), S.ptr(mean), S.stream())
        self._mean_density_dev = mean
        self.iter_density += 1

    def render(self, rays_o, rays_d, mvp=None, h=None, w=None, staged=False, max_ray_batch=4096, **kwargs):
        if self.dmtet:                                   # nerf/renderer.py:1160-1161
            return self.run_dmtet(rays_o, rays_d, mvp, h, w, **kwargs)
        return self.run_cuda(rays_o, rays_d, **kwargs)
