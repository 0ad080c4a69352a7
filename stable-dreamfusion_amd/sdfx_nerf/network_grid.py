"""NeRFNetwork — the `-O` field of nerf/network_grid.py: 16x2 hash grid -> MLP 32-64-64-4,
density blob, finite-difference normals, Lambertian shading, frequency-encoded background MLP."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.amp import custom_bwd, custom_fwd
from torch.autograd import Function

from freqencoder import FreqEncoder
from gridencoder import GridEncoder

from . import fused_field as _ff
from . import fused_shade as _fs
from .renderer import NeRFRenderer, safe_normalize
import _devswitch

# fused encode -> MLP -> activation kernels for the fp16-autocast path (SDFX_FUSED_FIELD=0 keeps the
# reference's module-by-module evaluation: GridEncoder -> nn.Linear stack -> torch activations)
_FUSED = _devswitch.get("SDFX_FUSED_FIELD", 1)
# evaluate the sample and its six finite-difference neighbours in ONE field call (the field is point-wise, so the
# values are those of the reference's seven separate common_forward calls, network_grid.py:81-96, 108-115)
_BATCH_STENCIL = _devswitch.get("SDFX_BATCH_STENCIL", 1)
_ROW_LIMIT = _devswitch.get("SDFX_ROW_LIMIT", 1)   # skip the padding rows of fixed-capacity buffers in the field kernels
_STENCIL_KERNEL = _devswitch.get("SDFX_STENCIL_KERNEL", 1)   # the [7, M, 3] stencil batch from one kernel (csrc/field.hip)
# normal / shading / orientation glue between the field and the compositor in one HIP kernel each way
_FUSED_SHADE = _devswitch.get("SDFX_FUSED_SHADE", 1)
# ... and the compositor and the entropy / orientation sums in the same kernel (csrc/render.hip)
_FUSED_RENDER = _devswitch.get("SDFX_FUSED_RENDER", 1)
# test-time frames: march + field + compositing + compaction of nerf/renderer.py:759-794 in one persistent kernel (csrc/infer.hip)
_FUSED_INFER = _devswitch.get("SDFX_FUSED_INFER", 1)
# albedo of a stencil batch stored / differentiated for its base samples only (csrc/field.hip: sdfx_set_albedo_rows)
_BASE_ALBEDO = _devswitch.get("SDFX_BASE_ALBEDO", 1)
# The background MLP (4096 rays x 1.4 k MACs) is evaluated in float32 even under autocast: its gradient is the image
# gradient times the loss scale, un-attenuated by compositing weights, and is what overflows fp16 first — in half it
# caps the loss scale ~64x lower (field gradients underflow) and costs a GradScaler skip every ~12 iterations.
# SDFX_BG_FP32=0 restores the reference's autocast behaviour (nn.Linear in half, nerf/network_grid.py:132-139).
_BG_FP32 = _devswitch.get("SDFX_BG_FP32", 1)


class _trunc_exp(Function):
    """exp with the backward argument clamped at 15 (activation.py:5-18)."""

    @staticmethod
    @custom_fwd(device_type="cuda", cast_inputs=torch.float)
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    @custom_bwd(device_type="cuda")
    def backward(ctx, g):
        x = ctx.saved_tensors[0]
        return g * torch.exp(x.clamp(max=15))


trunc_exp = _trunc_exp.apply


def biased_softplus(x, bias=0):
    return F.softplus(x - bias)


class MLP(nn.Module):
    """nerf/network_grid.py:13-32"""

    def __init__(self, dim_in, dim_out, dim_hidden, num_layers, bias=True):
        super().__init__()
        self.dim_in, self.dim_out, self.dim_hidden, self.num_layers = dim_in, dim_out, dim_hidden, num_layers
        self.net = nn.ModuleList([
            nn.Linear(dim_in if l == 0 else dim_hidden, dim_out if l == num_layers - 1 else dim_hidden, bias=bias)
            for l in range(num_layers)])

    def forward(self, x):
        for l in range(self.num_layers):
            x = self.net[l](x)
            if l != self.num_layers - 1:
                x = F.relu(x, inplace=True)
        return x


class NeRFNetwork(NeRFRenderer):
    def __init__(self, opt, num_layers=3, hidden_dim=64, num_layers_bg=2, hidden_dim_bg=32):
        super().__init__(opt)
        self.num_layers = num_layers
        self.hidden_dim = hidden_dim
        # get_encoder('hashgrid', input_dim=3, log2_hashmap_size=19, desired_resolution=2048 * bound,
        #             interpolation='smoothstep')  (network_grid.py:49, encoding.py:74-76)
        self.encoder = GridEncoder(input_dim=3, num_levels=16, level_dim=2, base_resolution=16, log2_hashmap_size=19,
                                   desired_resolution=2048 * self.bound, gridtype="hash", align_corners=False,
                                   interpolation="smoothstep")
        self.in_dim = self.encoder.output_dim
        self.sigma_net = MLP(self.in_dim, 4, hidden_dim, num_layers, bias=True)
        self.density_activation = trunc_exp if self.opt.density_activation == "exp" else biased_softplus

        if self.opt.bg_radius > 0:
            self.num_layers_bg = num_layers_bg
            self.hidden_dim_bg = hidden_dim_bg
            self.encoder_bg = FreqEncoder(input_dim=3, degree=6)  # get_encoder('frequency', multires=6)
            self.in_dim_bg = self.encoder_bg.output_dim
            self.bg_net = MLP(self.in_dim_bg, 3, hidden_dim_bg, num_layers_bg, bias=True)
        else:
            self.bg_net = None
        e = 1e-2  # finite-difference step of network_grid.py:81 (kept on the device: no per-call host tensor)
        self.register_buffer("_fd_offsets", torch.tensor([[e, 0, 0], [-e, 0, 0], [0, e, 0], [0, -e, 0], [0, 0, e], [0, 0, -e]],
                                                         dtype=torch.float32), persistent=False)

    accepts_curve_step = True    # density(x, curve_step=...): see common_forward

    def common_forward(self, x, slabs=1, ray_ordered=False, curve_step=0.0):
        """`slabs` = 7: x is the [7, N, 3] batch of a finite-difference stencil; `ray_ordered`: consecutive rows are
        consecutive samples of a ray; `curve_step` > 0: the rows walk a space-filling curve through a regular grid of points that far
        apart in world units (the occupancy refresh's Morton-ordered cell centres). All of them only steer the encoder's work split
        (include/sdfx.h, sdfx_grid_encode_forward_hint; same values either way)."""
        if _FUSED and _ff.supported(self.encoder, self.sigma_net, x, self.opt.density_activation, self.max_level):
            shape = x.shape[:-1]
            # sample spacing in the encoder's unit cube: dt_min = 2 sqrt(3) / max_steps (raymarching.cu:385) over 2 bound
            step = (3.0 ** 0.5) / (self.opt.max_steps * self.bound) if ray_ordered else -float(curve_step) / (2 * self.bound)
            sigma, albedo = _ff.fused_field(x.reshape(-1, 3), self.encoder, self.sigma_net, self.bound,
                                            self.opt.blob_density, self.opt.blob_radius, slabs, step)
            return sigma.view(*shape), albedo.view(*shape, 3)
        enc = self.encoder(x, bound=self.bound, max_level=self.max_level)
        h = self.sigma_net(enc)
        sigma = self.density_activation(h[..., 0] + self.density_blob(x))
        albedo = torch.sigmoid(h[..., 1:])
        return sigma, albedo

    def finite_difference_normal(self, x, epsilon=1e-2):
        """Six more field evaluations at x +- eps along each axis (network_grid.py:81-96)."""
        def sig(dx, dy, dz):
            off = torch.tensor([[dx, dy, dz]], device=x.device)
            return self.common_forward((x + off).clamp(-self.bound, self.bound))[0]
        dx_pos, dx_neg = sig(epsilon, 0.0, 0.0), sig(-epsilon, 0.0, 0.0)
        dy_pos, dy_neg = sig(0.0, epsilon, 0.0), sig(0.0, -epsilon, 0.0)
        dz_pos, dz_neg = sig(0.0, 0.0, epsilon), sig(0.0, 0.0, -epsilon)
        normal = torch.stack([0.5 * (dx_pos - dx_neg) / epsilon, 0.5 * (dy_pos - dy_neg) / epsilon,
                              0.5 * (dz_pos - dz_neg) / epsilon], dim=-1)
        return -normal

    def normal(self, x):
        normal = self.finite_difference_normal(x)
        normal = safe_normalize(normal)
        return torch.nan_to_num(normal)

    def _stencil_forward(self, x, epsilon=1e-2):
        """sigma, albedo at x and -grad(sigma) by central differences, from one batched field evaluation of
        [x, x+e_x, x-e_x, x+e_y, x-e_y, x+e_z, x-e_z] (offset points clamped to the box as the reference does)."""
        N = x.shape[0]
        e = epsilon
        offs = self._fd_offsets if e == 1e-2 else self._fd_offsets * (e / 1e-2)
        neigh = (x.unsqueeze(0) + offs.unsqueeze(1)).clamp(-self.bound, self.bound)      # [6, N, 3]
        pts = torch.cat([x.unsqueeze(0), neigh], dim=0).reshape(-1, 3)
        sigma_all, albedo_all = self.common_forward(pts, slabs=7, ray_ordered=self.training)
        s = sigma_all.view(7, N)
        normal = -torch.stack([0.5 * (s[1] - s[2]) / e, 0.5 * (s[3] - s[4]) / e, 0.5 * (s[5] - s[6]) / e], dim=-1)
        return s[0], albedo_all.view(7, N, 3)[0], normal

    def forward_fused(self, x, dirs, rays, rays_o, light_offset, total, ratio=1, shading="lambertian"):
        """forward() for shading != 'albedo' with the light direction given per RAY (rays_o + light_offset, normalised)
        instead of per sample, `dirs` un-normalised; also returns clamp(normal . dir, 0)^2 for loss_orient.
        Arithmetic of network_grid.py:98-130 on the batched 7-point field evaluation (csrc/shade.hip)."""
        N = x.shape[0]
        neigh = (x.unsqueeze(0) + self._fd_offsets.unsqueeze(1)).clamp(-self.bound, self.bound)
        pts = torch.cat([x.unsqueeze(0), neigh], dim=0).reshape(-1, 3)
        sigma_all, albedo_all = self.common_forward(pts, slabs=7, ray_ordered=True)
        color, normal, orient = _fs.fused_shade(sigma_all, albedo_all[:N] if shading == "lambertian" else None, dirs, rays,
                                                rays_o, light_offset, ratio, total, shading)
        return sigma_all[:N], color, normal, orient

    def forward_render(self, x, dirs, ts, rays, rays_o, light_offset, total, ratio=1, shading="lambertian", T_thresh=1e-4):
        """Field at the 7 stencil points -> ONE kernel for normal, shading, compositing and the two regulariser sums
        (csrc/render.hip). `shading`: a name, or a 0-dim device tensor holding 1 / 2 / 3 (lambertian / textureless / normal).
        Returns weights (detached), weights_sum, depth, image, ray_sums [N, 2] = per-ray (entropy sum, orientation sum)."""
        if _FUSED and _STENCIL_KERNEL and _ff.supported(self.encoder, self.sigma_net, x, self.opt.density_activation, self.max_level):
            step = (3.0 ** 0.5) / (self.opt.max_steps * self.bound)
            live = total if (_ROW_LIMIT and torch.is_tensor(total) and total.dtype == torch.int32 and total.is_cuda) else None
            sigma_all, albedo_all = _ff.fused_field(x.reshape(-1, 3), self.encoder, self.sigma_net, self.bound, self.opt.blob_density,
                                                    self.opt.blob_radius, 7, step, stencil_eps=1e-2, row_total=live, base_albedo=bool(_BASE_ALBEDO))
        else:
            neigh = (x.unsqueeze(0) + self._fd_offsets.unsqueeze(1)).clamp(-self.bound, self.bound)
            pts = torch.cat([x.unsqueeze(0), neigh], dim=0).reshape(-1, 3)
            sigma_all, albedo_all = self.common_forward(pts, slabs=7, ray_ordered=True)
        # (a slice — even the whole range — costs its backward a zero fill and a copy: only taken when the field returned all 7 slabs' albedo)
        albedo0 = albedo_all if albedo_all.shape[0] == x.shape[0] else albedo_all[:x.shape[0]]
        return _fs.fused_render(sigma_all, albedo0, dirs, ts, rays, rays_o, light_offset, ratio, shading, total, T_thresh)

    def infer_fused_available(self, shading, light_d=None):
        """The persistent inference kernel (csrc/infer.hip) covers 'albedo' shading of the -O field under fp16 autocast."""
        return bool(_FUSED_INFER and shading == "albedo" and self.density_bitfield.is_cuda and torch.is_autocast_enabled("cuda")
                    and self.opt.density_activation == "exp" and self.max_level is None and self.encoder.num_levels == 16
                    and self.encoder.level_dim == 2 and self.encoder.input_dim == 3 and self.sigma_net.num_layers == 3
                    and self.sigma_net.dim_hidden == 64)

    @torch.no_grad()
    def render_infer_fused(self, rays_o, rays_d, nears, fars, noises=None, T_thresh=1e-4, return_samples=False):
        """weights_sum [N], depth [N], image [N, 3] of the eval loop of nerf/renderer.py:759-794 from one kernel launch."""
        import numpy as np
        import _field
        import _gridencoder
        import _sdfx as S
        dev, N = rays_o.device, rays_o.shape[0]
        n = self.sigma_net.net
        packed = torch.empty(_field.packed_words(), dtype=torch.int32, device=dev)
        _field.pack(*[t.detach().float().contiguous() for t in (n[0].weight, n[0].bias, n[1].weight, n[1].bias, n[2].weight, n[2].bias)],
                    packed)
        emb = S.half_image(self.encoder.embeddings)          # autocast: fp16 table (grid.py:46-47), kept while current
        f = dict(dtype=torch.float32, device=dev)
        ws, depth, image = torch.empty(N, **f), torch.empty(N, **f), torch.empty(N, 3, **f)
        counter = torch.empty(1, dtype=torch.int32, device=dev)
        ns = torch.empty(N, dtype=torch.int32, device=dev) if return_samples else None
        chk = lambda t, name: S.check_tensor(t.contiguous(), name, torch.float32)
        S.call("sdfx_render_infer", S.ptr(chk(rays_o, "rays_o")), S.ptr(chk(rays_d, "rays_d")), S.ptr(chk(nears, "nears")),
               S.ptr(chk(fars, "fars")), S.ptr(None if noises is None else chk(noises, "noises")),
               S.ptr(S.check_tensor(self.density_bitfield, "density_bitfield", torch.uint8)), float(self.bound), 0,
               float(self.opt.dt_gamma), int(self.opt.max_steps), N, self.cascade, self.grid_size, S.ptr(emb),
               _gridencoder.offsets_host(self.encoder.offsets), 16, float(np.log2(self.encoder.per_level_scale)),
               int(self.encoder.base_resolution), self.encoder.gridtype_id, int(bool(self.encoder.align_corners)), self.encoder.interp_id,
               S.ptr(packed), float(self.opt.blob_density), float(self.opt.blob_radius), float(T_thresh), S.ptr(counter), S.ptr(ws),
               S.ptr(depth), S.ptr(image), S.ptr(ns), S.stream())
        return (ws, depth, image, ns) if return_samples else (ws, depth, image)

    def fused_render_available(self, shading):
        return bool(_FUSED_RENDER and _FUSED_SHADE and _BATCH_STENCIL and (torch.is_tensor(shading) or shading in _fs.MODES))

    def fused_shade_available(self, shading):
        return bool(_FUSED_SHADE and _BATCH_STENCIL and shading in _fs.MODES)

    def forward(self, x, d, l=None, ratio=1, shading="albedo"):
        if shading != "albedo" and _BATCH_STENCIL:
            sigma, albedo, normal = self._stencil_forward(x)
            normal = torch.nan_to_num(safe_normalize(normal))
        else:
            sigma, albedo = self.common_forward(x)
            normal = None if shading == "albedo" else self.normal(x)
        if shading == "albedo":
            color = albedo
        else:
            lambertian = ratio + (1 - ratio) * (normal * l).sum(-1).clamp(min=0)
            if shading == "textureless":
                color = lambertian.unsqueeze(-1).repeat(1, 3)
            elif shading == "normal":
                color = (normal + 1) / 2
            else:
                color = albedo * lambertian.unsqueeze(-1)
        return sigma, color, normal

    def density(self, x, curve_step=0.0):
        sigma, albedo = self.common_forward(x, curve_step=curve_step)
        return {"sigma": sigma, "albedo": albedo}

    def background(self, d):
        h = self.encoder_bg(d)
        if _BG_FP32:
            with torch.autocast("cuda", enabled=False):
                return torch.sigmoid(self.bg_net(h.float()))
        h = self.bg_net(h)
        return torch.sigmoid(h)

    def get_params(self, lr):
        params = [{"params": self.encoder.parameters(), "lr": lr * 10},
                  {"params": self.sigma_net.parameters(), "lr": lr}]
        if self.opt.bg_radius > 0:
            params.append({"params": self.bg_net.parameters(), "lr": lr})
        if getattr(self.opt, "dmtet", False) and not getattr(self.opt, "lock_geo", False):   # network_grid.py:168-170
            params.append({"params": self.sdf, "lr": lr})
            params.append({"params": self.deform, "lr": lr})
        return params
