"""The DMTet fine-tune stage's renderer (BASELINE configs[4]): `NeRFRenderer.run_dmtet` / `.init_tet` and the two mesh
regularisers, over this package's operators — marching tetrahedra (csrc/dmtet.hip) and rasterize / interpolate / antialias
(csrc/raster.hip, sdfx_nerf/dmtet.py).

What it computes is what nerf/renderer.py:862-964 (frame), :818-859 (sdf initialisation) and :179-257 (regularisers) compute; the
outputs are pinned to the reference's own (tests/golden/dmtet_ref.npz, tests/test_dmtet_golden.py). How it computes it is this
package's: a frame is a pipeline of four stages (mesh -> raster -> surface shading -> composition), the regularisers work on scalar
edge keys (one 1-D `unique` instead of row-wise uniques, an index_add instead of a sparse matrix product).
"""
from __future__ import annotations

import torch

from . import dmtet as D


def _unit(v, eps=1e-20):
    """v / max(|v|, sqrt(eps)) (nerf/utils.py:109-110)."""
    return v / torch.sqrt(torch.clamp((v * v).sum(-1, keepdim=True), min=eps))


# ------------------------------------------------------------------------------------------------ mesh regularisers
def _half_edges(faces):
    """The 3 F directed edges (a -> b) of the triangles, edge k of face f at row 3 f + k: (v0 v1), (v1 v2), (v2 v0)."""
    f = faces.long()
    return f.reshape(-1), f[:, [1, 2, 0]].reshape(-1)


def normal_consistency(face_normals, faces):
    """mean over the mesh's undirected edges of |1 - clamp(n_left . n_right, -1, 1)|, n_left / n_right the normals of the face that
    runs the edge from its lower to its higher vertex index and of the face that runs it the other way (nerf/renderer.py:179-215,
    236-245). An edge without one of the two (a boundary) pairs with face 0, as there."""
    with torch.autocast("cuda", enabled=False), torch.no_grad():
        a, b = _half_edges(faces)
        n_vert = int(faces.max()) + 1 if faces.numel() else 1
        key = torch.minimum(a, b) * n_vert + torch.maximum(a, b)
        _, edge_of = torch.unique(key, return_inverse=True)
        n_edges = int(edge_of.max()) + 1 if edge_of.numel() else 0
        face_of = torch.arange(faces.shape[0], device=faces.device).repeat_interleave(3)
        ascending = a <= b
        left = torch.zeros(n_edges, dtype=torch.long, device=faces.device)
        right = torch.zeros(n_edges, dtype=torch.long, device=faces.device)
        left[edge_of[ascending]] = face_of[ascending]
        right[edge_of[~ascending]] = face_of[~ascending]
    with torch.autocast("cuda", enabled=False):
        cos = (face_normals[left] * face_normals[right]).sum(-1, keepdim=True).clamp(-1.0, 1.0)
        return (1.0 - cos).abs().mean()


def laplacian_smooth_loss(verts, faces):
    """mean_i | deg(i) v_i - sum_{j ~ i} v_j | with the uniform graph Laplacian of the mesh, every neighbour counted once
    (nerf/renderer.py:217-234, 247-257)."""
    with torch.autocast("cuda", enabled=False):
        v = verts.float()
        n_vert = v.shape[0]
        with torch.no_grad():
            a, b = _half_edges(faces)
            pair = torch.unique(torch.cat([a * n_vert + b, b * n_vert + a]))      # directed neighbour pairs, each once
            i, j = pair // n_vert, pair % n_vert
            degree = torch.zeros(n_vert, dtype=v.dtype, device=v.device).index_add_(0, i, torch.ones_like(i, dtype=v.dtype))
        lap = degree[:, None] * v - torch.zeros_like(v).index_add_(0, i, v[j])
        return lap.norm(dim=1).mean()


# ------------------------------------------------------------------------------------------------ one frame
def _mesh(r):
    """Marching tetrahedra of the deformed grid -> (verts [V, 3], faces int64 [F, 3], sdf, deform) (nerf/renderer.py:875-879)."""
    deform = torch.tanh(r.deform) / r.opt.tet_grid_size
    verts, faces = r.dmtet_model(r.verts + deform, r.sdf, r.indices)
    return verts, faces, r.sdf, deform


def _normals(verts, faces):
    """Unit face normals [F, 3] and area-free vertex normals [V, 3] (sum of the unit normals of the incident faces; a vertex whose
    sum vanishes gets +z, nerf/renderer.py:881-897)."""
    corner = [verts[faces[:, k]] for k in range(3)]
    fn = _unit(torch.cross(corner[1] - corner[0], corner[2] - corner[0], dim=-1))
    vn = torch.zeros_like(verts)
    for k in range(3):
        vn = vn.index_add(0, faces[:, k], fn)
    keep = (vn * vn).sum(-1, keepdim=True) > 1e-20
    return fn, torch.where(keep, vn, vn.new_tensor([0.0, 0.0, 1.0]))


def _empty_frame(r, rays_d, mvp, h, w, bg_color, sdf, deform):
    """Marching tetrahedra found no surface (every sdf of one sign, e.g. before init_tet): the frame is the background, the opacity
    zero, and `sdf` / `deform` get zero gradients."""
    zero = (sdf.sum() + deform.sum()) * 0
    B, dev = mvp.shape[0], sdf.device
    bg = _background(r, rays_d, bg_color, h, w)
    out = {"depth": torch.zeros(B, h, w, 1, device=dev) + zero, "image": torch.zeros(B, h, w, 3, device=dev) + bg + zero,
           "weights_sum": torch.zeros(B, h, w, device=dev) + zero}
    if r.opt.lambda_2d_normal_smooth > 0 or r.opt.lambda_normal > 0:
        out["normal_image"] = torch.zeros(B, h, w, 3, device=dev) + zero
    if r.training:
        if getattr(r.opt, "lambda_mesh_normal", 0) > 0:
            out["normal_loss"] = zero
        if getattr(r.opt, "lambda_mesh_laplacian", 0) > 0:
            out["lap_loss"] = zero
    return out


def _background(r, rays_d, bg_color, h, w):
    if bg_color is None:
        bg_color = r.background(rays_d) if r.opt.bg_radius > 0 else 1
    if torch.is_tensor(bg_color) and bg_color.dim() > 1:
        bg_color = bg_color.view(-1, h, w, 3)
    return bg_color


def _surface_color(r, shading, albedo, normal, light_d, ambient_ratio):
    if getattr(r.opt, "lock_geo", False) and shading in ("textureless", "normal"):
        shading = "lambertian"                                                   # nerf/renderer.py:917-918
    if shading == "albedo":
        return albedo
    if shading == "normal":
        return (normal + 1) / 2
    lambert = ambient_ratio + (1 - ambient_ratio) * (normal * light_d).sum(-1).float().clamp(min=0)
    if shading == "textureless":
        return lambert.unsqueeze(-1).repeat(1, 1, 1, 3)
    return albedo * lambert.unsqueeze(-1)


def run_dmtet(r, rays_o, rays_d, mvp, h, w, light_d=None, ambient_ratio=1.0, shading="albedo", bg_color=None, **kwargs):
    """One frame of the DMTet stage: results {image [B, h, w, 3], depth [B, h, w, 1], weights_sum [B, h, w]} (+ normal_image,
    normal_loss, lap_loss when the options ask for them) — the dictionary `NeRFRenderer.render` returns for `opt.dmtet`."""
    eye = rays_o[:, 0, :]                                                         # one camera position per batch entry
    if light_d is None:
        light_d = _unit(eye + torch.randn_like(eye)).view(-1, 1, 1, 3)
    verts, faces, sdf, deform = _mesh(r)
    if faces.shape[0] == 0:
        return _empty_frame(r, rays_d, mvp, h, w, bg_color, sdf, deform)
    face_n, vert_n = _normals(verts, faces)
    tri = faces.int()

    # raster: clip-space vertices [B, V, 4] = [verts, 1] mvp^T, coverage + barycentrics, then position and normal per pixel
    homog = torch.cat([verts, torch.ones_like(verts[:, :1])], dim=-1)
    clip = torch.bmm(homog.unsqueeze(0).expand(mvp.shape[0], -1, -1), mvp.transpose(1, 2)).float()
    rast, _ = D.rasterize(r.glctx, clip, tri, (h, w))
    covered = rast[..., 3:] > 0
    pos, _ = D.interpolate(verts.unsqueeze(0), rast, tri)
    normal, _ = D.interpolate(vert_n.unsqueeze(0).contiguous(), rast, tri)
    normal = _unit(normal)

    # surface shading: the field's albedo at the covered pixels only
    flat, hit = pos.view(-1, 3), covered.view(-1).detach()
    albedo = torch.zeros_like(flat, dtype=torch.float32)
    if hit.any():
        albedo[hit] = r.density(flat[hit])["albedo"].float()
    color = _surface_color(r, shading, albedo.view(-1, h, w, 3), normal, light_d, ambient_ratio)

    # composition: antialiased colour and coverage over the background
    color = D.antialias(color, rast, clip, tri).clamp(0, 1)
    alpha = D.antialias(covered.float(), rast, clip, tri).clamp(0, 1)
    out = {"depth": rast[:, :, :, [2]], "image": color + (1 - alpha) * _background(r, rays_d, bg_color, h, w),
           "weights_sum": alpha.squeeze(-1)}
    if r.opt.lambda_2d_normal_smooth > 0 or r.opt.lambda_normal > 0:
        out["normal_image"] = D.antialias((normal + 1) / 2, rast, clip, tri).clamp(0, 1)
    if r.training:
        if getattr(r.opt, "lambda_mesh_normal", 0) > 0:
            out["normal_loss"] = normal_consistency(face_n, tri)
        if getattr(r.opt, "lambda_mesh_laplacian", 0) > 0:
            out["lap_loss"] = laplacian_smooth_loss(verts, tri)
    return out


@torch.no_grad()
def init_tet(r, mesh=None):
    """sdf / tet_scale from the density field a NeRF stage left behind (nerf/renderer.py:818-859): scale the grid to the occupied
    region (+ 0.1), then sdf += clamp(sigma - threshold, -1, 1) at the scaled vertices. The mesh branch of the reference needs cubvh's
    signed-distance query, a third-party package this image does not have."""
    if mesh is not None:
        raise NotImplementedError("init_tet(mesh): cubvh's signed-distance query is not available")
    thresh = min(r.mean_density, r.density_thresh) if r.cuda_ray else r.density_thresh
    if r.opt.density_activation == "softplus":
        thresh = thresh * 25
    occupied = r.density(r.verts)["sigma"] > thresh
    r.tet_scale = r.verts[occupied].abs().amax(dim=0) + 1e-1
    r.verts = r.verts * r.tet_scale
    r.sdf.data += (r.density(r.verts)["sigma"] - thresh).clamp(-1, 1).to(r.sdf.dtype)
