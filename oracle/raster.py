"""TEST INFRASTRUCTURE — CPU restatement of the rasterisation contract csrc/raster.hip implements (never imported by the product).

The reference's DMTet stage (nerf/renderer.py:862-964) calls three operations of nvdiffrast — `dr.rasterize` (:900),
`dr.interpolate` (:903-904), `dr.antialias` (:932-933). nvdiffrast is a third-party dependency (requirements.txt:
`git+https://github.com/NVlabs/nvdiffrast/`, unpinned) and is ABSENT from /root/reference and from this image: there is no source
to follow line by line and nothing to run. **Parity unpinned** for these three operations: what is restated here is the published
contract (Laine et al. 2020, "Modular Primitives for High-Performance Differentiable Rendering", §3.1-3.4; tensor formats of the
library's documentation): rast = (u, v, z/w, triangle id + 1) at the pixel centres, nearest surface wins, perspective-correct
barycentrics of vertices 0 / 1, attribute = u a0 + v a1 + (1 - u - v) a2, gradients through u, v into the clip-space positions;
antialiasing = blending across silhouette edges by the position at which the edge crosses the segment between two pixel centres.

It is written DIFFERENTLY from the kernels so that it checks them: discrete decisions (which triangle covers a pixel, which edge
is a silhouette) in numpy float64 with classic screen-space edge functions; everything differentiable as float64 torch expressions
of the clip-space positions, so that the gradients come from torch.autograd and not from the hand-derived formulas of raster.hip.

`Dr` offers the three calls with nvdiffrast's signatures: the reference's own `run_dmtet` runs on it unchanged
(tests/golden/make_goldens_from_reference.py -> tests/golden/dmtet_ref.npz)."""
from __future__ import annotations

import numpy as np
import torch


def _pixel_xy(pos, H, W):
    """screen position in pixels (float64 torch / numpy alike): x = (x / w / 2 + 1 / 2) W"""
    iw = 1.0 / pos[..., 3]
    return (pos[..., 0] * iw * 0.5 + 0.5) * W, (pos[..., 1] * iw * 0.5 + 0.5) * H


def rasterize_ids(pos: np.ndarray, tri: np.ndarray, H: int, W: int):
    """ids [H, W] int64 (-1 = background): the triangle nearest to the eye at every pixel centre (depth ties: the lower index;
    centres exactly on an edge: the top-left fill rule)."""
    pos = pos.astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):     # vertices with w <= 0 are rejected per triangle below
        sx, sy = _pixel_xy(pos, H, W)
        zw = pos[:, 2] / pos[:, 3]
    zbuf = np.full((H, W), np.inf)
    ids = np.full((H, W), -1, np.int64)
    for f in range(tri.shape[0]):
        a, b, c = (int(v) for v in tri[f])
        if min(pos[a, 3], pos[b, 3], pos[c, 3]) <= 0:
            continue
        xs, ys = np.array([sx[a], sx[b], sx[c]]), np.array([sy[a], sy[b], sy[c]])
        area = (xs[1] - xs[0]) * (ys[2] - ys[0]) - (ys[1] - ys[0]) * (xs[2] - xs[0])
        if area == 0:
            continue
        x0, x1 = max(0, int(np.floor(xs.min() - 0.5))), min(W - 1, int(np.ceil(xs.max() - 0.5)))
        y0, y1 = max(0, int(np.floor(ys.min() - 0.5))), min(H - 1, int(np.ceil(ys.max() - 0.5)))
        if x1 < x0 or y1 < y0:
            continue
        px, py = np.meshgrid(np.arange(x0, x1 + 1) + 0.5, np.arange(y0, y1 + 1) + 0.5)
        # screen-space barycentrics from the edge functions (signed areas / area)
        b0 = ((xs[1] - px) * (ys[2] - py) - (ys[1] - py) * (xs[2] - px)) / area
        b1 = ((xs[2] - px) * (ys[0] - py) - (ys[2] - py) * (xs[0] - px)) / area
        b2 = 1.0 - b0 - b1
        # fill rule (top-left in pixel-index space): strictly inside, or exactly on an edge whose interior side is +x, or — for
        # a horizontal edge — +y (the row index). The gradients of the barycentrics are constants of the triangle.
        grads = (((ys[1] - ys[2]) / area, (xs[2] - xs[1]) / area), ((ys[2] - ys[0]) / area, (xs[0] - xs[2]) / area),
                 ((ys[0] - ys[1]) / area, (xs[1] - xs[0]) / area))
        inside = np.ones_like(b0, dtype=bool)
        for bary_i, (gx, gy) in zip((b0, b1, b2), grads):
            owns = bool(gx > 0 or (gx == 0 and gy > 0))
            inside &= (bary_i > 0) | ((bary_i == 0) & owns)
        z = b0 * zw[a] + b1 * zw[b] + b2 * zw[c]                      # z/w is affine in screen space
        inside &= (z >= -1) & (z <= 1)
        sub_z, sub_i = zbuf[y0:y1 + 1, x0:x1 + 1], ids[y0:y1 + 1, x0:x1 + 1]
        win = inside & (z < sub_z)
        sub_z[win] = z[win]
        sub_i[win] = f
    return ids


def _bary(pos_t, tri_t, ids_t, H, W):
    """perspective-correct barycentrics and z/w of the covered pixels as float64 torch expressions of `pos_t`"""
    cov = torch.nonzero(ids_t.reshape(-1) >= 0).flatten()
    f = ids_t.reshape(-1)[cov]
    v = pos_t[tri_t[f]]                                                  # [P, 3, 4]
    sx, sy = _pixel_xy(v, H, W)                                          # [P, 3]
    px = (cov % W).double() + 0.5
    py = (cov // W).double() + 0.5
    e = lambda i, j: (sx[:, i] - px) * (sy[:, j] - py) - (sy[:, i] - py) * (sx[:, j] - px)
    bs = torch.stack([e(1, 2), e(2, 0), e(0, 1)], -1)
    bs = bs / bs.sum(-1, keepdim=True)                                    # screen-space barycentrics
    bp = bs / v[..., 3]
    bp = bp / bp.sum(-1, keepdim=True)                                    # perspective-correct
    zw = (bs * (v[..., 2] / v[..., 3])).sum(-1)
    return cov, f, bp, zw


class Dr:
    """The nvdiffrast calls of run_dmtet on the CPU (float64 inside, float32 tensors in and out, B = 1)."""

    class RasterizeCudaContext:      # dr.RasterizeCudaContext() / dr.RasterizeGLContext() of renderer.py:309-311
        pass

    RasterizeGLContext = RasterizeCudaContext

    @staticmethod
    def rasterize(glctx, pos, tri, resolution):
        H, W = int(resolution[0]), int(resolution[1])
        assert pos.shape[0] == 1
        ids = torch.from_numpy(rasterize_ids(pos[0].detach().numpy(), tri.detach().numpy(), H, W))
        p64 = pos[0].double()
        cov, f, bp, zw = _bary(p64, tri.long(), ids, H, W)
        rast = torch.zeros(H * W, 4, dtype=torch.float64)
        rast = rast.index_put((cov,), torch.stack([bp[:, 0], bp[:, 1], zw.detach(), (f + 1).double()], -1))
        return rast.view(1, H, W, 4).to(pos.dtype), None

    @staticmethod
    def interpolate(attr, rast, tri):
        assert attr.shape[0] == 1 and rast.shape[0] == 1
        H, W = rast.shape[1], rast.shape[2]
        r = rast[0].reshape(-1, 4)
        ids = r[:, 3].detach().long() - 1
        cov = torch.nonzero(ids >= 0).flatten()
        t = tri.long()[ids[cov]]
        a = attr[0].double()
        u, v = r[cov, 0:1].double(), r[cov, 1:2].double()
        val = u * a[t[:, 0]] + v * a[t[:, 1]] + (1 - u - v) * a[t[:, 2]]
        out = torch.zeros(H * W, attr.shape[2], dtype=torch.float64).index_put((cov,), val)
        return out.view(1, H, W, -1).to(attr.dtype), None

    @staticmethod
    def antialias(color, rast, pos, tri):
        assert color.shape[0] == 1
        H, W, C = color.shape[1], color.shape[2], color.shape[3]
        pairs = antialias_pairs(rast[0].detach().numpy(), pos[0].detach().numpy(), tri.detach().numpy(), H, W)
        c = color[0].reshape(-1, C).double()
        if not pairs:
            return color.clone()
        dst, src, va, vb, fpx, fpy, opx, opy, sgn = (torch.tensor(x) for x in zip(*pairs))
        p64 = pos[0].double()
        ax, ay = _pixel_xy(p64[va.long()], H, W)
        bx, by = _pixel_xy(p64[vb.long()], H, W)
        ex, ey, dx, dy = bx - ax, by - ay, (opx - fpx).double(), (opy - fpy).double()
        qx, qy = ax - fpx.double(), ay - fpy.double()
        alpha = (qx * ey - qy * ex) / (dx * ey - dy * ex)
        w = (alpha - 0.5) * sgn.double()                               # sgn = -1: the surface's own pixel receives, +1: the other
        out = c.index_add(0, dst.long(), w[:, None] * (c[src.long()] - c[dst.long()]))
        return out.view(1, H, W, C).to(color.dtype)


def antialias_pairs(rast, pos, tri, H, W):
    """The discrete part of the antialiasing: [(dst pixel, src pixel, edge vertex a, edge vertex b, centre of the nearer surface's
    pixel (x, y), centre of the other pixel (x, y), sign)] for every adjacent pixel pair whose blend weight is not zero."""
    pos = pos.astype(np.float64)
    tri = tri.astype(np.int64)
    ids = rast[..., 3].astype(np.int64) - 1
    zw = rast[..., 2].astype(np.float64)
    sx, sy = _pixel_xy(pos, H, W)
    # edge -> opposite vertices of the triangles that own it
    owners = {}
    for f in range(tri.shape[0]):
        for k in range(3):
            a, b, o = int(tri[f, k]), int(tri[f, (k + 1) % 3]), int(tri[f, (k + 2) % 3])
            owners.setdefault((min(a, b), max(a, b)), []).append((f, o))
    out = []
    for (dxp, dyp) in ((1, 0), (0, 1)):
        ys, xs = np.nonzero(ids[:H - dyp, :W - dxp] != ids[dyp:, dxp:])
        for y0, x0 in zip(ys.tolist(), xs.tolist()):
            x1, y1 = x0 + dxp, y0 + dyp
            t0, t1 = ids[y0, x0], ids[y1, x1]
            fg = (0 if zw[y0, x0] <= zw[y1, x1] else 1) if (t0 >= 0 and t1 >= 0) else (0 if t0 >= 0 else 1)
            t = t1 if fg else t0
            f_px, o_px = ((x1, y1), (x0, y0)) if fg else ((x0, y0), (x1, y1))
            fx, fy, ox, oy = f_px[0] + 0.5, f_px[1] + 0.5, o_px[0] + 0.5, o_px[1] + 0.5
            best = None
            for k in range(3):
                a, b, o = int(tri[t, k]), int(tri[t, (k + 1) % 3]), int(tri[t, (k + 2) % 3])
                ex, ey = sx[b] - sx[a], sy[b] - sy[a]
                others = [oo for (ff, oo) in owners[(min(a, b), max(a, b))] if ff != t]
                if others:
                    o2 = others[0]
                    s0 = ex * (sy[o] - sy[a]) - ey * (sx[o] - sx[a])
                    s1 = ex * (sy[o2] - sy[a]) - ey * (sx[o2] - sx[a])
                    if s0 * s1 < 0:
                        continue
                den = (ox - fx) * ey - (oy - fy) * ex
                if den == 0:
                    continue
                qx, qy = sx[a] - fx, sy[a] - fy
                alpha = (qx * ey - qy * ex) / den
                beta = (qx * (oy - fy) - qy * (ox - fx)) / den
                if 0 < alpha < 1 and 0 <= beta <= 1 and (best is None or alpha < best[0]):
                    best = (alpha, a, b)
            if best is None or best[0] == 0.5:
                continue
            alpha, a, b = best
            pf, po = f_px[1] * W + f_px[0], o_px[1] * W + o_px[0]
            if alpha < 0.5:
                out.append((pf, po, a, b, fx, fy, ox, oy, -1.0))
            else:
                out.append((po, pf, a, b, fx, fy, ox, oy, 1.0))
    return out
